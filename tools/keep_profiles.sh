#!/usr/bin/env bash
# tools/keep_profiles.sh <tag> -- copy what tools/round_refresh.sh <tag> left under gpurun_out/ (scratch) into
# profiles/ (tracked) under the round's names.  Runs in the build container after the gpurun call has merged its
# outputs back; text only, nothing is re-measured here.
set -euo pipefail
TAG="${1:?tag, e.g. r04}"
SRC="gpurun_out/refresh_${TAG}"
DST="profiles"
[[ -d "${SRC}" ]] || { echo "no ${SRC}"; exit 1; }
keep() { [[ -s "$1" ]] && cp "$1" "${DST}/${TAG}_$2" && echo "kept ${DST}/${TAG}_$2"; return 0; }
keep "${SRC}/bench.json"           bench.json
keep "${SRC}/rocprof_summary.txt"  rocprofv3_bench_summary.txt
keep "${SRC}/traffic.json"         traffic_minhash_bulk.json
keep "${SRC}/bench_extra.jsonl"    bench_extra.jsonl
keep "${SRC}/bench_shapes.jsonl"   bench_shapes.jsonl
keep "${SRC}/bench_kshapes.jsonl"  bench_kshapes.jsonl
keep "${SRC}/bench_sort.txt"       bench_sort.txt
keep "${SRC}/bench_weighted.txt"   bench_weighted_sweeps.txt
keep "${SRC}/host_path.txt"        host_path.txt
keep "gpurun_out/pmc_extra_${TAG}/summary.txt"     rocprofv3_counters_bench_with_extras.txt
keep "gpurun_out/pmc_weighted_${TAG}/summary.txt"  pmc_weighted_walk.txt
{ echo "# python -m pytest tests -q -m gpu on an MI355X box (tools/round_refresh.sh ${TAG}); smoke() after it"
  grep -aE "passed|failed|error" "${SRC}/pytest_gpu.log" | tail -3
  grep -aE "^(FAILED|ERROR)" "${SRC}/pytest_gpu.log" || true
  echo "# smoke:"; tail -2 "${SRC}/smoke.log"; } > "${DST}/${TAG}_pytest_gpu.txt"
echo "kept ${DST}/${TAG}_pytest_gpu.txt"
