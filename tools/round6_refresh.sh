#!/usr/bin/env bash
# tools/round6_refresh.sh <tag> -- round 6's evidence in one gpurun call (outputs under gpurun_out/refresh_<tag>/): GPU parity suite,
# smoke, the headline bench line with every extra config (c3 / c4 / c4_sparse / c5 per GPU, c3 / c5 at 10M rows), a rocprofv3 kernel
# trace of the same command, the stamped HBM-traffic passes of the headline launch, the counter passes over the bucketing / config-5
# kernels and over the CSR weighted kernel, bench.py at N = 2 and N = 8 sharing this box's GPU over the host-staged transport (by-band
# exchange and config 5 across ranks included), and the secondary benches.
# MHX_GIT_COMMIT (exported by the caller: there is no .git on the box) goes into the stamps of the counter profiles.
set -uo pipefail
TAG="${1:-run}"
OUT="gpurun_out/refresh_${TAG}"
mkdir -p "${OUT}"
{ echo "# $(date -u +%FT%TZ) host $(hostname) commit ${MHX_GIT_COMMIT:-unknown}"; for f in /sys/class/drm/card*/device/unique_id; do echo "$f $(cat "$f" 2>/dev/null)"; done
  rocm-smi --showrasinfo all 2>&1 | head -40; } > "${OUT}/box.txt" 2>&1
timeout 1500 python -m pytest tests -q -m gpu -p no:cacheprovider > "${OUT}/pytest_gpu.log" 2>&1; echo "pytest rc=$?"; grep -aE "passed|failed" "${OUT}/pytest_gpu.log" | tail -1
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > "${OUT}/smoke.log" 2>&1; rc=$?; echo "smoke rc=${rc}"
if [[ ${rc} -ne 0 ]]; then echo "smoke failed: not profiling on this box"; tail -5 "${OUT}/smoke.log"; exit 1; fi
# the HBM-traffic passes of the headline launch first: their stamped summary goes where bench.py looks for it (profiles/ of THIS copy of the
# tree), so that the bench line below replays counters taken on the very binary it times
timeout 400 bash tools/traffic.sh "${TAG}" > "${OUT}/traffic.log" 2>&1; echo "traffic rc=$?"
python tools/traffic_summary.py "gpurun_out/traffic_${TAG}" > "${OUT}/traffic.json" 2> "${OUT}/traffic.err" || true
[[ -s "${OUT}/traffic.json" ]] && cp "${OUT}/traffic.json" profiles/r06_traffic_minhash_bulk.json
timeout 900 python bench.py > "${OUT}/bench.log" 2>&1; echo "bench rc=$?"; tail -1 "${OUT}/bench.log" > "${OUT}/bench.json"; cut -c1-260 "${OUT}/bench.json"
PROFILE_ONLY=trace timeout 600 bash tools/profile.sh "${TAG}" > "${OUT}/profile.log" 2>&1; echo "profile rc=$?"
python tools/rocpd_summary.py "gpurun_out/prof_${TAG}" > "${OUT}/rocprof_summary.txt" 2>&1 || true
timeout 400 bash tools/r5_passes.sh "${TAG}" > "${OUT}/r5_passes.log" 2>&1; echo "r5_passes rc=$?"
timeout 300 bash tools/pmc_weighted.sh "${TAG}_sparse001" --csr --density 0.01 --rows 100000 --variants "path=0" > "${OUT}/pmc_sparse001.log" 2>&1; echo "pmc sparse rc=$?"
for n in 2 8; do
  timeout 900 python bench.py --gpus ${n} --share-devices --allgather-transport host --check-rows 1024 > "${OUT}/bench_n${n}_host.log" 2>&1; echo "bench n=${n} rc=$?"
  grep "^{" "${OUT}/bench_n${n}_host.log" | tail -1 > "${OUT}/bench_n${n}_host.json"; cut -c1-200 "${OUT}/bench_n${n}_host.json"
done
timeout 400 python tools/bench_extra.py > "${OUT}/bench_extra.jsonl" 2> "${OUT}/bench_extra.err"; echo "bench_extra rc=$?"
timeout 300 python tools/host_path.py > "${OUT}/host_path.txt" 2>&1; echo "host_path rc=$?"
timeout 200 python tools/bench_sort.py > "${OUT}/bench_sort.txt" 2>&1; echo "sort rc=$?"
timeout 300 python tools/bench_shapes.py > "${OUT}/bench_shapes.jsonl" 2> "${OUT}/bench_shapes.err"; echo "shapes rc=$?"
timeout 300 bash tools/sweep_weighted_csr.sh > "${OUT}/sweep_weighted_csr.txt" 2>&1; echo "sweep rc=$?"
{ echo "# after the run:"; rocm-smi --showrasinfo all 2>&1 | head -40; } >> "${OUT}/box.txt" 2>&1
find gpurun_out -name "*.db" -delete 2>/dev/null; find gpurun_out -type f -size +8M -delete 2>/dev/null; du -sh gpurun_out | tail -1
