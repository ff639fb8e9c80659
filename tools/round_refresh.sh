#!/usr/bin/env bash
# tools/round_refresh.sh <tag> -- everything the round's evidence is rebuilt from, in one gpurun call:
# GPU parity suite, smoke, the headline bench line, a rocprofv3 kernel trace of the same command, counter passes
# (own runs, no tracing) over bench.py with its extra configs and over the weighted walk, the HBM traffic passes,
# the secondary benches and the host path.  Outputs under gpurun_out/refresh_<tag>/.
set -uo pipefail
TAG="${1:-run}"
OUT="gpurun_out/refresh_${TAG}"
mkdir -p "${OUT}"
timeout 1100 python -m pytest tests -q -m gpu -p no:cacheprovider > "${OUT}/pytest_gpu.log" 2>&1; echo "pytest rc=$?"; grep -aE "passed|failed" "${OUT}/pytest_gpu.log" | tail -1
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > "${OUT}/smoke.log" 2>&1; rc=$?; echo "smoke rc=${rc}"
# a box whose GPU faults on every launch (seen once: "Memory access fault ... Reason: Unknown" from every process) would make
# each profiler pass below sit out its time limit: stop here instead of spending the budget on it
if [[ ${rc} -ne 0 ]]; then echo "smoke failed: not profiling on this box"; tail -5 "${OUT}/smoke.log"; exit 1; fi
timeout 600 python bench.py > "${OUT}/bench.log" 2>&1; echo "bench rc=$?"; tail -1 "${OUT}/bench.log" > "${OUT}/bench.json"; cut -c1-260 "${OUT}/bench.json"
PROFILE_ONLY=trace timeout 300 bash tools/profile.sh "${TAG}" > "${OUT}/profile.log" 2>&1; echo "profile rc=$?"
python tools/rocpd_summary.py "gpurun_out/prof_${TAG}" > "${OUT}/rocprof_summary.txt" 2>&1 || true
timeout 400 bash tools/traffic.sh "${TAG}" > "${OUT}/traffic.log" 2>&1; echo "traffic rc=$?"
python tools/traffic_summary.py "gpurun_out/traffic_${TAG}" > "${OUT}/traffic.json" 2> "${OUT}/traffic.err" || true
timeout 500 bash tools/pmc_extra.sh "${TAG}" > "${OUT}/pmc_extra.log" 2>&1; echo "pmc_extra rc=$?"
timeout 300 bash tools/pmc_weighted.sh "${TAG}" --variants "path=0" > "${OUT}/pmc_weighted.log" 2>&1; echo "pmc_weighted rc=$?"
timeout 400 python tools/bench_extra.py > "${OUT}/bench_extra.jsonl" 2> "${OUT}/bench_extra.err"; echo "bench_extra rc=$?"
timeout 300 python tools/host_path.py > "${OUT}/host_path.txt" 2>&1; echo "host_path rc=$?"
timeout 300 python tools/bench_shapes.py > "${OUT}/bench_shapes.jsonl" 2> "${OUT}/bench_shapes.err"; echo "shapes rc=$?"
timeout 200 python tools/bench_sort.py > "${OUT}/bench_sort.txt" 2>&1; echo "sort rc=$?"
{ for d in 1.0 0.3 0.05 0.01; do timeout 200 python tools/bench_weighted.py --rows 20000 --density $d --variants "path=0;path=2" --reps 3;
    timeout 200 python tools/bench_weighted.py --csr --density $d --rows 20000 --variants "path=0;path=2" --reps 3; done;
  timeout 200 python tools/bench_weighted.py --rows 20000 --dist lognormal --variants "path=0;path=2";
  timeout 200 python tools/bench_weighted.py --rows 20000 --dist sorted --variants "path=0;path=2";
  echo "## config 4: one wave per row (kernel=0), chunk after chunk (2), one workgroup per row (1: round 3), the last lanes never rescued (rescue=-1), round 3 plan launches (plan=1)";
  timeout 200 python tools/bench_weighted.py --check 2048 --reps 5 --variants "kernel=0;kernel=2;kernel=1;rescue=-1;plan=1;path=2";
  echo "## config 4, values in (the device takes numpy's log)"; timeout 200 python tools/bench_weighted.py --values --check 0 --reps 5 --variants "kernel=0;kernel=1";
  echo "## lognormal weights, 20k rows"; timeout 200 python tools/bench_weighted.py --rows 20000 --dist lognormal --check 2048 --reps 4 --variants "kernel=0;rescue=-1;kernel=1"; } > "${OUT}/bench_weighted.txt" 2>&1; echo "weighted rc=$?"
timeout 400 python tools/bench_shapes.py --cases k128,k136,k150,k160,k176,k192,k200,k216,k240,k256 --packed 0 --p3 0,1 --share 0,1 --reps 10 > "${OUT}/bench_kshapes.jsonl" 2> "${OUT}/bench_kshapes.err"; echo "kshapes rc=$?"
# the rocprofv3 databases are hundreds of MB; what is judged are the summaries made from them above
find gpurun_out -name "*.db" -delete 2>/dev/null; find gpurun_out -type f -size +8M -delete 2>/dev/null; du -sh gpurun_out | tail -1
