#!/usr/bin/env bash
# tools/round_refresh.sh <tag> -- everything the round's evidence is rebuilt from, in one gpurun call:
# GPU parity suite, smoke, the headline bench line, rocprofv3 trace + counter passes of the same command,
# the secondary benches and the host path.  Outputs under gpurun_out/refresh_<tag>/.
set -uo pipefail
TAG="${1:-run}"
OUT="gpurun_out/refresh_${TAG}"
mkdir -p "${OUT}"
timeout 900 python -m pytest tests -q -m gpu > "${OUT}/pytest_gpu.log" 2>&1; echo "pytest rc=$?"; grep -E "passed|failed" "${OUT}/pytest_gpu.log" | tail -1
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > "${OUT}/smoke.log" 2>&1; echo "smoke rc=$?"
timeout 600 python bench.py > "${OUT}/bench.log" 2>&1; echo "bench rc=$?"; tail -1 "${OUT}/bench.log" > "${OUT}/bench.json"; cut -c1-260 "${OUT}/bench.json"
timeout 900 bash tools/profile.sh "${TAG}" > "${OUT}/profile.log" 2>&1; echo "profile rc=$?"
timeout 600 bash tools/traffic.sh "${TAG}" > "${OUT}/traffic.log" 2>&1; echo "traffic rc=$?"
python tools/traffic_summary.py "gpurun_out/traffic_${TAG}" > "${OUT}/traffic.json" 2>&1 || true
timeout 1200 python tools/bench_extra.py > "${OUT}/bench_extra.jsonl" 2> "${OUT}/bench_extra.err"; echo "bench_extra rc=$?"
timeout 300 python tools/host_path.py > "${OUT}/host_path.txt" 2>&1; echo "host_path rc=$?"
python tools/rocpd_summary.py "gpurun_out/prof_${TAG}" > "${OUT}/rocprof_summary.txt" 2>&1 || true
