#!/usr/bin/env bash
# tools/round5_refresh.sh <tag> -- the round's evidence in one gpurun call (outputs under gpurun_out/refresh_<tag>/):
# GPU parity suite, smoke, the headline bench line (with configs 3 / 4 / 5 at their per-GPU shapes and configs 3 / 5 at their
# stated size), a rocprofv3 kernel trace of the same command, the HBM traffic passes of the headline launch, the counter passes
# over the bucketing / config-5 kernels, and bench.py at N = 2 and N = 8 with the ranks sharing this box's one GPU over the
# explicit host-staged all-gather transport (config 3 end to end across ranks).
set -uo pipefail
TAG="${1:-run}"
OUT="gpurun_out/refresh_${TAG}"
mkdir -p "${OUT}"
# which box this is, and what its RAS counters say before anything runs: a "Memory access fault ... Reason: Unknown" (met twice in
# round 4, never in round 5) can then be told from a kernel's fault by where it happened, not argued
{ echo "# $(date -u +%FT%TZ) host $(hostname)"; for f in /sys/class/drm/card*/device/unique_id; do echo "$f $(cat "$f" 2>/dev/null)"; done
  rocm-smi --showrasinfo all 2>&1 | head -60; rocm-smi --showmemuse --showuse 2>&1 | head -20; dmesg 2>/dev/null | grep -iE "amdgpu|gpu fault|page fault" | tail -20; } > "${OUT}/box.txt" 2>&1
timeout 1500 python -m pytest tests -q -m gpu -p no:cacheprovider > "${OUT}/pytest_gpu.log" 2>&1; echo "pytest rc=$?"; grep -aE "passed|failed" "${OUT}/pytest_gpu.log" | tail -1
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > "${OUT}/smoke.log" 2>&1; rc=$?; echo "smoke rc=${rc}"
if [[ ${rc} -ne 0 ]]; then echo "smoke failed: not profiling on this box"; tail -5 "${OUT}/smoke.log"; exit 1; fi
timeout 900 python bench.py > "${OUT}/bench.log" 2>&1; echo "bench rc=$?"; tail -1 "${OUT}/bench.log" > "${OUT}/bench.json"; cut -c1-260 "${OUT}/bench.json"
PROFILE_ONLY=trace timeout 600 bash tools/profile.sh "${TAG}" > "${OUT}/profile.log" 2>&1; echo "profile rc=$?"
python tools/rocpd_summary.py "gpurun_out/prof_${TAG}" > "${OUT}/rocprof_summary.txt" 2>&1 || true
timeout 400 bash tools/traffic.sh "${TAG}" > "${OUT}/traffic.log" 2>&1; echo "traffic rc=$?"
python tools/traffic_summary.py "gpurun_out/traffic_${TAG}" > "${OUT}/traffic.json" 2> "${OUT}/traffic.err" || true
timeout 400 bash tools/r5_passes.sh "${TAG}" > "${OUT}/r5_passes.log" 2>&1; echo "r5_passes rc=$?"
for n in 2 8; do
  timeout 900 python bench.py --gpus ${n} --share-devices --allgather-transport host --check-rows 1024 > "${OUT}/bench_n${n}_host.log" 2>&1; echo "bench n=${n} rc=$?"
  grep "^{" "${OUT}/bench_n${n}_host.log" | tail -1 > "${OUT}/bench_n${n}_host.json"; cut -c1-200 "${OUT}/bench_n${n}_host.json"
done
{ echo "## config 4: logs in (weighted.refill: 0 = auto: fetcher / walker waves, 13 = the one-wave-per-row kernel, 1 = round 4's; debug 1: rows staged and scanned, not walked; debug 4: the walkers alone)"; timeout 200 python tools/bench_weighted.py --check 2048 --reps 5 --variants "refill=0;refill=13;refill=1;refill=0,debug=1;refill=0,debug=4;refill=5;refill=6;refill=0";
  echo "## config 4, values in"; timeout 200 python tools/bench_weighted.py --values --check 2048 --reps 5 --variants "refill=0;refill=13;refill=0";
  echo "## lognormal weights (sigma 2), 20k rows, logs in / values in"; timeout 200 python tools/bench_weighted.py --check 1024 --rows 20000 --dist lognormal --reps 3 --variants "refill=0;refill=13";
  timeout 200 python tools/bench_weighted.py --values --check 1024 --rows 20000 --dist lognormal --reps 3 --variants "refill=0;refill=13"; } > "${OUT}/bench_weighted.txt" 2>&1; echo "weighted rc=$?"
{ echo "# after the run:"; rocm-smi --showrasinfo all 2>&1 | head -60; dmesg 2>/dev/null | grep -iE "amdgpu|gpu fault|page fault" | tail -20; } >> "${OUT}/box.txt" 2>&1
find gpurun_out -name "*.db" -delete 2>/dev/null; find gpurun_out -type f -size +8M -delete 2>/dev/null; du -sh gpurun_out | tail -1
