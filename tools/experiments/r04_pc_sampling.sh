#!/usr/bin/env bash
# tools/experiments/r04_pc_sampling.sh <case> -- rocprofv3 PC sampling (beta) over tools/experiments/r04_pc_probe.py, own run, no tracing, no counters.
# Stochastic (hardware) sampling first, host-trap sampling if the device refuses it.  Output: gpurun_out/pcs_<case>/
set -uo pipefail
CASE="${1:-ragged100}"
OUT="gpurun_out/pcs_${CASE}"
mkdir -p "${OUT}"
export TMPDIR=/tmp ROCPROFILER_PC_SAMPLING_BETA_ENABLED=1 CASE REPS="${REPS:-100}"
timeout 150 rocprofv3 --pc-sampling-beta-enabled --pc-sampling-method stochastic --pc-sampling-unit cycles --pc-sampling-interval 1048576 \
  -d "${OUT}/stochastic" -o pcs --output-format csv -- python tools/experiments/r04_pc_probe.py > "${OUT}/stochastic.log" 2>&1
echo "stochastic rc=$?"; tail -3 "${OUT}/stochastic.log"
if ! ls "${OUT}"/stochastic/*pc_sampling* > /dev/null 2>&1; then
  timeout 150 rocprofv3 --pc-sampling-beta-enabled --pc-sampling-method host_trap --pc-sampling-unit time --pc-sampling-interval 1 \
    -d "${OUT}/host_trap" -o pcs --output-format csv -- python tools/experiments/r04_pc_probe.py > "${OUT}/host_trap.log" 2>&1
  echo "host_trap rc=$?"; tail -3 "${OUT}/host_trap.log"
fi
find "${OUT}" -type f | head -20; du -sh "${OUT}"
for f in $(find "${OUT}" -name "*pc_sampling*.csv" | head -2); do echo "== $f"; head -3 "$f"; wc -l "$f"; done
