#!/usr/bin/env bash
# tools/pmc_weighted.sh <tag> [bench_weighted args...] -- rocprofv3 counter passes (no tracing) over tools/bench_weighted.py,
# then a kernel trace, summarised into gpurun_out/pmc_weighted_<tag>/summary.txt
set -uo pipefail
TAG="${1:-run}"; shift || true
OUT="gpurun_out/pmc_weighted_${TAG}"
mkdir -p "${OUT}"
export TMPDIR=/tmp
CMD=(python tools/bench_weighted.py --check 0 --reps 2 "$@")
pass() { # name counters...
  local name="$1"; shift
  rocprofv3 --pmc "$@" -d "${OUT}/${name}" -o pmc -- "${CMD[@]}" > "${OUT}/${name}.log" 2>&1
  echo "${name} rc=$?"
}
pass sq1 SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU
pass sq2 SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SALU SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE
pass sq3 SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_BRANCH SQ_IFETCH SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM SQ_INST_CYCLES_SMEM
pass tcc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum
rocprofv3 --kernel-trace --stats -d "${OUT}/trace" -o trace -- "${CMD[@]}" > "${OUT}/trace.log" 2>&1
echo "trace rc=$?"
python tools/rocpd_summary.py "${OUT}" > "${OUT}/summary.txt" 2>&1
wc -l "${OUT}/summary.txt"
# gpurun copies back at most 64 MiB: keep the summary and the logs, drop rocprofv3's databases
find "${OUT}" -mindepth 1 -maxdepth 1 -type d -exec rm -rf {} +
