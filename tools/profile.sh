#!/usr/bin/env bash
# tools/profile.sh -- rocprofv3 passes for bench.py on the GPU box (run through gpurun).
# Usage: tools/profile.sh <tag> [bench args...]   -> gpurun_out/prof_<tag>/...
# Counters are collected in their own passes (never together with tracing).
set -uo pipefail
TAG="${1:-run}"; shift || true
OUT="gpurun_out/prof_${TAG}"
mkdir -p "${OUT}"
export TMPDIR=/tmp
# the trace pass runs bench.py's own default step counts (20 timed, 3 warm-up) so that its per-kernel average
# can be set against the bench line; the counter passes only need a few dispatches
# (the trace keeps bench.py's extra configs 3/4/5 so that their kernels appear in the summary too)
TRACE=(python bench.py --cpu-sample 0 --no-e2e --check-rows 0 "$@")
BENCH=(python bench.py --steps 5 --warmup 1 --cpu-sample 0 --no-e2e --no-extra --check-rows 0 "$@")
rocprofv3 --kernel-trace --stats -d "${OUT}/trace" -o trace -- "${TRACE[@]}" > "${OUT}/trace.log" 2>&1
echo "trace rc=$?"
[[ "${PROFILE_ONLY:-}" == "trace" ]] && exit 0
pass() { # name counters...
  local name="$1"; shift
  rocprofv3 --pmc "$@" -d "${OUT}/${name}" -o pmc -- "${BENCH[@]}" > "${OUT}/${name}.log" 2>&1
  echo "${name} rc=$?"
}
pass pmc_sq1 SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU
pass pmc_sq2 SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SALU SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE
pass pmc_fetch FETCH_SIZE
pass pmc_write WRITE_SIZE
find "${OUT}" -name "*.csv" | head -40
