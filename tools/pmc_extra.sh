#!/usr/bin/env bash
# tools/pmc_extra.sh <tag> -- rocprofv3 counter passes (no tracing) over bench.py WITH its extra configs 3 / 4 / 5, so that
# the kernels either side of the headline launch (band digests, b-bit pack, the two bucketing passes, the weighted walk)
# have counters of this round; summarised into gpurun_out/pmc_extra_<tag>/summary.txt
set -uo pipefail
TAG="${1:-run}"
OUT="gpurun_out/pmc_extra_${TAG}"
mkdir -p "${OUT}"
export TMPDIR=/tmp
CMD=(python bench.py --steps 2 --warmup 1 --cpu-sample 0 --no-e2e --check-rows 0)
pass() { local name="$1"; shift
  timeout 120 rocprofv3 --pmc "$@" -d "${OUT}/${name}" -o pmc -- "${CMD[@]}" > "${OUT}/${name}.log" 2>&1
  echo "${name} rc=$?"; }
pass sq1 SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU
pass sq2 SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE SQ_INSTS_BRANCH SQ_ACTIVE_INST_LDS
pass rd  TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum
pass wr  TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum
pass hit TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum
python tools/rocpd_summary.py "${OUT}" | grep -v -i "rocclr" > "${OUT}/summary.txt" 2>&1
wc -l "${OUT}/summary.txt"
