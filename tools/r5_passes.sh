#!/usr/bin/env bash
# tools/r5_passes.sh <tag> -- round 5's profiler passes over tools/r5_probe.py; run through gpurun.
# One kernel-trace run (per-kernel durations) and separate --pmc runs (never combined with tracing): L2 <-> fabric
# requests (EA read / write), L2 hits, LDS bank conflicts and wait shares.  Output: gpurun_out/r5_<tag>/<pass>/...db
set -uo pipefail
TAG="${1:-run}"
OUT="gpurun_out/r5_${TAG}"
mkdir -p "${OUT}"
export TMPDIR=/tmp R5_PROFILED=1
timeout 120 rocprofv3 --kernel-trace -d "${OUT}/trace" -o trace -- python tools/r5_probe.py > "${OUT}/trace.log" 2>&1; echo "trace rc=$?"
pass() { local name="$1"; shift
  timeout 120 rocprofv3 --pmc "$@" -d "${OUT}/${name}" -o pmc -- python tools/r5_probe.py > "${OUT}/${name}.log" 2>&1
  echo "${name} rc=$?"; }
pass rd  TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_DRAM_sum
pass wr  TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_EA0_WRREQ_DRAM_sum
pass hit TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_ATOMIC_sum
pass lds SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_LDS
pass sq  SQ_INSTS_VALU SQ_INSTS_SALU SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVES
pass wait SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD
python tools/rocpd_summary.py "${OUT}" > "${OUT}/summary.txt" 2>&1
python tools/r5_summary_json.py "${OUT}" > "${OUT}/traffic_lsh_sort.json" 2> "${OUT}/traffic_lsh_sort.err"
unset R5_PROFILED
timeout 120 python tools/r5_probe.py > "${OUT}/events.json" 2> "${OUT}/events.err"; echo "events rc=$?"; cat "${OUT}/events.json" | cut -c1-600
find "${OUT}" -name "*.db" -delete 2>/dev/null
grep -v rocclr "${OUT}/summary.txt" | cut -c1-170 | head -150
