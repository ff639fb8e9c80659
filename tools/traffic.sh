#!/usr/bin/env bash
# tools/traffic.sh <tag> -- TCC (L2 <-> fabric) counter passes over tools/traffic_probe.py; run through gpurun.
# Each pass is its own rocprofv3 run with --pmc only (no tracing).  Output: gpurun_out/traffic_<tag>/<pass>/...db
set -uo pipefail
TAG="${1:-run}"
OUT="gpurun_out/traffic_${TAG}"
mkdir -p "${OUT}"
export TMPDIR=/tmp
pass() { local name="$1"; shift
  timeout 90 rocprofv3 --pmc "$@" -d "${OUT}/${name}" -o pmc -- python tools/traffic_probe.py > "${OUT}/${name}.log" 2>&1
  echo "${name} rc=$?"; }
pass rd  TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_DRAM_sum
pass wr  TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_EA0_WRREQ_DRAM_sum
pass fetch FETCH_SIZE
pass write WRITE_SIZE
pass hit TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum
pass sq SQ_INSTS_VALU SQ_INSTS_SALU SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVES
python tools/rocpd_summary.py "${OUT}" > "${OUT}/summary.txt" 2>&1
grep -v rocclr "${OUT}/summary.txt" | cut -c1-150
