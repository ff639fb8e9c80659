#!/usr/bin/env bash
# tools/trace_extra.sh <tag> -- rocprofv3 kernel trace of tools/bench_extra.py (every secondary kernel), summarised
set -uo pipefail
TAG="${1:-run}"
OUT="gpurun_out/prof_extra_${TAG}"
mkdir -p "${OUT}"
export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d "${OUT}/trace" -o trace -- python tools/bench_extra.py --weighted-rows 20000 > "${OUT}/bench_extra.jsonl" 2> "${OUT}/trace.log"
echo "trace rc=$?"
python tools/rocpd_summary.py "${OUT}" > "${OUT}/summary.txt" 2>&1
wc -l "${OUT}/summary.txt" "${OUT}/bench_extra.jsonl"
