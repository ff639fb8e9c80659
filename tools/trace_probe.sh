#!/usr/bin/env bash
# tools/trace_probe.sh <tag> <script.py> [args] -- rocprofv3 --kernel-trace --stats of a probe script under the in-tree libmhx and under every
# variant in build/variants/ (tools/build_variant.sh): per-kernel durations side by side in gpurun_out/trace_<tag>/<lib>.txt
cd "$(dirname "${BASH_SOURCE[0]}")/.."
TAG="$1"; SCRIPT="$2"; shift; shift
export TMPDIR=/tmp
OUT="gpurun_out/trace_${TAG}"; mkdir -p "${OUT}"
for lib in datasketch_amd/libmhx.so build/variants/libmhx_*.so; do
  name="$(basename ${lib} .so)"
  MHX_LIBRARY="$PWD/$lib" timeout 300 rocprofv3 --kernel-trace --stats -d "${OUT}/${name}/trace" -o trace -- python "${SCRIPT}" "$@" > "${OUT}/${name}.log" 2>&1
  python tools/rocpd_summary.py "${OUT}/${name}" 2>&1 | grep -v rocclr | cut -c1-200 > "${OUT}/${name}.txt"
  rm -rf "${OUT}/${name}"
done
