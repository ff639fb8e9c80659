#!/usr/bin/env bash
# tools/ab_probe.sh <script.py> [rounds] -- same-box A/B of libmhx variants (build/variants/libmhx_*.so, tools/build_variant.sh) against the
# in-tree build over any probe script that prints its own timings (tools/r5_probe.py, tools/bench_weighted.py ...): interleaved rounds.
cd "$(dirname "${BASH_SOURCE[0]}")/.."
SCRIPT="$1"; ROUNDS="${2:-2}"; shift; shift || true
for round in $(seq 1 "${ROUNDS}"); do
  for lib in datasketch_amd/libmhx.so build/variants/libmhx_*.so; do
    echo "== round ${round} $(basename ${lib})"
    MHX_LIBRARY="$PWD/$lib" timeout 300 python "${SCRIPT}" "$@" 2>&1 | tail -25
  done
done
