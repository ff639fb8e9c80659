#!/usr/bin/env bash
# tools/ab_weighted.sh -- same-box A/B of libmhx variants (build/variants/) on the weighted benchmark
cd "$(dirname "${BASH_SOURCE[0]}")/.."
for lib in datasketch_amd/libmhx.so build/variants/libmhx_*.so; do
  echo "== $(basename $lib)"
  MHX_LIBRARY="$PWD/$lib" timeout 300 python tools/bench_extra.py --only weighted --weighted-rows ${ROWS:-40000} 2>&1 | grep -o '"name": "[^"]*", "ms": [0-9.]*'
done
