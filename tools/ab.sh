#!/usr/bin/env bash
# tools/ab.sh [bench args] -- same-box A/B of libmhx variants under build/variants/ against the in-tree
# build: two interleaved rounds, prints the mean kernel time per launch of each.
cd "$(dirname "${BASH_SOURCE[0]}")/.."
for round in 1 2; do
  for lib in datasketch_amd/libmhx.so build/variants/libmhx_*.so; do
    ms=$(MHX_LIBRARY="$PWD/$lib" timeout 120 python bench.py --steps 20 --warmup 3 --cpu-sample 0 --no-e2e --check-rows 512 "$@" 2>/dev/null \
         | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('%.4f' % d['roofline']['kernel_ms'])")
    echo "round=$round $(basename $lib) kernel_ms=$ms"
  done
done
