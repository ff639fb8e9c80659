#!/usr/bin/env bash
# tools/ab.sh [bench args] -- same-box A/B of libmhx variants under build/variants/ against the in-tree
# build: interleaved rounds (boxes differ by +-5 %, so only numbers of one call compare), prints the mean kernel
# time per launch of each.  AB_EXTRA="repeats" adds tools/bench_extra.py --only <name> per library.
cd "$(dirname "${BASH_SOURCE[0]}")/.."
for round in 1 2 3; do
  for lib in datasketch_amd/libmhx.so build/variants/libmhx_*.so; do
    ms=$(MHX_LIBRARY="$PWD/$lib" timeout 120 python bench.py --steps 20 --warmup 3 --cpu-sample 0 --no-e2e --no-extra --check-rows 512 "$@" 2>/dev/null \
         | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('%.4f' % d['roofline']['kernel_ms'])")
    echo "round=$round $(basename $lib) kernel_ms=$ms"
  done
done
for name in ${AB_EXTRA:-}; do
  for lib in datasketch_amd/libmhx.so build/variants/libmhx_*.so; do
    echo "== $(basename $lib) bench_extra --only $name"
    MHX_LIBRARY="$PWD/$lib" timeout 200 python tools/bench_extra.py --only "$name" 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    try: d = json.loads(l)
    except ValueError: continue
    if 'ms' in d: print('   %-70s %.4f ms' % (d['name'][:70], d['ms']))
"
  done
done
