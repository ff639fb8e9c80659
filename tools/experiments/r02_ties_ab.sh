# same-box comparison: tie-tolerant proof inside the first launch (in-tree) vs the previous commit (build/variants/libmhx_base.so)
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; mkdir -p gpurun_out
for round in 1 2; do
for lib in datasketch_amd/libmhx.so build/variants/libmhx_base.so; do
  echo "== $(basename $lib) round $round"
  MHX_LIBRARY="$PWD/$lib" timeout 300 python tools/bench_extra.py --only ragged 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    try: d = json.loads(l)
    except ValueError: continue
    if 'ms' in d: print('   %-70s %.4f ms  redone=%s pairwise=%s blocks=%s' % (d['name'][:70], d['ms'], d.get('sieve_sets_redone'), d.get('pairwise_sets'), d.get('sieve_blocks')))
"
done
done
