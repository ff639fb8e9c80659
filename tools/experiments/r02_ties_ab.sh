cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; mkdir -p gpurun_out
timeout 500 python -m pytest tests/test_gpu_round2.py tests/test_gpu_parity.py -m gpu -x -q -k "repeat or duplic or sieve or fuzz or ragged or golden or config1 or adversarial or boundaries" 2>&1 | tail -n 8
for round in 1 2; do
for opt in 0 1; do
  echo "== minhash.ties=$opt (1 = dedup pass only) round $round"
  timeout 200 python tools/bench_extra.py --only repeats --opt minhash.ties=$opt 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    try: d = json.loads(l)
    except ValueError: continue
    if 'ms' in d: print('   %-70s %.4f ms  redone=%s pairwise=%s' % (d['name'][:70], d['ms'], d.get('sieve_sets_redone'), d.get('pairwise_sets')))
"
done
done
timeout 120 python bench.py --steps 20 --warmup 3 --cpu-sample 0 --no-e2e --no-extra --check-rows 512 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('headline kernel_ms', d['roofline']['kernel_ms'])"
