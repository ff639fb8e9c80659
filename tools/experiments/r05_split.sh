# the last A/B of the fetcher / walker kernel (kept as run; earlier ones are in profiles/r05_ab_weighted_split.txt)
mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -x -q -k "weighted" 2>&1 | tail -3 > gpurun_out/split15_tests.txt
{ echo "## 256 samples"; timeout 250 python tools/bench_weighted.py --check 512 --rows 100000 --samples 256 --reps 3 --variants "refill=13;refill=0;refill=13;refill=0";
  echo "## 256 samples values"; timeout 250 python tools/bench_weighted.py --values --check 512 --rows 100000 --samples 256 --reps 3 --variants "refill=13;refill=0";
  echo "## 192 samples"; timeout 250 python tools/bench_weighted.py --check 512 --rows 100000 --samples 192 --reps 3 --variants "refill=13;refill=0";
  echo "## 384 samples"; timeout 250 python tools/bench_weighted.py --check 512 --rows 50000 --samples 384 --reps 3 --variants "refill=13;refill=0";
  echo "## 128 samples"; timeout 250 python tools/bench_weighted.py --check 512 --rows 100000 --reps 3 --variants "refill=13;refill=0"; } > gpurun_out/split15.txt 2>&1
cat gpurun_out/split15_tests.txt; cut -c1-100 gpurun_out/split15.txt; grep -c '"oracle_equal": true' gpurun_out/split15.txt; grep '"equal_to_first": false' gpurun_out/split15.txt | cut -c1-60
