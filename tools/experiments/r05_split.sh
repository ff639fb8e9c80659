mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -x -q -k "weighted" 2>&1 | tail -3 > gpurun_out/split14_tests.txt
{ echo "## dim 2048"; timeout 250 python tools/bench_weighted.py --check 1024 --rows 100000 --dim 2048 --reps 3 --variants "refill=13;refill=0;refill=13;refill=0";
  echo "## dim 2048 values"; timeout 250 python tools/bench_weighted.py --values --check 1024 --rows 100000 --dim 2048 --reps 3 --variants "refill=13;refill=0";
  echo "## dim 1024"; timeout 250 python tools/bench_weighted.py --check 1024 --rows 100000 --dim 1024 --reps 3 --variants "refill=13;refill=0;refill=13;refill=0";
  echo "## dim 1024 values"; timeout 250 python tools/bench_weighted.py --values --check 1024 --rows 100000 --dim 1024 --reps 3 --variants "refill=13;refill=0";
  echo "## dim 3000"; timeout 250 python tools/bench_weighted.py --check 1024 --rows 100000 --dim 3000 --reps 3 --variants "refill=13;refill=0"; } > gpurun_out/split14.txt 2>&1
cat gpurun_out/split14_tests.txt; cut -c1-100 gpurun_out/split14.txt; grep -c '"oracle_equal": true' gpurun_out/split14.txt; grep '"equal_to_first": false' gpurun_out/split14.txt | cut -c1-60
