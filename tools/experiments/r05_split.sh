mkdir -p gpurun_out
timeout 800 python -m pytest tests -m gpu -x -q -k "weighted" 2>&1 | tail -5 > gpurun_out/split7_tests.txt
{ echo "## logs in"; timeout 250 python tools/bench_weighted.py --check 2048 --reps 5 --variants "refill=13;refill=0;refill=0,debug=4;refill=13";
  echo "## values in"; timeout 250 python tools/bench_weighted.py --values --check 2048 --reps 5 --variants "refill=13;refill=0;refill=13";
  echo "## lognormal"; timeout 250 python tools/bench_weighted.py --check 1024 --rows 20000 --dist lognormal --reps 3 --variants "refill=13;refill=0";
  echo "## lognormal values"; timeout 250 python tools/bench_weighted.py --values --check 1024 --rows 20000 --dist lognormal --reps 3 --variants "refill=13;refill=0";
  echo "## sparse"; timeout 250 python tools/bench_weighted.py --check 1024 --rows 20000 --density 0.05 --reps 3 --variants "refill=13;refill=0";
  echo "## dim 2048"; timeout 250 python tools/bench_weighted.py --check 1024 --rows 50000 --dim 2048 --reps 3 --variants "refill=13;refill=0";
  echo "## samples 256"; timeout 250 python tools/bench_weighted.py --check 512 --rows 50000 --samples 256 --reps 3 --variants "refill=13;refill=0"; } > gpurun_out/split7.txt 2>&1
cat gpurun_out/split7_tests.txt; cut -c1-100 gpurun_out/split7.txt; grep -c '"oracle_equal": true' gpurun_out/split7.txt; grep '"equal_to_first": false' gpurun_out/split7.txt | cut -c1-60
