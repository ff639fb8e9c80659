mkdir -p gpurun_out
{ echo "## logs in"; timeout 250 python tools/bench_weighted.py --check 2048 --reps 5 --variants "refill=0;refill=14;refill=0;refill=14";
  echo "## values in"; timeout 250 python tools/bench_weighted.py --values --check 2048 --reps 5 --variants "refill=0;refill=14;refill=0;refill=14";
  echo "## lognormal"; timeout 250 python tools/bench_weighted.py --check 1024 --rows 20000 --dist lognormal --reps 3 --variants "refill=0;refill=14";
  echo "## few rows"; timeout 250 python tools/bench_weighted.py --check 300 --rows 300 --reps 3 --variants "refill=0;refill=14";
  echo "## 1 row"; timeout 250 python tools/bench_weighted.py --check 1 --rows 1 --reps 3 --variants "refill=0;refill=14";
  echo "## 7 rows"; timeout 250 python tools/bench_weighted.py --check 7 --rows 7 --reps 3 --variants "refill=0;refill=14";
  echo "## sparse"; timeout 250 python tools/bench_weighted.py --check 1024 --rows 20000 --density 0.05 --reps 3 --variants "refill=0;refill=14"; } > gpurun_out/split10.txt 2>&1
cut -c1-100 gpurun_out/split10.txt; grep -c '"oracle_equal": true' gpurun_out/split10.txt;  grep '"equal_to_first": false' gpurun_out/split10.txt | cut -c1-60
