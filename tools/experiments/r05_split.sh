mkdir -p gpurun_out
{ echo "## logs in"; timeout 250 python tools/bench_weighted.py --check 2048 --reps 5 --variants "refill=0;refill=0,debug=5;refill=0;refill=0,debug=5;refill=0,debug=4";
  echo "## values in"; timeout 250 python tools/bench_weighted.py --values --check 2048 --reps 5 --variants "refill=0;refill=0,debug=5;refill=0;refill=0,debug=5";
  echo "## lognormal"; timeout 250 python tools/bench_weighted.py --check 1024 --rows 20000 --dist lognormal --reps 3 --variants "refill=0;refill=0,debug=5";
  echo "## few"; timeout 250 python tools/bench_weighted.py --check 300 --rows 300 --density 0.3 --reps 2 --variants "refill=0;refill=13";  } > gpurun_out/split13.txt 2>&1
cut -c1-100 gpurun_out/split13.txt; grep -c '"oracle_equal": true' gpurun_out/split13.txt; grep '"equal_to_first": false' gpurun_out/split13.txt | cut -c1-60
