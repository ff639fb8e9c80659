# the last A/B of the fetcher / walker kernel (kept as run; earlier ones are in profiles/r05_ab_weighted_split.txt)
mkdir -p gpurun_out
{ echo "## logs in"; timeout 250 python tools/bench_weighted.py --check 1024 --reps 5 --variants "refill=0;refill=0,debug=6;refill=0,debug=7;refill=0;refill=0,debug=6;refill=0,debug=7";
  echo "## values in"; timeout 250 python tools/bench_weighted.py --values --check 1024 --reps 5 --variants "refill=0;refill=0,debug=6;refill=0,debug=7;refill=0;refill=0,debug=6"; } > gpurun_out/split16.txt 2>&1
cut -c1-100 gpurun_out/split16.txt; grep '"equal_to_first": false' gpurun_out/split16.txt | cut -c1-60
