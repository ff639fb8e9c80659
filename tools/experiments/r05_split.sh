mkdir -p gpurun_out
timeout 800 python -m pytest tests -m gpu -x -q -k "weighted" 2>&1 | tail -5 > gpurun_out/split6_tests.txt
{ for lib in "" tools/libmhx_ku2.so tools/libmhx_ku8.so; do
  echo "## lib=${lib:-default} logs in"; MHX_LIBRARY=${lib:+$PWD/$lib} timeout 250 python tools/bench_weighted.py --check 1024 --reps 5 --variants "refill=13;refill=0";
  echo "## lib=${lib:-default} values in"; MHX_LIBRARY=${lib:+$PWD/$lib} timeout 250 python tools/bench_weighted.py --values --check 1024 --reps 5 --variants "refill=13;refill=0";
  echo "## lib=${lib:-default} lognormal"; MHX_LIBRARY=${lib:+$PWD/$lib} timeout 250 python tools/bench_weighted.py --check 512 --rows 20000 --dist lognormal --reps 3 --variants "refill=13;refill=0";
done; } > gpurun_out/split6.txt 2>&1
cat gpurun_out/split6_tests.txt; cut -c1-100 gpurun_out/split6.txt; grep -c '"oracle_equal": true' gpurun_out/split6.txt; grep '"equal_to_first": false' gpurun_out/split6.txt | cut -c1-60
