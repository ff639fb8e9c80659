set -x
for d in 0.02 0.05 0.1 0.2 0.35 0.5; do
  python tools/bench_weighted.py --csr --density $d --rows 20000 --dim 1024 --samples 64 --variants "direct=1000;direct=1" --check 0
  python tools/bench_weighted.py --csr --density $d --rows 20000 --dim 4096 --samples 128 --variants "direct=1000;direct=1" --check 0
  python tools/bench_weighted.py --csr --density $d --rows 20000 --dim 1024 --samples 256 --variants "direct=1000;direct=1" --check 0
done
