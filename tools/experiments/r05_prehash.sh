# A/B: mhx_lsh_sort_bands on a signature matrix, band digests first (lsh.prehash 0) against hashing inside the scatter pass (1)
mkdir -p gpurun_out
timeout 600 python tools/bench_sort.py > gpurun_out/prehash.txt 2>&1
N=5000000 timeout 600 python tools/bench_sort.py >> gpurun_out/prehash.txt 2>&1
cut -c1-160 gpurun_out/prehash.txt
timeout 900 python -m pytest tests -m gpu -x -q -k "bucket or sort or lsh or candidate or query" 2>&1 | tail -3
