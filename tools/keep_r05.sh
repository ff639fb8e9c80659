#!/usr/bin/env bash
# tools/keep_r05.sh -- copy what tools/round5_refresh.sh <tag> left under gpurun_out/ (scratch) into profiles/ (tracked); text only
set -euo pipefail
TAG="${1:-r05}"
S=gpurun_out/refresh_${TAG}
cp $S/bench.json profiles/r05_bench.json
cp $S/rocprof_summary.txt profiles/r05_rocprofv3_bench_summary.txt
cp $S/traffic.json profiles/r05_traffic_minhash_bulk.json
cp $S/bench_n2_host.json profiles/r05_bench_n2_host.json
cp $S/bench_n8_host.json profiles/r05_bench_n8_host.json
cp $S/bench_weighted.txt profiles/r05_bench_weighted_fetch_modes.txt
cp $S/box.txt profiles/r05_box.txt
cp gpurun_out/r5_${TAG}/traffic_lsh_sort.json profiles/r05_traffic_lsh_sort.json
{ echo "# Round 5, final kernels: tools/r5_passes.sh over tools/r5_probe.py (kernel trace + separate --pmc passes), 1.25M x 256 uint32 signatures, 32 bands x 8"
  grep -v rocclr gpurun_out/r5_${TAG}/summary.txt | cut -c1-170; echo "# HIP events, no profiler:"; cat gpurun_out/r5_${TAG}/events.json; } > profiles/r05_rocprofv3_sort_and_c5_summary.txt
{ echo "# python -m pytest tests -q -m gpu on an MI355X box (tools/round5_refresh.sh r05); smoke() after it"
  grep -aE "passed|failed|error" $S/pytest_gpu.log | tail -3; echo "# smoke:"; tail -2 $S/smoke.log; } > profiles/r05_pytest_gpu.txt
python tools/readme_table.py
