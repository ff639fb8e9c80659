#!/usr/bin/env bash
# tools/keep_r06.sh [tag] -- copy what tools/round6_refresh.sh <tag> left under gpurun_out/ (scratch) into profiles/ (tracked); text only
set -uo pipefail
TAG="${1:-r06}"
S=gpurun_out/refresh_${TAG}
keep() { [[ -s "$1" ]] && cp "$1" "profiles/r06_$2" && echo "kept profiles/r06_$2"; return 0; }
keep $S/bench.json bench.json
keep $S/rocprof_summary.txt rocprofv3_bench_summary.txt
keep $S/traffic.json traffic_minhash_bulk.json
keep $S/bench_n2_host.json bench_n2_host.json
keep $S/bench_n8_host.json bench_n8_host.json
keep $S/box.txt box.txt
keep $S/bench_extra.jsonl bench_extra.jsonl
keep $S/host_path.txt host_path.txt
keep $S/bench_sort.txt bench_sort.txt
keep $S/bench_shapes.jsonl bench_shapes.jsonl
keep $S/sweep_weighted_csr.txt sweep_weighted_csr.txt
keep gpurun_out/r5_${TAG}/traffic_lsh_sort.json traffic_lsh_sort.json
keep gpurun_out/pmc_weighted_${TAG}_sparse001/summary.txt pmc_weighted_csr_direct.txt
{ echo "# Round 6, final kernels: tools/r5_passes.sh over tools/r5_probe.py (kernel trace + separate --pmc passes), 1.25M x 256 uint32 signatures, 32 bands x 8"
  grep -v rocclr gpurun_out/r5_${TAG}/summary.txt | cut -c1-170; echo "# HIP events, no profiler:"; cat gpurun_out/r5_${TAG}/events.json; } > profiles/r06_rocprofv3_sort_and_c5_summary.txt
{ echo "# python -m pytest tests -q -m gpu on an MI355X box (tools/round6_refresh.sh ${TAG}); smoke() after it"
  grep -aE "passed|failed|error" $S/pytest_gpu.log | tail -3; echo "# smoke:"; tail -2 $S/smoke.log; } > profiles/r06_pytest_gpu.txt
