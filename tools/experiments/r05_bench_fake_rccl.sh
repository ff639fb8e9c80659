# bench.py's RCCL branch (transport "rccl": the AllGather probe, extra.c3_sharded with the in-place all-gather) run to the end on a 1-GPU box:
# MHX_RCCL_LIBRARY points libmhx at tests/fake_rccl.c (ranks sharing one device).  The numbers mean nothing (the stand-in stages through /dev/shm);
# what is checked is that the code path the 8-GPU run takes completes and its parity gates pass.
mkdir -p gpurun_out
/opt/rocm/bin/hipcc -x c -shared -fPIC -O2 -D__HIP_PLATFORM_AMD__ -I/opt/rocm/include tests/fake_rccl.c -o /tmp/libfake_rccl.so -L/opt/rocm/lib -lamdhip64 || exit 1
for n in 2 8; do
  MHX_RCCL_LIBRARY=/tmp/libfake_rccl.so timeout 600 python bench.py --gpus $n --share-devices --steps 3 --warmup 1 --sets 20000 --c3-rows 30000 --check-rows 256 --clock-warmup 0 > gpurun_out/bench_fake_rccl_n$n.log 2>&1
  echo "n=$n rc=$?"; grep "^{" gpurun_out/bench_fake_rccl_n$n.log | tail -1 > gpurun_out/bench_fake_rccl_n$n.json
  python - <<PY
import json
b=json.load(open("gpurun_out/bench_fake_rccl_n$n.json"))
print(b["n_gpus"], b["config"].get("allgather_transport"), b["config"].get("rccl_library_override"))
print({k:(v if not isinstance(v,(list,dict)) else "...") for k,v in b["allgather"].items()})
c3=b.get("extra",{}).get("c3_sharded",{})
print({k:(v if not isinstance(v,(list,dict)) else "...") for k,v in c3.items()})
print(c3.get("allgather"))
PY
done
