cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; mkdir -p gpurun_out/prof_list
for lib in datasketch_amd/libmhx.so build/variants/libmhx_base.so; do
  tag=$(basename $lib .so)
  (cd /tmp && MHX_LIBRARY="$GRAFT_REPO_ROOT/$lib" timeout 200 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof_list/$tag/trace -o trace -- python $GRAFT_REPO_ROOT/bench.py --steps 30 --warmup 3 --cpu-sample 0 --no-e2e --no-extra --check-rows 512 > /dev/null 2>&1)
  echo "== $tag"; python tools/rocpd_summary.py gpurun_out/prof_list/$tag | grep -v "^#" | head -6
done
