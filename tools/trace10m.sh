export TMPDIR=/tmp; cd /root/repo
N=10000000 timeout 300 rocprofv3 --kernel-trace --stats -d gpurun_out/trace10m/trace -o trace -- python tools/ab_option.py lsh.bigbins 0,1 > gpurun_out/trace10m.log 2>&1
python tools/rocpd_summary.py gpurun_out/trace10m 2>&1 | grep -v rocclr | cut -c1-180 > gpurun_out/trace10m.txt
rm -rf gpurun_out/trace10m
