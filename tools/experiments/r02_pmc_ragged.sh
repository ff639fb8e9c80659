# counter passes (no tracing) over the two ragged corpora of tools/bench_extra.py: what bounds short / ragged sets?
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_ragged; mkdir -p $OUT
cd /tmp
MHX_REPEATS_RATE=0.0 timeout 200 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU -d $OUT/sq1 -o pmc -- python $GRAFT_REPO_ROOT/tools/bench_extra.py --only repeats > $OUT/sq1.log 2>&1
MHX_REPEATS_RATE=0.0 timeout 200 rocprofv3 --pmc SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SALU SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE -d $OUT/sq2 -o pmc -- python $GRAFT_REPO_ROOT/tools/bench_extra.py --only repeats > $OUT/sq2.log 2>&1
MHX_REPEATS_RATE=0.0 timeout 200 rocprofv3 --pmc SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VMEM SQ_WAVES SQ_WAIT_INST_ANY -d $OUT/sq3 -o pmc -- python $GRAFT_REPO_ROOT/tools/bench_extra.py --only repeats > $OUT/sq3.log 2>&1
cd $GRAFT_REPO_ROOT
python tools/rocpd_summary.py gpurun_out/pmc_ragged | grep "Li0ELi1E\|u64, 0, 1\|^#" 
