# two counter passes of one convolution shape: tools/_pmc2.sh <tag> <one_conv args...>
tag=$1; shift
export TMPDIR=/tmp
out=gpurun_out/r5/pmc_$tag
mkdir -p $out
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY \
  --kernel-trace -d $out/a -- python3 tools/one_conv.py "$@" > $out/a.log 2>&1
rocprofv3 --pmc SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_SALU \
  --kernel-trace -d $out/b -- python3 tools/one_conv.py "$@" > $out/b.log 2>&1
rocprofv3 --pmc TA_BUSY_avr TCP_PENDING_STALL_CYCLES_sum TCC_HIT_sum TCC_MISS_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum GRBM_GUI_ACTIVE \
  --kernel-trace -d $out/c -- python3 tools/one_conv.py "$@" > $out/c.log 2>&1
for p in a b c; do python3 tools/pmc_db.py $out/$p p; tail -2 $out/$p.log; done > gpurun_out/r5/pmc_$tag.txt 2>&1
cp $out/a.log gpurun_out/r5/pmc_$tag.a.log; rm -rf $out
