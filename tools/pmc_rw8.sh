#!/bin/bash
# Wave-state counters of the 64 -> 64 forward conv (conv_rw8_kernel<2,8>), three separate --pmc passes:  tools/pmc_rw8.sh  (on the GPU box)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
out=$R/gpurun_out/pmc_rw8
mkdir -p $out
i=0
for c in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES" \
         "SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL" \
         "SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_VMEM_WR_TA_DATA_FIFO_FULL SQ_INST_LEVEL_VMEM SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_INST_LEVEL_LDS"; do
  i=$((i+1))
  FWD_ONLY=0 timeout 300 rocprofv3 --pmc $c -d $out/p$i -o pmc --output-format csv -- python $R/tools/fwd_bench.py > $out/p$i.log 2>&1
  (cd $R && python tools/pmc_family.py gpurun_out/pmc_rw8/p$i conv_rw8 > gpurun_out/pmc_rw8/p$i.txt 2>&1)
  rm -rf $out/p$i
done
cat $out/p*.txt
