#!/bin/bash
# SQ wave-state / LDS / instruction counters of the row-streaming compose forward (tools/compose_stream_bench.py, one shape, inference form):
#   tools/pmc_compose_stream.sh [N,H,W]       three rocprofv3 --pmc passes (no trace options), printed per dispatch
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export CS_SHAPES=${1:-128,128,128}
out=gpurun_out/pmc_cs
rm -rf $out; mkdir -p $out
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" \
           "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_SCA" \
           "GRBM_GUI_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_SALU SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_WAVES"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $set -d $out/p$i -o pmc --output-format csv -- python tools/compose_stream_bench.py > $out/log$i.txt 2>&1
  python tools/pmc_print.py $(dirname $(ls $out/p$i/*/*counter_collection.csv $out/p$i/*counter_collection.csv 2>/dev/null | head -1)) compose_stream
done
rm -rf $out/p*/
