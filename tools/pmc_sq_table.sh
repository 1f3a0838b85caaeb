#!/bin/bash
# tools/pmc_sq_table.sh: two rocprofv3 --pmc passes (8 SQ counters each, no trace options) over two eager training steps of bench.py -> gpurun_out/pmc_sq_table.txt
#   SQT_CMD="python tools/cfg3_step.py heavy 8 2" SQT_OUT=gpurun_out/sq_table_cfg3_heavy.txt tools/pmc_sq_table.sh conv_ks wgrad conv_igemm_ws conv_pw   (another workload)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/pmc_sqt
cmd=${SQT_CMD:-python bench.py --no-cpu-baseline --no-extras --no-graph --steps 2 --warmup 1}
res=${SQT_OUT:-gpurun_out/pmc_sq_table.txt}
rm -rf $out; mkdir -p $out
timeout 600 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA -d $out/p1 -o pmc --output-format csv -- $cmd > $out/log1.txt 2>&1
timeout 600 rocprofv3 --pmc SQ_WAVES SQ_BUSY_CU_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE -d $out/p2 -o pmc --output-format csv -- $cmd > $out/log2.txt 2>&1
python tools/pmc_sq_table.py $out/p1 $out/p2 "$@" > $res 2>&1
rm -rf $out/p1 $out/p2
cat $res
