#!/bin/bash
# tools/pmc_sq_cmd.sh <out name> "<command>" [kernel name substrings...]: the two SQ counter passes of tools/pmc_sq_table.sh over ANY small command
# (a microbenchmark under tools/), table -> gpurun_out/<out name>.txt.  DD_LIB etc. are inherited from the environment.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
name=$1; cmd=$2; shift 2
out=gpurun_out/pmc_cmd_$name
rm -rf $out; mkdir -p $out
timeout 600 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA -d $out/p1 -o pmc --output-format csv -- $cmd > $out/log1.txt 2>&1
timeout 600 rocprofv3 --pmc SQ_WAVES SQ_BUSY_CU_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE -d $out/p2 -o pmc --output-format csv -- $cmd > $out/log2.txt 2>&1
python tools/pmc_sq_table.py $out/p1 $out/p2 "$@" > gpurun_out/$name.txt 2>&1
rm -rf $out
cat gpurun_out/$name.txt
