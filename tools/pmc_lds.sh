#!/bin/bash
# LDS bank-conflict share per kernel family (SQ_LDS_BANK_CONFLICT = extra cycles, SQ_LDS_IDX_ACTIVE = all LDS-array cycles), one rocprofv3 --pmc pass
# (no trace options) over two eager steps of the bench:  tools/pmc_lds.sh [families...]
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/pmc_lds
rm -rf $out; mkdir -p $out
timeout 600 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_BUSY_CU_CYCLES -d $out -o pmc --output-format csv -- python bench.py --no-cpu-baseline --no-extras --no-graph --steps 2 --warmup 1 > $out/log.txt 2>&1
python tools/pmc_family.py $out ${@:-conv_igemm wgrad_dma conv_bwd compose_fwd compose_bwd head_fwd head_bwd}
rm -rf $out/*/
