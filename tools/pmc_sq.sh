#!/bin/bash
# Where do a kernel family's waves spend their cycles?  One rocprofv3 --pmc pass (no trace options) of SQ wave-state counters over two eager steps:
#   tools/pmc_sq.sh [families...]     (SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* count quad-cycles per wave: compare them with each other)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/pmc_sq
rm -rf $out; mkdir -p $out
timeout 600 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_MISC SQ_WAIT_ANY SQ_ACTIVE_INST_ANY -d $out -o pmc --output-format csv -- python bench.py --no-cpu-baseline --no-extras --no-graph --steps 2 --warmup 1 > $out/log.txt 2>&1
python tools/pmc_family.py $out ${@:-compose_fwd compose_bwd conv_bwd conv_rw head_bwd}
rm -rf $out/*/
