#!/bin/bash
# One gpurun call that refreshes every measured artefact of a state:  tools/refresh_profiles.sh <tag>   (e.g. r02_a)
# Writes gpurun_out/<tag>/...; copy what is to be judged into profiles/ (tools/collect_profiles.py <tag>).
# Trace and counter passes are separate rocprofv3 runs (never --pmc together with a trace option).
set -x
tag=${1:-r02_x}
out=gpurun_out/$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $out/smoke.txt 2>&1
timeout 600 python bench.py > $out/bench.json 2> $out/bench.err
timeout 300 python bench.py --no-cpu-baseline --no-extras --host-inputs 2>/dev/null | tail -1 > $out/host_inputs.txt
timeout 300 python bench.py --mode inference --steps 10 --warmup 2 2>/dev/null | tail -1 > $out/inference.json
timeout 600 rocprofv3 --kernel-trace --stats -d $out/prof -o $tag -- python bench.py --no-cpu-baseline --no-extras > $out/prof.log 2>&1
python tools/rocprof_summary.py $(ls $out/prof/*/*results.db $out/prof/*results.db 2>/dev/null | head -1) 60 > $out/kernel_stats.txt 2>&1
for c in FETCH_SIZE WRITE_SIZE "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 GRBM_GUI_ACTIVE"; do
  n=$(echo $c | cut -d' ' -f1)
  timeout 600 rocprofv3 --pmc $c -d $out/pmc_$n -o pmc --output-format csv -- python bench.py --no-cpu-baseline --no-extras --no-graph --steps 2 --warmup 1 > $out/pmc_$n.log 2>&1
  python tools/pmc_family.py $out/pmc_$n conv_igemm conv_rw wgrad_dma conv_bwd convt_fwd convt_bwd compose_fwd compose_bwd head_fwd head_bwd > $out/pmc_$n.txt 2>&1
done
# the multi-process path of the bench on ONE device (two ranks, gloo transport; RCCL needs >= 2 GPUs): same code above the transport
HSA_ENABLE_IPC_MODE_LEGACY=0 DD_FORCE_DEVICE=0 DD_DIST_BACKEND=gloo timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 2 --steps 5 --warmup 3 --batch 32 > $out/two_ranks_one_gpu.txt 2> $out/two_ranks_one_gpu.err
rm -rf $out/prof $out/pmc_FETCH_SIZE $out/pmc_WRITE_SIZE $out/pmc_SQ_VALU_MFMA_BUSY_CYCLES
cat $out/smoke.txt | tail -2; cut -c1-300 $out/bench.json; cat $out/pmc_*.txt; head -12 $out/kernel_stats.txt
