#!/bin/bash
# Round 6: one gpurun call that refreshes the measured artefacts of the closing state:  tools/refresh_r06.sh <tag>   (e.g. r06_a)
# (tools/b96_knockouts.sh build for "NO_DMA NO_WROLE NO_DROLE NO_FLUSH NO_SEL NO_DMA+NO_WROLE NO_DMA+NO_DROLE STAMP NO_DMA+STAMP" on the CPU side first).
# Writes gpurun_out/<tag>/...; tools/collect_profiles.py <tag> "<note>" copies what is to be judged into profiles/.
# Trace and counter passes are separate rocprofv3 runs (never --pmc together with a trace option).
set -x
tag=${1:-r06_a}
out=gpurun_out/$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $out/smoke.txt 2>&1
timeout 900 python bench.py --dump-launches $out/conv_launches.json > $out/bench.json 2> $out/bench.err
timeout 300 python bench.py --mode inference --steps 10 --warmup 2 2>/dev/null | tail -1 > $out/inference.json
timeout 600 rocprofv3 --kernel-trace --stats -d $out/prof -o $tag -- python bench.py --no-cpu-baseline --no-extras > $out/prof.log 2>&1
python tools/rocprof_summary.py $(ls $out/prof/*/*results.db $out/prof/*results.db 2>/dev/null | head -1) 60 > $out/kernel_stats.txt 2>&1
timeout 600 rocprofv3 --kernel-trace --stats -d $out/prof_inf -o inf -- python bench.py --mode inference --steps 10 --warmup 2 > $out/prof_inf.log 2>&1
python tools/rocprof_summary.py $(ls $out/prof_inf/*/*results.db $out/prof_inf/*results.db 2>/dev/null | head -1) 40 > $out/inference_kernel_stats.txt 2>&1
for cfg in light heavy; do
  timeout 600 rocprofv3 --kernel-trace --stats -d $out/prof_cfg3$cfg -o cfg3 -- python tools/cfg3_step.py $cfg 8 5 > $out/prof_cfg3$cfg.log 2>&1
  python tools/rocprof_summary.py $(ls $out/prof_cfg3$cfg/*/*results.db $out/prof_cfg3$cfg/*results.db 2>/dev/null | head -1) 40 > $out/cfg3_${cfg}_kernel_stats.txt 2>&1
  grep "^cfg-3" $out/prof_cfg3$cfg.log >> $out/cfg3_${cfg}_kernel_stats.txt
  python tools/cfg3_launches.py $cfg 8 > $out/cfg3_${cfg}_launches.txt 2>&1
done
for c in FETCH_SIZE WRITE_SIZE "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 GRBM_GUI_ACTIVE"; do
  n=$(echo $c | cut -d' ' -f1)
  timeout 600 rocprofv3 --pmc $c -d $out/pmc_$n -o pmc --output-format csv -- python bench.py --no-cpu-baseline --no-extras --no-graph --steps 2 --warmup 1 > $out/pmc_$n.log 2>&1
  python tools/pmc_family.py $out/pmc_$n conv_igemm conv_rw conv_ks conv_pw wgrad_pw wgrad_dma conv_bwd conv_bwd96 convt_fwd convt_bwd compose_stream_fwd compose_stream_bwd compose_stream_wgrad head_fwd head_bwd maxpool assemble_input > $out/pmc_$n.txt 2>&1
done
tools/pmc_sq_table.sh > /dev/null 2>&1; cp gpurun_out/pmc_sq_table.txt $out/sq_table.txt
HSA_ENABLE_IPC_MODE_LEGACY=0 DD_FORCE_DEVICE=0 DD_DIST_BACKEND=gloo timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 2 --steps 5 --warmup 3 --batch 32 > $out/two_ranks_one_gpu.txt 2> $out/two_ranks_one_gpu.err
HSA_ENABLE_IPC_MODE_LEGACY=0 DD_FORCE_COLLECTIVES=1 timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29519 bench.py --gpus 1 --steps 20 --warmup 5 --no-extras --no-cpu-baseline > $out/one_rank_rccl.txt 2> $out/one_rank_rccl.err
( DD_DETERMINISTIC=1 python tools/det_diag.py cfg2; python tools/det_diag.py cfg2 ) > $out/deterministic.txt 2>&1
python tools/example_profile.py 8 > $out/example_profile.txt 2>&1
# round 6: the new kernel against the two launches it replaced (same box, one process per variant), its knock-outs and cycle stamps, the gate diagnosis
( timeout 900 python tools/ab_bench.py "conv_bwd96:" "two_launches:DD_CONV_BWD96=0" --repeat 3 ) > $out/ab_round6.txt 2>&1
( VARIANTS="NO_DMA NO_WROLE NO_DROLE NO_FLUSH NO_SEL NO_DMA+NO_WROLE NO_DMA+NO_DROLE" tools/b96_knockouts.sh run
  for v in STAMP NO_DMA+STAMP; do echo "== $v"; DD_LIB=tools/exp/libdd_b96_$v.so python tools/conv_bwd_bench.py 96 96 64 128 bf16 1 2>&1 | grep "^block" | sort | uniq; done ) > $out/b96_knockouts.txt 2>&1
for sh in "96 96 64" "192 96 64"; do for c in FETCH_SIZE WRITE_SIZE; do
  timeout 300 rocprofv3 --pmc $c -d $out/pmc_b96_$c -o pmc --output-format csv -- python tools/conv_bwd_bench.py $sh 128 bf16 3 > /dev/null 2>&1
  echo "$sh: $(python tools/pmc_family.py $out/pmc_b96_$c conv_bwd96)"; rm -rf $out/pmc_b96_$c; done; done > $out/b96_hbm.txt 2>&1
( python tools/gate_diag.py tiramisu_multiscale bf16; echo "== layer-wise compose net"; DD_FUSE_COMPOSE=0 DD_FUSE_COMPOSE_BWD=0 python tools/gate_diag.py tiramisu_multiscale bf16 ) 2>&1 | grep -v "amdgpu\|Warning\|detach\|return oracle" > $out/gate_diag.txt
rm -rf $out/prof $out/prof_inf $out/prof_cfg3light $out/prof_cfg3heavy $out/pmc_FETCH_SIZE $out/pmc_WRITE_SIZE $out/pmc_SQ_VALU_MFMA_BUSY_CYCLES
cut -c1-400 $out/bench.json; cat $out/ab_round6.txt
