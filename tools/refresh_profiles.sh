#!/bin/bash
# One gpurun call that refreshes every measured artefact of a state:  tools/refresh_profiles.sh <tag>   (e.g. r02_a)
# Writes gpurun_out/<tag>/...; copy what is to be judged into profiles/ (tools/collect_profiles.py <tag>).
# Trace and counter passes are separate rocprofv3 runs (never --pmc together with a trace option).
set -x
tag=${1:-r03_x}
out=gpurun_out/$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $out/smoke.txt 2>&1
timeout 600 python bench.py --dump-launches $out/conv_launches.json > $out/bench.json 2> $out/bench.err
timeout 300 python bench.py --no-cpu-baseline --no-extras --host-inputs 2>/dev/null | tail -1 > $out/host_inputs.txt
timeout 300 python bench.py --mode inference --steps 10 --warmup 2 2>/dev/null | tail -1 > $out/inference.json
timeout 600 rocprofv3 --kernel-trace --stats -d $out/prof -o $tag -- python bench.py --no-cpu-baseline --no-extras > $out/prof.log 2>&1
python tools/rocprof_summary.py $(ls $out/prof/*/*results.db $out/prof/*results.db 2>/dev/null | head -1) 60 > $out/kernel_stats.txt 2>&1
# the other half of BASELINE's metric (cfg-5 inference, fp16) and BASELINE config 3 (heavy Tiramisu) under the same profiler
timeout 600 rocprofv3 --kernel-trace --stats -d $out/prof_inf -o inf -- python bench.py --mode inference --steps 10 --warmup 2 > $out/prof_inf.log 2>&1
python tools/rocprof_summary.py $(ls $out/prof_inf/*/*results.db $out/prof_inf/*results.db 2>/dev/null | head -1) 40 > $out/inference_kernel_stats.txt 2>&1
timeout 600 rocprofv3 --kernel-trace --stats -d $out/prof_cfg3 -o cfg3 -- python tools/cfg3_step.py heavy 8 5 > $out/prof_cfg3.log 2>&1
python tools/rocprof_summary.py $(ls $out/prof_cfg3/*/*results.db $out/prof_cfg3/*results.db 2>/dev/null | head -1) 40 > $out/cfg3_heavy_kernel_stats.txt 2>&1
grep "^cfg-3" $out/prof_cfg3.log >> $out/cfg3_heavy_kernel_stats.txt
timeout 600 rocprofv3 --kernel-trace --stats -d $out/prof_cfg3l -o cfg3l -- python tools/cfg3_step.py light 8 5 > $out/prof_cfg3l.log 2>&1
python tools/rocprof_summary.py $(ls $out/prof_cfg3l/*/*results.db $out/prof_cfg3l/*results.db 2>/dev/null | head -1) 40 > $out/cfg3_light_kernel_stats.txt 2>&1
grep "^cfg-3" $out/prof_cfg3l.log >> $out/cfg3_light_kernel_stats.txt
for c in FETCH_SIZE WRITE_SIZE "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 GRBM_GUI_ACTIVE"; do
  n=$(echo $c | cut -d' ' -f1)
  timeout 600 rocprofv3 --pmc $c -d $out/pmc_$n -o pmc --output-format csv -- python bench.py --no-cpu-baseline --no-extras --no-graph --steps 2 --warmup 1 > $out/pmc_$n.log 2>&1
  python tools/pmc_family.py $out/pmc_$n conv_igemm conv_rw conv_ks conv_pw wgrad_pw wgrad_dma conv_bwd convt_fwd convt_bwd compose_stream_fwd compose_stream_bwd compose_stream_wgrad head_fwd head_bwd maxpool assemble_input > $out/pmc_$n.txt 2>&1
done
# BASELINE config 3 from the counters: HBM traffic and matrix-pipe busy of both Tiramisu configurations, per kernel family
for cfg in light heavy; do
  for c in FETCH_SIZE WRITE_SIZE "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 GRBM_GUI_ACTIVE"; do
    n=$(echo $c | cut -d' ' -f1)
    timeout 600 rocprofv3 --pmc $c -d $out/pmc3_${cfg}_$n -o pmc --output-format csv -- python tools/cfg3_step.py $cfg 8 2 > $out/pmc3_${cfg}_$n.log 2>&1
    python tools/pmc_family.py $out/pmc3_${cfg}_$n conv_ks conv_pw wgrad_pw wgrad_dma conv_igemm conv_rw conv_bwd compose_stream kpcn masked_add colsum maxpool > $out/cfg3_${cfg}_pmc_$n.txt 2>&1
    rm -rf $out/pmc3_${cfg}_$n
  done
done
# the multi-process path of the bench on ONE device (two ranks, gloo transport; RCCL needs >= 2 GPUs): same code above the transport
HSA_ENABLE_IPC_MODE_LEGACY=0 DD_FORCE_DEVICE=0 DD_DIST_BACKEND=gloo timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 2 --steps 5 --warmup 3 --batch 32 > $out/two_ranks_one_gpu.txt 2> $out/two_ranks_one_gpu.err
# one rank over RCCL (the nccl backend): communicator, side-stream all-reduces, barriers -- functional record, not a scaling number
HSA_ENABLE_IPC_MODE_LEGACY=0 DD_FORCE_COLLECTIVES=1 timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29519 bench.py --gpus 1 --steps 20 --warmup 5 --no-extras --no-cpu-baseline > $out/one_rank_rccl.txt 2> $out/one_rank_rccl.err
# round 5: the microbenchmarks and knock-outs DESIGN section 7 cites (tools/exp/* are built on the CPU side before the call)
[ -x tools/exp/rw_loop_ubench ] && tools/exp/rw_loop_ubench > $out/rw_loop_ubench.txt 2>&1
[ -x tools/exp/lds_probe ] && tools/exp/lds_probe > $out/lds_probe.txt 2>&1
tools/pmc_sq_table.sh > /dev/null 2>&1; cp gpurun_out/pmc_sq_table.txt $out/sq_table.txt
( for v in hip nodma0 nowrole noflush0; do echo "== $v"; if [ $v = hip ]; then python tools/wgrad_only_bench.py 2>/dev/null; elif [ -f tools/exp/libdd_$v.so ]; then DD_LIB=tools/exp/libdd_$v.so python tools/wgrad_only_bench.py 2>/dev/null; fi; done ) > $out/wgrad_only_knockouts.txt 2>&1
( DD_DETERMINISTIC=1 python tools/det_diag.py cfg2; python tools/det_diag.py cfg2 ) > $out/deterministic.txt 2>&1
# round 5, second half: the literal ArchitectureExample.json step by family, the A/B pairs DESIGN 7.5 cites (one process per variant, same box)
python tools/example_profile.py 8 > $out/example_profile.txt 2>&1
( timeout 600 python tools/ab_bench.py "default:" "loss_unfused:DD_FUSE_LOSS_INVERT=0,DD_LOSS_SIMPLE=0,DD_LOSS_GENERAL=0" --repeat 2;
  timeout 600 python tools/ab_bench.py "frames_in_place:" "extract_tiles:DD_FRAME_INPUT=0" --inference --repeat 2 ) > $out/ab_round5.txt 2>&1
[ -f tools/exp/libdd_natpitch.so ] && ( timeout 400 python tools/ab_bench.py "padded_strips:" "natural_pitch:DD_LIB=tools/exp/libdd_natpitch.so" --repeat 2 ) >> $out/ab_round5.txt 2>&1
# round 5, last part: the issue-level counter rows of the cfg-3 kernels (conv_ks, stacked weight gradients) and the per-layer times of the Tiramisu shapes
SQT_CMD="python tools/cfg3_step.py heavy 8 2" SQT_OUT=$out/sq_table_cfg3_heavy.txt bash tools/pmc_sq_table.sh conv_ks wgrad conv_igemm_ws conv_pw conv_rw > /dev/null 2>&1
SQT_CMD="python tools/cfg3_step.py light 8 2" SQT_OUT=$out/sq_table_cfg3_light.txt bash tools/pmc_sq_table.sh conv_ks wgrad conv_igemm_ws conv_pw conv_rw > /dev/null 2>&1
( python tools/ks_shape_bench.py; echo "== DD_CONV_KS_THIN=1"; DD_CONV_KS_THIN=1 python tools/ks_shape_bench.py ) 2>&1 | grep -v amdgpu.ids > $out/ks_shape_bench.txt
python tools/cfg3_launches.py light 8 > $out/cfg3_light_launches.txt 2>&1
python tools/cfg3_launches.py heavy 8 > $out/cfg3_heavy_launches.txt 2>&1
rm -rf $out/prof $out/prof_inf $out/prof_cfg3 $out/prof_cfg3l $out/pmc_FETCH_SIZE $out/pmc_WRITE_SIZE $out/pmc_SQ_VALU_MFMA_BUSY_CYCLES
cat $out/smoke.txt | tail -2; cut -c1-300 $out/bench.json; cat $out/pmc_*.txt; head -12 $out/kernel_stats.txt
