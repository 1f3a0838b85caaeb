cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02_inf
timeout 600 rocprofv3 --kernel-trace --stats -d gpurun_out/r02_inf/prof -o inf -- python bench.py --mode inference --steps 10 --warmup 2 > gpurun_out/r02_inf/prof.log 2>&1
python tools/rocprof_summary.py $(ls gpurun_out/r02_inf/prof/*/*results.db gpurun_out/r02_inf/prof/*results.db 2>/dev/null | head -1) 30 > gpurun_out/r02_inf/kernel_stats.txt 2>&1
rm -rf gpurun_out/r02_inf/prof
cat gpurun_out/r02_inf/kernel_stats.txt
