set -x
mkdir -p gpurun_out/r01j
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 1200 python -m pytest tests -m gpu -q 2>&1 | tail -3 > gpurun_out/r01j/tests.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/r01j/smoke.txt 2>&1
timeout 400 python bench.py > gpurun_out/r01j/bench.json 2> gpurun_out/r01j/bench.err
timeout 300 python bench.py --no-cpu-baseline --host-inputs 2>/dev/null | tail -1 > gpurun_out/r01j/host_inputs.txt
timeout 300 python bench.py --mode inference --steps 10 --warmup 2 2>/dev/null | tail -1 > gpurun_out/r01j/inference.json
timeout 600 rocprofv3 --kernel-trace --stats -d gpurun_out/r01j/prof -o r01j -- python bench.py --no-cpu-baseline > gpurun_out/r01j/prof.log 2>&1
python tools/rocprof_summary.py $(ls gpurun_out/r01j/prof/*/*results.db gpurun_out/r01j/prof/*results.db 2>/dev/null | head -1) > gpurun_out/r01j/kernel_stats.txt 2>&1
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 600 rocprofv3 --pmc $c -d gpurun_out/r01j/pmc_$c -o pmc --output-format csv -- python bench.py --no-cpu-baseline --no-graph --steps 2 --warmup 1 > gpurun_out/r01j/pmc_$c.log 2>&1
  python tools/pmc_family.py gpurun_out/r01j/pmc_$c conv_igemm wgrad_dma > gpurun_out/r01j/pmc_$c.txt 2>&1
done
rm -rf gpurun_out/r01j/prof gpurun_out/r01j/pmc_FETCH_SIZE gpurun_out/r01j/pmc_WRITE_SIZE
cat gpurun_out/r01j/tests.txt gpurun_out/r01j/smoke.txt; cut -c1-220 gpurun_out/r01j/bench.json; cat gpurun_out/r01j/pmc_*.txt; head -5 gpurun_out/r01j/kernel_stats.txt
