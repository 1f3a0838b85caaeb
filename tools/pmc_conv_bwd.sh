cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
python tools/conv_bwd_bench.py
DD_FUSE_CONV_BWD=0 python tools/conv_bwd_bench.py
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 300 rocprofv3 --pmc $c -d gpurun_out/pmc_cbw_$c -o pmc --output-format csv -- python tools/conv_bwd_bench.py 64 64 128 128 bf16 3 > /dev/null 2>&1
  python tools/pmc_family.py gpurun_out/pmc_cbw_$c conv_bwd
done
rm -rf gpurun_out/pmc_cbw_*
