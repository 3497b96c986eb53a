# Cache behaviour of the gather kernels (flow_warp*, warp_blend, resize): are the bilinear taps served by L1 / L2 or by HBM?
# Separate rocprofv3 --pmc passes over one forward each; summed per kernel by tools/pmc_lds.py.
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
F="--steps 1 --warmup 1 --no-cpu-baseline --no-profile --no-host-io --no-configs"
rm -rf gpurun_out/cache1 gpurun_out/cache2 gpurun_out/cache3 gpurun_out/cache4
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum -d gpurun_out/cache1 --output-format csv -- python3 bench.py $F > gpurun_out/cache1.log 2>&1
rocprofv3 --pmc TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum -d gpurun_out/cache2 --output-format csv -- python3 bench.py $F > gpurun_out/cache2.log 2>&1
rocprofv3 --pmc FETCH_SIZE -d gpurun_out/cache3 --output-format csv -- python3 bench.py $F > gpurun_out/cache3.log 2>&1
rocprofv3 --pmc GRBM_GUI_ACTIVE -d gpurun_out/cache4 --output-format csv -- python3 bench.py $F > gpurun_out/cache4.log 2>&1
python tools/pmc_lds.py gpurun_out/cache1 gpurun_out/cache2 gpurun_out/cache3 gpurun_out/cache4 > gpurun_out/cache_summary.txt 2>&1
rm -rf gpurun_out/cache1 gpurun_out/cache2 gpurun_out/cache3 gpurun_out/cache4
