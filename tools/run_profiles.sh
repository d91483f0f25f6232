# (the profiled passes run --no-root-eval-carry --no-leaf-dedupe --no-eval-cache: every network launch is then a full-size one)
set -e
TAG=${1:-r03}
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout -k 10 400 python bench.py > gpurun_out/${TAG}_bench_c3.json 2> gpurun_out/${TAG}_bench.err
O=$GRAFT_REPO_ROOT/gpurun_out
cd /tmp
# (profiled passes evaluate every root and every leaf afresh: every launch of the network kernels is then a full-size one, so the per-kernel
# averages and per-launch counter means are not diluted by the empty launches the carried-over roots leave behind)
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/${TAG}_stats -- python3 $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-root-eval-carry --no-leaf-dedupe --no-eval-cache --aux-steps 0 > $O/${TAG}_stats.log 2>&1
timeout -k 10 200 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/${TAG}_pmc_fetch -- python3 $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-root-eval-carry --no-leaf-dedupe --no-eval-cache --aux-steps 0 --profile-plies 2 > $O/${TAG}_pmc_fetch.log 2>&1
timeout -k 10 200 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/${TAG}_pmc_write -- python3 $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-root-eval-carry --no-leaf-dedupe --no-eval-cache --aux-steps 0 --profile-plies 2 > $O/${TAG}_pmc_write.log 2>&1
cd $GRAFT_REPO_ROOT
python tools/pmc_summary.py gpurun_out/${TAG}_pmc_kernels.json gpurun_out/${TAG}_pmc_fetch gpurun_out/${TAG}_pmc_write > gpurun_out/${TAG}_pmc_summary.log 2>&1
find gpurun_out/${TAG}_stats -name "*kernel_stats.csv" -exec cp {} gpurun_out/${TAG}_bench_c3_kernel_stats.csv \;
rm -rf gpurun_out/${TAG}_stats gpurun_out/${TAG}_pmc_fetch gpurun_out/${TAG}_pmc_write
cat gpurun_out/${TAG}_pmc_summary.log | head -8
