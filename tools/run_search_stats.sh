# kernel stats of one default step (everything on): what k_search_round / k_assign_rows / k_play_move cost per launch
set -e
TAG=${1:-r05}
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out
cd /tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/${TAG}_dstats -- python3 $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 0 --no-cpu-baseline --aux-steps 0 > $O/${TAG}_dstats.log 2>&1
cd $GRAFT_REPO_ROOT
find gpurun_out/${TAG}_dstats -name "*kernel_stats.csv" -exec cp {} gpurun_out/${TAG}_default_path_kernel_stats.csv \;
rm -rf gpurun_out/${TAG}_dstats
head -8 gpurun_out/${TAG}_default_path_kernel_stats.csv | cut -c1-160
