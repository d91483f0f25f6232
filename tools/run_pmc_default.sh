# FETCH_SIZE / WRITE_SIZE passes over one whole DEFAULT step (carry-over, dedupe, evaluation cache all on; 70 plies):
# what k_search_round moves per launch on the path a default step really runs, to set beside bench.py's bytes_per_launch
# (the other passes, run_profiles.sh, switch the eliminations off so that every network launch is a full-size one)
set -e
TAG=${1:-r05}
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out
cd /tmp
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/${TAG}_pmcd_fetch -- python3 $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 0 --no-cpu-baseline --aux-steps 0 > $O/${TAG}_pmcd_fetch.log 2>&1
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/${TAG}_pmcd_write -- python3 $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 0 --no-cpu-baseline --aux-steps 0 > $O/${TAG}_pmcd_write.log 2>&1
cd $GRAFT_REPO_ROOT
python tools/pmc_summary.py gpurun_out/${TAG}_pmc_default_path.json gpurun_out/${TAG}_pmcd_fetch gpurun_out/${TAG}_pmcd_write > gpurun_out/${TAG}_pmcd_summary.log 2>&1
rm -rf gpurun_out/${TAG}_pmcd_fetch gpurun_out/${TAG}_pmcd_write
grep "k_search_round\|k_play_move\|k_assign" gpurun_out/${TAG}_pmcd_summary.log | cut -c1-400
tail -2 gpurun_out/${TAG}_pmcd_fetch.log | cut -c1-600
