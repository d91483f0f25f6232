set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout -k 10 900 python -m pytest tests -x -q -m gpu > gpurun_out/r02_tests.log 2>&1
timeout -k 10 400 python bench.py > gpurun_out/r02_bench_c3.json 2> gpurun_out/r02_bench.err
O=$GRAFT_REPO_ROOT/gpurun_out
cd /tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/r02_stats -- python3 $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 0 --no-cpu-baseline > $O/r02_stats.log 2>&1
timeout -k 10 200 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/r02_pmc_fetch -- python3 $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 0 --no-cpu-baseline --profile-plies 2 > $O/r02_pmc_fetch.log 2>&1
timeout -k 10 200 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/r02_pmc_write -- python3 $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 0 --no-cpu-baseline --profile-plies 2 > $O/r02_pmc_write.log 2>&1
cd $GRAFT_REPO_ROOT
python tools/pmc_summary.py gpurun_out/r02_pmc_kernels.json gpurun_out/r02_pmc_fetch gpurun_out/r02_pmc_write > gpurun_out/r02_pmc_summary.log 2>&1
find gpurun_out/r02_stats -name "*kernel_stats.csv" -exec cp {} gpurun_out/r02_bench_c3_kernel_stats.csv \;
rm -rf gpurun_out/r02_stats gpurun_out/r02_pmc_fetch gpurun_out/r02_pmc_write
tail -3 gpurun_out/r02_tests.log; cat gpurun_out/r02_pmc_summary.log | head -5
