# SQ counter pass over the bench workload (2 plies): MFMA-pipe busy share, LDS conflicts, wait cycles
set -e
TAG=${1:-r03}
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out
cd /tmp
rocprofv3 -L > $O/rocprof_counters.txt 2>&1 || true
C1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVES"
C2="SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM SQ_WAIT_INST_LDS SQ_INSTS_MFMA"
timeout -k 10 200 rocprofv3 --pmc $C1 --kernel-trace --output-format csv -d $O/sq1 -- python3 $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-root-eval-carry --no-leaf-dedupe --no-eval-cache --aux-steps 0 --profile-plies 2 > $O/sq1.log 2>&1 || echo "pass 1 failed"
timeout -k 10 200 rocprofv3 --pmc $C2 --kernel-trace --output-format csv -d $O/sq2 -- python3 $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-root-eval-carry --no-leaf-dedupe --no-eval-cache --aux-steps 0 --profile-plies 2 > $O/sq2.log 2>&1 || echo "pass 2 failed"
cd $GRAFT_REPO_ROOT
python tools/pmc_summary.py gpurun_out/${TAG}_pmc_sq.json gpurun_out/sq1 gpurun_out/sq2 > gpurun_out/sq_summary.log 2>&1 || true
rm -rf gpurun_out/sq1 gpurun_out/sq2
grep "k_tower\|k_search_round" gpurun_out/sq_summary.log | cut -c1-900
tail -3 gpurun_out/sq1.log; tail -3 gpurun_out/sq2.log
