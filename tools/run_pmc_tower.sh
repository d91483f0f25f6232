# SQ counter passes over the trunk kernel alone: G=512 (one workgroup per CU: a wave has its SIMD to itself)
# and G=16384 (bench size).  usage: bash tools/run_pmc_tower.sh VARIANT TAG
set -e
V=${1:-2}; TAG=${2:-r02}
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out
C1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVES"
C2="SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM SQ_WAIT_INST_LDS SQ_INSTS_MFMA"
cd /tmp
for G in 512 16384; do
  timeout -k 10 200 rocprofv3 --pmc $C1 --kernel-trace --output-format csv -d $O/sqt1_$G -- python3 $GRAFT_REPO_ROOT/tools/tower_only.py $G 6 $V 5 > $O/sqt1.log 2>&1 || echo "pass 1 failed"
  timeout -k 10 200 rocprofv3 --pmc $C2 --kernel-trace --output-format csv -d $O/sqt2_$G -- python3 $GRAFT_REPO_ROOT/tools/tower_only.py $G 6 $V 5 > $O/sqt2.log 2>&1 || echo "pass 2 failed"
  python3 $GRAFT_REPO_ROOT/tools/pmc_summary.py $O/${TAG}_pmc_sq_tower_v${V}_G$G.json $O/sqt1_$G $O/sqt2_$G > $O/sqt_summary_$G.log 2>&1 || true
  rm -rf $O/sqt1_$G $O/sqt2_$G
  grep "k_tower" $O/sqt_summary_$G.log | cut -c1-1200
done
