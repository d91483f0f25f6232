# second SQ counter set for the trunk kernel alone (instruction-class issue time, queue levels, FIFO stalls)
set -e
V=${1:-2}; TAG=${2:-r02}
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out
C1="SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_VMEM_RD SQ_WAVE_CYCLES SQ_INSTS_VMEM"
C2="SQ_IFETCH SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_LDS_DATA_FIFO_FULL SQ_VALU_MFMA_COEXEC_CYCLES SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL"
C3="SQ_INST_LEVEL_LDS SQ_INSTS_LDS"
C4="SQ_INST_LEVEL_VMEM SQ_INSTS_VMEM"
cd /tmp
for G in 512 16384; do
  i=0
  for C in "$C1" "$C2" "$C3" "$C4"; do
    i=$((i+1))
    timeout -k 10 200 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $O/sqx${i}_$G -- python3 $GRAFT_REPO_ROOT/tools/tower_only.py $G 6 $V 5 > $O/sqx$i.log 2>&1 || echo "pass $i failed"
  done
  python3 $GRAFT_REPO_ROOT/tools/pmc_summary.py $O/${TAG}_pmc_sq2_tower_v${V}_G$G.json $O/sqx1_$G $O/sqx2_$G > $O/sqx_summary_$G.log 2>&1 || true
  python3 $GRAFT_REPO_ROOT/tools/pmc_summary.py $O/${TAG}_pmc_sq3_tower_v${V}_G$G.json $O/sqx3_$G $O/sqx4_$G >> $O/sqx_summary_$G.log 2>&1 || true
  rm -rf $O/sqx1_$G $O/sqx2_$G $O/sqx3_$G $O/sqx4_$G
  grep "k_tower" $O/sqx_summary_$G.log | sed 's/_KB_mean_per_launch//g; s/.launches_[A-Z_0-9]*.: 5,//g' | cut -c1-1500
done
