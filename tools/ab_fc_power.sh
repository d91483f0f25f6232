# what the policy FC's power does to the trunk kernel's clock: bench.py with k_policy_fc (0), k_policy_fc1w (1) and two timing-only
# bodies of k_policy_fc1w - MFMAs without operand DMA (3), operand DMA without MFMAs (4); wrong logits, so the games differ, the
# workload's shape does not.  Probes build of xq_policy.hip.  usage: bash tools/ab_fc_power.sh TAG
TAG=${1:-r04c}
cd $GRAFT_REPO_ROOT
O=gpurun_out
for rep in 1 2 3; do
  for v in 0 1 3 4; do
    timeout -k 10 200 python bench.py --fc-variant $v --no-cpu-baseline --aux-steps 0 --steps 4 > $O/${TAG}_pw_fc${v}_$rep.json 2> $O/${TAG}_pw.err || exit 1
  done
done
python - <<PY
import json
names = {0: "k_policy_fc", 1: "k_policy_fc1w", 3: "k_policy_fc1w, MFMAs only (no operand DMA)", 4: "k_policy_fc1w, operand DMA only (no MFMAs)"}
for rep in (1, 2, 3):
    for v in (0, 1, 3, 4):
        t = open("$O/${TAG}_pw_fc%d_%d.json" % (v, rep)).read()
        d = json.loads(t[t.index('{"metric"'):])
        print("%-46s %.1f ms per step; trunk %.4f of peak at %.3f GHz; mean plies %.1f" % (names[v], d["ms_per_step"], d["roofline"]["frac"], d["roofline"]["clock_ghz"] or 0, d["games"]["mean_plies"]))
PY
