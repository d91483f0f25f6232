# bench.py with k_policy_fc (the default) and with k_policy_fc1w, alternating, on one box.  usage: bash tools/ab_policy_fc.sh TAG
TAG=${1:-r04c}
cd $GRAFT_REPO_ROOT
O=gpurun_out
for rep in 1 2 3 4 5; do
  timeout -k 10 200 python bench.py --fc-variant 0 --no-cpu-baseline --aux-steps 0 --steps 5 > $O/${TAG}_ab_fc0_$rep.json 2> $O/${TAG}_ab.err || exit 1
  timeout -k 10 200 python bench.py --fc-variant 1 --no-cpu-baseline --aux-steps 0 --steps 5 > $O/${TAG}_ab_fc1_$rep.json 2>> $O/${TAG}_ab.err || exit 1
done
python - <<PY
import json
for rep in (1, 2, 3, 4, 5):
    for v in (0, 1):
        t = open("$O/${TAG}_ab_fc%d_%d.json" % (v, rep)).read()
        d = json.loads(t[t.index('{"metric"'):])
        print("policy FC %s: %.1f games/s, %.1f ms per step, trunk %.4f at %.2f GHz" % ("k_policy_fc  " if v == 0 else "k_policy_fc1w", d["value"], d["ms_per_step"], d["roofline"]["frac"], d["roofline"]["clock_ghz"] or 0))
PY
