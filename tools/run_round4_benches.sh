# the auxiliary bench lines of a round: C2, C5 on one GPU, refill, the driver's 20-step form, and the
# kernel stats of the default path (everything on).  usage: bash tools/run_round4_benches.sh TAG
set -e
TAG=${1:-r04}
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out
timeout -k 10 200 python bench.py --games 4096 --sims 15 --blocks 4 --steps 2 --warmup 1 --no-cpu-baseline > $O/${TAG}_bench_c2.json 2> $O/${TAG}_bench_c2.err
echo "c2 done"
timeout -k 10 500 python bench.py --sims 200 --blocks 20 --root-noise 0.3,0.25 --temp-cutoff 30 --steps 1 --warmup 0 --aux-steps 1 --no-cpu-baseline > $O/${TAG}_bench_c5_1gpu.json 2> $O/${TAG}_bench_c5.err
echo "c5 done"
timeout -k 10 300 python bench.py --refill 32768 --steps 1 --warmup 0 --no-cpu-baseline > $O/${TAG}_bench_c3_refill.json 2> $O/${TAG}_bench_refill.err
echo "refill done"
timeout -k 10 400 python bench.py --steps 20 --warmup 5 > $O/${TAG}_bench_c3_steps20.json 2> $O/${TAG}_bench_steps20.err
echo "steps20 done"
cd /tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/${TAG}_dstats -- python3 $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 0 --no-cpu-baseline --aux-steps 0 > $O/${TAG}_dstats.log 2>&1
cd $GRAFT_REPO_ROOT
find gpurun_out/${TAG}_dstats -name "*kernel_stats.csv" -exec cp {} gpurun_out/${TAG}_default_path_kernel_stats.csv \;
rm -rf gpurun_out/${TAG}_dstats
python - <<PY
import json
for f in ("c2", "c5_1gpu", "c3_refill", "c3_steps20"):
    t = open("gpurun_out/${TAG}_bench_%s.json" % f).read()
    d = json.loads(t[t.index('{"metric"'):])
    print(f, round(d["value"], 1), "games/s", round(d["roofline"]["frac"], 4), d["roofline"].get("clock_ghz"))
PY
