set -e
cd $GRAFT_REPO_ROOT
hipcc --offload-arch=gfx950 -O3 -o /tmp/mfma_shape_clock tools/probes/mfma_shape_clock.hip
timeout -k 10 120 /tmp/mfma_shape_clock 0.3 3 > gpurun_out/r05_probe_mfma_shape_clock.txt 2>&1
cat gpurun_out/r05_probe_mfma_shape_clock.txt
O=gpurun_out
R=$GRAFT_REPO_ROOT
export TMPDIR=/tmp
(cd /tmp && timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $R/$O/r05_ktrace -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --aux-steps 0 > $R/$O/r05_ktrace.log 2>&1)
python tools/gap_histogram.py $O/r05_ktrace $O/r05_gap_histogram.txt
rm -rf $O/r05_ktrace
timeout -k 10 200 python bench.py --games 4096 --sims 15 --blocks 4 --steps 2 --warmup 1 --no-cpu-baseline > $O/r05_bench_c2.json 2> $O/r05_bench_c2.err
echo "c2 done"
timeout -k 10 500 python bench.py --sims 200 --blocks 20 --root-noise 0.3,0.25 --temp-cutoff 30 --steps 1 --warmup 0 --aux-steps 1 --no-cpu-baseline > $O/r05_bench_c5_1gpu.json 2> $O/r05_bench_c5.err
echo "c5 done"
timeout -k 10 300 python bench.py --refill 32768 --steps 1 --warmup 0 --no-cpu-baseline > $O/r05_bench_c3_refill.json 2> $O/r05_bench_refill.err
echo "refill done"
timeout -k 10 400 python bench.py > $O/r05_bench_c3.json 2> $O/r05_bench.err
python - <<PY
import json
for f in ("c2", "c5_1gpu", "c3_refill", "c3"):
    t = open("gpurun_out/r05_bench_%s.json" % f).read()
    d = json.loads(t[t.index('{"metric"'):])
    print(f, round(d["value"], 1), "games/s", round(d["roofline"]["frac"], 4), d["roofline"].get("clock_ghz"), {k: round(v, 1) for k, v in d.items() if k.startswith("value_")})
PY
