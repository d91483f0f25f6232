set -e
cd $GRAFT_REPO_ROOT
timeout -k 10 600 python -m pytest tests/test_gpu_round5.py -x -q -m gpu 2>&1 | tail -3
bash tools/run_verify_soak.sh | tail -30
timeout -k 10 200 python bench.py --games 4096 --sims 15 --blocks 4 --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/r05i_bench_c2.json 2> gpurun_out/r05i_bench_c2.err
timeout -k 10 400 python bench.py --no-cpu-baseline > gpurun_out/r05i_bench_c3.json 2> gpurun_out/r05i_bench.err
bash tools/run_search_stats.sh r05i
python - <<PY
import json
for f in ("c2", "c3"):
    t = open("gpurun_out/r05i_bench_%s.json" % f).read()
    d = json.loads(t[t.index('{"metric"'):])
    print(f, round(d["value"], 1), "games/s", round(d["roofline"]["frac"], 4), d["roofline"].get("clock_ghz"), d["roofline_tree"]["kernel"], {k: round(v, 1) for k, v in d.items() if k.startswith("value_")})
PY
