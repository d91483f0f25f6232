"""`python bench.py --gpus N` must start its N ranks itself: the launcher, the
world-size checks and the N-rank control flow (sharded seeds, production all-gather, barrier,
max-over-ranks timing, one JSON line from rank 0) run here on CPU over gloo with a stand-in engine."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def _env(**kw):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(OMP_NUM_THREADS="1", PYTHONPATH=ROOT + os.pathsep + env.get("PYTHONPATH", ""))
    env.update(kw)
    return env


def test_bare_bench_gpus_2_launches_two_ranks():
    p = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--steps", "2", "--warmup", "1", "--games", "3", "--sims", "16"],
                       env=_env(XQ_BENCH_BACKEND="gloo", XQ_BENCH_STANDIN="tests.bench_standin"),
                       capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-3000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, p.stdout                       # rank 0 only
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["steps"] == 2 and out["warmup"] == 1 and out["scaling"] == "weak"
    assert out["gathered_games"] == 6                      # 3 games per rank, both shards in the gathered tensor
    assert out["value"] > 0 and abs(out["value"] - 6 * 2 / (out["ms_per_step"] * 2e-3)) < 1e-6 * out["value"]
    assert "not a benchmark" in out["data"]


def test_rank_refuses_a_world_size_that_differs_from_gpus():
    p = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--steps", "1", "--warmup", "0"],
                       env=_env(WORLD_SIZE="1", RANK="0", LOCAL_RANK="0", XQ_BENCH_BACKEND="gloo"),
                       capture_output=True, text=True, timeout=120, cwd=ROOT)
    assert p.returncode != 0 and "--gpus 2 but WORLD_SIZE=1" in p.stderr


def test_launcher_decision_happens_before_any_gpu_library_is_touched():
    """Everything above the launch decision must be import-light: no torch, no ctypes library."""
    src = open(BENCH).read()
    head = src[:src.index("def launch_ranks")]
    main_body = head[head.index("def main():"):]
    assert "import torch" not in main_body and "_lib.lib()" not in main_body
    top = src[:src.index("def net_flops_per_row")]
    assert "import torch" not in top and "chinesechessai_amd" not in top
