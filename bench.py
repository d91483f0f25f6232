#!/usr/bin/env python3
"""bench.py — self-play games/sec at fixed MCTS sims (BASELINE.json metric) on N MI355X.

    python bench.py --gpus N --steps K --warmup W

N > 1 needs one rank per GPU (RCCL).  Either the caller starts the ranks (`python -m
torch.distributed.run --nproc-per-node N ... bench.py --gpus N`: WORLD_SIZE is set, bench.py is a
rank) or nobody did (bare `python bench.py --gpus N`): then bench.py starts them itself as a CHILD
process (torch.distributed.run; never exec) before anything touches the GPU, relays rank 0's JSON
line and exits with the child's code — the counterpart of the reference's pool start
(self_play.py:404-408).  A rank whose WORLD_SIZE differs from --gpus refuses to run.

A "step" = one pass of the hot path over one batch: G concurrent games per GPU played from the
start position to the end (rules -> MCTS -> network leaf evaluation -> (state, pi, z) samples),
plus — for N > 1 — the epoch-end all-gather of the sample records (started when the step's last game
is packed, waited for inside the timed region, running beside the next step's play).  Workload at N=1 =
BASELINE.json configs[2], the configuration the metric is quoted on (S = 50):
16,384 concurrent games, 50 sims, 6-block ResNet bf16, random-init weights, synthetic start
positions, per-game seeds base+g.  Games shard across GPUs with no data-path collective
(weak scaling: G per GPU fixed).

The headline runs with the result-identical root evaluation carry-over (the new root's priors are the ones the
played child received during the previous ply's search: 6 network forwards per ply instead of the reference's 7,
games bit-identical, tests/test_gpu_round3.py) and evaluator row compaction (slots without a pending leaf are not
network rows); `value_no_carry` (every root evaluated afresh, as the reference does) and
`value_full_policy_head` (all 8,100 policy columns) are measured beside it on a few extra steps.

Prints ONE JSON line (rank 0).  `roofline` = the dominant kernel of the step, the hand-written
single-launch trunk k_tower1wa (csrc/xq_tower1wa.hpp; MFMA-bound, ~80 % of the step), over its full-size
launches (rows = games; the carried-over rounds launch it with zero rows and are listed apart); `roofline_net` =
the whole network forward over all launches and the rows they really evaluated; `roofline_tree` = the tree/rules
kernel k_search_round (HBM-bound integer work); all measured live with events on the stream the kernels run on.
`cpu_baseline` = the CPU oracle ("port" of the reference algorithm, net on CPU torch) timed on the
host cores on a bounded sample (rank 0, N = 1 only).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import numpy as np

MFMA_PEAK_BF16_TFLOPS = 2500.0     # dense bf16, /opt/skills/guides/MI355X_MICROARCH.md
HBM_PEAK_GBS = 8000.0              # HBM3E spec, same guide (6.3 TB/s achievable)


def net_flops_per_row(blocks):
    """neural_network.py:25-45 shapes: conv1 + 2 convs per block + heads."""
    conv1 = 2 * 15 * 128 * 9 * 90
    res = 2 * 128 * 128 * 9 * 90
    heads = 2 * 128 * 32 * 90 + 2 * 2880 * 8100 + 2 * 128 * 8 * 90 + 2 * 720 * 128 + 2 * 128
    return conv1 + 2 * blocks * res + heads


def tree_bytes_per_descent(b=33.3, lvl=0.86):
    """Algorithmic HBM bytes of one descent in THIS layout: root board + scalars
    (64), PUCT reads 16 B/child/level, leaf move list 2b, packed leaf board 48, bf16 NHWC16 planes
    2880, legal logits gather 2b+2, move list re-read 2b, edge init 22b, backup RMW 24(l+1).
    b and l are the reference's measured means."""
    return 64 + 16 * b * lvl + 2 * b + 48 + 2880 + (2 * b + 2) + 2 * b + 22 * b + 24 * (lvl + 1)


def lib_sha16():
    """sha256 (16 hex digits) of the HIP library this process runs: what a PMC summary names as the build it profiled"""
    import hashlib
    path = os.path.join(ROOT, "chinesechessai_amd", "csrc", "libxq_hip.so")
    return hashlib.sha256(open(path, "rb").read()).hexdigest()[:16] if os.path.exists(path) else None


def pmc_traffic(kernel, G, S, blocks, fetch_factor=1.0):
    """(HBM bytes per launch of `kernel`, where that number comes from).  The PMC counters cannot be read
    from inside this process: the bytes come from the newest committed rocprofv3 pass over this same
    workload (profiles/*_pmc_kernels.json: separate --pmc FETCH_SIZE / WRITE_SIZE runs, KB per launch,
    tools/run_profiles.sh), and the source file is named next to the number.
    fetch_factor = 2 for kernels that read with 16 B/lane coalesced accesses (gfx950 FETCH_SIZE
    correction, MI355X_MICROARCH.md); (None, None) for configs that were not profiled."""
    if not (G == 16384 and S == 50 and blocks == 6):
        return None, None
    for name in ("r05_pmc_kernels.json", "r05a_pmc_kernels.json", "r04_pmc_kernels.json", "r03_pmc_kernels.json", "r02c_pmc_kernels.json", "r02b_pmc_kernels.json", "r02_pmc_kernels.json", "r01g_pmc_kernels.json", "r01f_pmc_kernels.json", "r01d_pmc_kernels.json"):
        path = os.path.join(ROOT, "profiles", name)
        if os.path.exists(path):
            doc = json.load(open(path))
            k = doc["kernels"].get(kernel)
            if k and "FETCH_SIZE_KB_mean_per_launch" in k and "WRITE_SIZE_KB_mean_per_launch" in k:
                byt = (fetch_factor * k["FETCH_SIZE_KB_mean_per_launch"] + k["WRITE_SIZE_KB_mean_per_launch"]) * 1024.0
                return byt, "profiles/%s (committed rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this workload%s; build of those passes: %s; not measured in this run)" % (
                    name, ", " + doc["note"] if doc.get("note") else "",
                    "libxq_hip.so sha16 " + doc["build_libxq_hip_sha16"] if doc.get("build_libxq_hip_sha16") else "not recorded")
    return None, None


# ------------------------------------------------------------------------------------------
# CPU baseline: the oracle (C restatement of the reference algorithm) with the net on CPU torch
# ------------------------------------------------------------------------------------------
def _cpu_worker(args):
    idx, blocks, sims, threads, rep_plies, games, barrier = args
    import torch
    torch.set_num_threads(threads)
    from chinesechessai_amd.neural_network import ChessNet
    from chinesechessai_amd.chess_env import decode_move
    from oracle import xq_oracle as xo
    torch.manual_seed(0)
    net = ChessNet(num_blocks=blocks).eval()

    def fn(ctx, nrows, boards, players, moves, nmoves, priors, values):
        rows = []
        for i in range(nrows):
            b = np.array([boards[i * 90 + k] for k in range(90)], dtype=np.int8).reshape(10, 9)
            mv = [decode_move(moves[i * 128 + j]) for j in range(nmoves[i])]
            rows.append((b, int(players[i]), mv))
        res = net.predict_batch(rows)             # rows keep the reference's multiplicity (<= 8 duplicates)
        for i, (d, v) in enumerate(res):
            for j, p in enumerate(d.values()):
                priors[i * 128 + j] = float(p)
            values[i] = float(v)
        return 0

    cb = xo.EVAL_FN(fn)
    ev = xo.Evaluator(cb, None)
    xo.lib()
    xo.self_play_game(999, sims, eval_red=ev, max_moves=1)          # first touch (untimed)
    out = []
    for rep, plies in enumerate(rep_plies):
        barrier.wait()                                               # pool start-up is outside every timed span
        t0 = time.time()
        n_plies = 0
        for k in range(games):
            rc, g = xo.self_play_game(1000 + (rep * 64 + idx) * 64 + k, sims, eval_red=ev, max_moves=plies)
            if rc == 0:
                n_plies += g.n_plies
        out.append((time.time() - t0, n_plies))
    return out


def cpu_baseline(blocks, sims, workers=4, plies=18, games=8, reps=3, max_cores=16, whole_game_reps=1):
    """The structural twin of the reference's 4-process path (NUM_WORKERS = 4, config.py:48;
    self_play.py:404-408): 4 worker processes, each playing `games` games one after the other with a
    private net replica on CPU torch, intra-op threads pinned to cores/4 (the reference's unpinned
    default oversubscribes).  Timing rule of SURVEY.md §8(d): >= 8 games per worker,
    pool start-up and first touch excluded, median of 3 repetitions.  Bounded sample: the first
    `plies` plies of every game, scaled to games/s by plies/70 (random-init games run to the 70-ply
    cap at a near-constant cost per ply) - and `whole_game_reps` of the repetitions play WHOLE 70-ply
    games, so the extrapolation is checked inside the same run (`whole_games` in the result);
    --cpu-baseline-full plays whole games in every repetition."""
    import multiprocessing as mp
    cores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 4)
    cores = min(cores, max_cores)                 # the box's CPU share for one GPU
    threads = max(1, cores // workers)
    rep_plies = [plies] * reps
    for k in range(min(whole_game_reps, reps)):
        rep_plies[reps - 1 - k] = 70
    ctx = mp.get_context("spawn")
    mgr = ctx.Manager()
    barrier = mgr.Barrier(workers)
    with ctx.Pool(workers) as pool:
        res = pool.map(_cpu_worker, [(i, blocks, sims, threads, rep_plies, games, barrier) for i in range(workers)])
    rates, walls = [], []
    for rep in range(reps):
        wall = max(r[rep][0] for r in res)
        rates.append(sum(r[rep][1] for r in res) / 70.0 / wall)
        walls.append(wall)
    whole = [rates[i] for i in range(reps) if rep_plies[i] == 70]
    return {"value": float(np.median(rates)), "unit": "games/s", "cores": workers * threads, "kind": "port",
            "whole_games": float(np.median(whole)) if whole else None,
            "sample": "%d worker processes x %d games each (oracle C rules+MCTS, %d sims, %d-block net fp32 on CPU torch, %d "
                      "threads each, duplicate leaf rows evaluated as the reference does); %d repetitions, plies played per "
                      "game in each: %s (a repetition of fewer than 70 plies is scaled by plies/70; the 70-ply ones are whole "
                      "games and check that scaling); value = median (%s games/s; walls %s s), pool start-up and first touch "
                      "excluded" % (workers, games, sims, blocks, threads, reps, ", ".join(str(v) for v in rep_plies),
                                    ", ".join("%.3f" % v for v in rates), ", ".join("%.1f" % v for v in walls))}


# ------------------------------------------------------------------------------------------
def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--games", type=int, default=16384, help="concurrent games per GPU")
    ap.add_argument("--sims", type=int, default=50)
    ap.add_argument("--blocks", type=int, default=6)
    ap.add_argument("--warmup-plies", type=int, default=0, help="0 = full warmup steps; >0 shortens a warmup step to this many plies")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-baseline-full", action="store_true", help="cpu_baseline on whole 70-ply games (8 per worker, median of 3: ~100 s)")
    ap.add_argument("--refill", type=int, default=0, metavar="TOTAL",
                    help="steady-state mode (reported beside the headline, never instead of it): every step plays TOTAL games "
                         "per GPU through the --games concurrent slots, a finished game's slot being refilled with the next seed")
    ap.add_argument("--profile-plies", type=int, default=0,
                    help="profiling aid only: stop every step after this many plies (the JSON line is then NOT a benchmark)")
    ap.add_argument("--event-stride", type=int, default=4,
                    help="HIP events around every N-th network forward / trunk launch / k_search_round of the timed region (1 = all)")
    ap.add_argument("--no-eval-cache", action="store_true",
                    help="switch the evaluation cache (position -> priors + value, kept two plies) off")
    ap.add_argument("--no-leaf-dedupe", action="store_true",
                    help="every pending leaf gets its own network row (default: equal positions of a round share one; "
                         "value_no_dedupe reports this form beside the headline)")
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "f32"])
    ap.add_argument("--net-chunk", type=int, default=0, help="rows per network launch (0 = all games at once)")
    ap.add_argument("--root-noise", default="", help="extension (BASELINE C5): 'alpha,eps' Dirichlet root noise, e.g. 0.3,0.25")
    ap.add_argument("--temp-cutoff", type=int, default=0, help="extension (C5): temperature 1.0 before this ply, ~0 after")
    ap.add_argument("--search-occ", type=int, default=0, help="diagnostic: k_search_round waves/SIMD variant (3, 5, 6, 8)")
    ap.add_argument("--policy-columns", default="reachable", choices=["all", "reachable"],
                    help="'reachable' (default): the policy FC computes the 2,294 of 8,100 columns a legal move can index - the "
                         "search gathers legal-move logits only (neural_network.py:148-169), the rest are dead outputs; "
                         "'all': the reference's full 8,100-column head")
    ap.add_argument("--no-root-eval-carry", action="store_true",
                    help="evaluate every root afresh like the reference (7 forwards per ply).  Default: the result-identical root "
                         "evaluation carry-over is on (the played child's network evaluation becomes the next root's, 6 forwards "
                         "per ply); value_no_carry in the JSON line is this mode, measured beside the headline")
    ap.add_argument("--aux-steps", type=int, default=2,
                    help="steps for each of the figures measured beside the headline at N = 1 (value_no_carry, "
                         "value_full_policy_head); 0 = skip them")
    ap.add_argument("--tree-reuse", action="store_true", help="extension: keep the played move's subtree (no reference oracle)")
    ap.add_argument("--virtual-loss", action="store_true", help="extension: up to 8 distinct leaves per game and round (8x the network rows)")
    ap.add_argument("--fused-tower", type=int, default=1, help="1 = whole trunk in one launch (k_tower), 0 = one launch per convolution")
    ap.add_argument("--tower-variant", type=int, default=-1, help="diagnostic: trunk kernel build (-1 = library default: 60 = k_tower1wa from 2,048 boards up, 36 below; 36 / 39 = k_tower16b with 2 / 4 boards per workgroup; 0 = 32x32x16)")
    ap.add_argument("--fc-variant", type=int, default=-1, help="diagnostic: policy FC kernel (-1 = library default = 1: k_policy_fc1w, one wave per SIMD with a generated asm body; 0 = k_policy_fc, 8 waves, HIP)")
    ap.add_argument("--conv-variant", type=int, default=0, help="diagnostic: 1 = 2 boards/WG, 2 = 4 boards/WG (0 = library default)")
    args = ap.parse_args()
    if args.gpus < 1:
        ap.error("--gpus must be >= 1")
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        sys.exit(launch_ranks(args.gpus))              # before anything initialises the GPU
    run_rank(args)


def launch_ranks(n):
    """`python bench.py --gpus N` with no launcher around it: start the N ranks as a child
    torch.distributed.run (one process per GPU, rendezvous on 127.0.0.1), pass its output through
    and return its exit code.  The child is a subprocess, never an exec; this process has not
    touched the GPU and never will.  SIGTERM / SIGINT to this process are passed on to the child's
    whole process group (the ranks must not outlive the launcher and keep the GPUs), and the child is
    torn down on any exception here."""
    import signal
    import socket
    import subprocess
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "4")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    child = subprocess.Popen(cmd, env=env, start_new_session=True)      # own process group: one signal reaches every rank

    def stop(signum):
        try:
            os.killpg(child.pid, signum)
        except (ProcessLookupError, PermissionError):
            pass

    def on_signal(signum, frame):
        stop(signum)
        try:
            child.wait(timeout=20)
        except subprocess.TimeoutExpired:
            stop(signal.SIGKILL)
            child.wait()
        sys.exit(128 + signum)

    old = {sg: signal.signal(sg, on_signal) for sg in (signal.SIGTERM, signal.SIGINT)}
    try:
        return child.wait()
    except BaseException:
        stop(signal.SIGTERM)
        try:
            child.wait(timeout=20)
        except subprocess.TimeoutExpired:
            stop(signal.SIGKILL)
            child.wait()
        raise
    finally:
        for sg, h in old.items():
            signal.signal(sg, h)


def run_rank_rehearsal(args, backend, rank, world):
    """See run_rank: N-rank control flow on CPU (gloo) with a test-provided stand-in engine."""
    import importlib
    import torch
    import torch.distributed as dist
    from chinesechessai_amd import distributed as xd
    standin = os.environ.get("XQ_BENCH_STANDIN")
    if not standin:
        sys.exit("bench.py: XQ_BENCH_BACKEND=%s needs XQ_BENCH_STANDIN=<module with make_step()> (tests only); "
                 "the HIP engine has no CPU fallback" % backend)
    dist.init_process_group(backend)
    assert dist.get_world_size() == args.gpus == world and dist.get_rank() == rank
    G, S = args.games, args.sims
    play = importlib.import_module(standin).make_step(G, S)       # seeds -> uint8 CPU tensor [G * 70 * RECORD_BYTES]
    gathered = [None]
    pipe = xd.RecordGather(G * 70 * xd.RECORD_BYTES, "cpu")     # the same overlapped gather as run_rank

    def step(base_seed):
        local = play(xd.game_seeds(base_seed, G * world, rank, world))
        pipe.next_buffer().copy_(local)
        pipe.launch()

    for w in range(args.warmup):
        step(7_000_000 + w * G * world)
    pipe.drain()
    dist.barrier()
    t0 = time.time()
    for k in range(args.steps):
        step(k * G * world)
    gathered[0] = pipe.drain()
    dist.barrier()
    t = torch.tensor([time.time() - t0], dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    dt = float(t.item())
    if rank == 0:
        rec = xd.records_to_numpy(gathered[0]).reshape(world, G, 70)
        print(json.dumps({
            "metric": "self-play games/sec @ %d MCTS sims" % S, "value": G * world * args.steps / dt, "unit": "games/s",
            "n_gpus": dist.get_world_size(), "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "n/a",
            "data": "REHEARSAL of the N-rank control flow on CPU (%s, stand-in engine %s) - not a benchmark" % (backend, standin),
            "config": {"workload": "%d games per rank, %d sims" % (G, S), "games_per_gpu": G, "sims": S,
                       "parallelism": "games sharded x%d, all-gather of samples at step end" % world},
            "gathered_records": int(rec["valid"].sum()), "gathered_games": int((rec["valid"].sum(axis=2) > 0).sum())}))
    dist.barrier()
    dist.destroy_process_group()


def run_rank(args):
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        sys.exit("bench.py: --gpus %d but WORLD_SIZE=%d: start %d ranks (torch.distributed.run --nproc-per-node %d) or "
                 "run `python bench.py --gpus %d` bare and let it start them" % (args.gpus, world, args.gpus, args.gpus, args.gpus))
    backend = os.environ.get("XQ_BENCH_BACKEND", "nccl")
    if backend != "nccl":
        # CPU rehearsal of the N-rank control flow (tests/test_bench_launcher_cpu.py): the launcher, the
        # rank/world checks, sharded seeds, the production all-gather, barrier + max-over-ranks timing
        # and the JSON line are the real ones; the HIP engine is replaced by the stand-in module the
        # test names.  Never a benchmark: the line says so.
        return run_rank_rehearsal(args, backend, rank, world)

    import torch
    import torch.distributed as dist
    from chinesechessai_amd import _lib
    from chinesechessai_amd import distributed as xd
    from chinesechessai_amd.engine import SelfPlayEngine, TorchNetEvaluator
    from chinesechessai_amd.neural_network import ChessNet

    if args.conv_variant:
        _lib.lib().xq_conv3x3_set_variant(args.conv_variant)
    if args.fc_variant >= 0:
        _lib.lib().xq_policy_fc_set_variant(args.fc_variant)
    if args.tower_variant >= 0:
        _lib.lib().xq_tower_set_variant(args.tower_variant)
    if args.search_occ:
        _lib.lib().xq_engine_set_search_occupancy(args.search_occ)
    # XQ_BENCH_FORCE_DIST=1 rehearses the RCCL code path (init, broadcast, all-gather, barrier, all-reduce)
    # with a single rank under torch.distributed.run
    use_dist = world > 1 or (os.environ.get("XQ_BENCH_FORCE_DIST") == "1" and "RANK" in os.environ)
    if use_dist:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        torch.cuda.set_device(local_rank)
        # RCCL prints a version banner on stdout when its first communicator comes up: stdout carries the ONE JSON line, so
        # the banner goes to stderr (file descriptor 1 points at 2 until the first collective has run)
        sys.stdout.flush()
        saved_stdout = os.dup(1)
        os.dup2(2, 1)
        try:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
            assert dist.get_world_size() == args.gpus, (dist.get_world_size(), args.gpus)
            dist.all_reduce(torch.zeros(1, device="cuda"))
            torch.cuda.synchronize()
        finally:
            sys.stdout.flush()
            os.dup2(saved_stdout, 1)
            os.close(saved_stdout)
    else:
        torch.cuda.set_device(0)
    dev = torch.cuda.current_device()
    G, S = args.games, args.sims

    # random-init weights of the architecture: rank 0's (seed 0) are THE weights; every other rank starts from its own
    # seed and receives rank 0's by broadcast, as a trainer's ranks would at the start of an epoch (the reference
    # ships the state_dict to every worker, self_play.py:386,394)
    torch.manual_seed(rank)
    net = ChessNet(num_blocks=args.blocks).eval().cuda()
    bcast_bytes = xd.broadcast_weights(net, src=0) if use_dist else 0
    dtype = torch.bfloat16 if args.dtype == "bf16" else torch.float32
    weights_equal = None

    def make_ev(policy_columns, leaf_dedupe=True, eval_cache=None, net_=None):
        return TorchNetEvaluator(net_ if net_ is not None else net, dtype=dtype, chunk=args.net_chunk or None,
                                 policy_columns=policy_columns, fused_tower=bool(args.fused_tower), leaf_dedupe=leaf_dedupe,
                                 eval_cache=(not args.no_eval_cache) if eval_cache is None else eval_cache)
    ev = make_ev(args.policy_columns, not args.no_leaf_dedupe)
    if use_dist:
        # every rank's leaf evaluator reads the same bits: MIN / MAX all-reduce of a digest of the folded weights
        weights_equal = xd.weights_equal_across_ranks(ev.inet.folded_weights())
    stream = torch.cuda.current_stream().cuda_stream
    records = torch.zeros((args.refill if args.refill else G) * _lib.MAX_PLIES * xd.RECORD_BYTES, dtype=torch.uint8, device="cuda")
    pipe = xd.RecordGather(records.numel(), "cuda") if use_dist else None

    # network forward / trunk kernel timed with events on their own (= the engine's) stream
    class Timed:
        """HIP events (on the engine's stream) around every `stride`-th network forward and its trunk launch: an event pair is
        a marker in the GPU's queue and not free (value_without_event_timing), and 1 forward in 4 of ~1,500 is sample enough;
        stride and the 7 rounds of a ply are coprime, so every round index is sampled equally often"""
        def __init__(self, ev, stride):
            self.ev, self.fw, self.tower, self.idx, self.count, self.stride = ev, [], [], [], 0, max(1, stride)
            self.orig = ev.evaluate

        def on(self, timed):
            if timed:
                def timed_eval(engine):
                    i = self.count
                    self.count += 1
                    if i % self.stride:
                        self.ev.inet.tower_events = None
                        return self.orig(engine)
                    self.ev.inet.tower_events = self.tower
                    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    a.record()
                    out = self.orig(engine)
                    b.record()
                    self.fw.append((a, b))
                    self.idx.append(i)
                    return out
                self.ev.evaluate = timed_eval
            else:
                self.ev.evaluate = self.orig
                self.ev.inet.tower_events = None

    tm = Timed(ev, args.event_stride)
    TG = args.refill if args.refill else G                  # games per GPU and step
    if args.refill and (args.refill < G or args.temp_cutoff or args.tree_reuse):
        sys.exit("bench.py: --refill TOTAL needs TOTAL >= --games and no per-ply temperature schedule / tree reuse")

    def step(eng, ev_, base_seed):
        seeds = xd.game_seeds(base_seed, TG * world, rank, world)
        if args.refill:
            buf = pipe.next_buffer() if use_dist else records
            step.outcomes, step.plies = eng.play_refill(ev_, seeds, buf.data_ptr())
            if use_dist:
                pipe.launch()
        else:
            sched = (lambda ply: 1.0 if ply < args.temp_cutoff else 0.001) if args.temp_cutoff > 0 else None
            eng.play(ev_, seeds, read=False, temperature_schedule=sched)
            if use_dist:
                # the all-gather of this step's samples runs beside the next step's play (two buffers in turn; every
                # gather is waited for inside the timed region: sync() drains the pipeline before the closing barrier)
                eng.pack_samples(pipe.next_buffer().data_ptr())
                pipe.launch()
            else:
                eng.pack_samples(records.data_ptr())

    def sync():
        if use_dist:
            pipe.drain()
            dist.barrier()
        torch.cuda.synchronize()

    eng = SelfPlayEngine(G, sims=S, planes_format=ev.planes_format, device=dev, stream=stream)
    # (a stride that shares a factor with the rounds of a ply would sample the same round indices over and over - 4 rounds at
    # 15 simulations against the default stride of 4: only the light round 0 - so it moves up to the next coprime number)
    import math
    while args.event_stride > 1 and math.gcd(args.event_stride, eng.rounds) != 1:
        args.event_stride += 1
    tm.stride = max(1, args.event_stride)
    if args.no_root_eval_carry:
        eng.set_root_eval_carry(False)
    if args.tree_reuse:
        eng.set_tree_reuse(True)
    if args.virtual_loss:
        eng.set_virtual_loss(True)
    # one-time initialisation that is not part of any step, so that the timed region is clean even
    # with --warmup 0: code-object load (one forward on the full-size buffers),
    # RCCL communicator set-up (one tiny all-gather)
    ev.bind(eng)
    ev.inet(ev.x, out_logits=ev.logits, out_values=ev.values)      # (all rows, no row map: a full-size launch of every kernel)
    if use_dist:
        xd.all_gather_records(torch.zeros(xd.RECORD_BYTES, dtype=torch.uint8, device="cuda"))
    torch.cuda.synchronize()
    if args.profile_plies > 0:
        eng.max_moves = args.profile_plies
    if args.root_noise:
        a_, e_ = (float(v) for v in args.root_noise.split(","))
        eng.set_root_noise(a_, e_, seed=12345 + rank)
    for w in range(args.warmup):
        saved = eng.max_moves
        if args.warmup_plies > 0:
            eng.max_moves = args.warmup_plies
        step(eng, ev, 7_000_000 + w * TG * world)
        eng.max_moves = saved
    sync()
    eng.row_history(cap=0, reset=True)
    eng.eval_cache_stats(reset=True)
    eng.profile(max(1, args.event_stride))
    tm.on(True)
    # shader clock held while the trunk kernel runs: one workgroup in 64 adds its cycles / 100 MHz ticks (xq_debug.h)
    clock_buf = torch.zeros(3, dtype=torch.int64, device="cuda")
    torch.cuda.synchronize()
    _lib.lib().xq_tower_set_clock_sample(clock_buf.data_ptr())
    try:
        t0 = time.time()
        ec_hits_timed = 0
        for k in range(args.steps):
            step(eng, ev, k * TG * world)
            if eng.eval_cache and ev.eval_cache_last is not None:       # (bind() starts the counts afresh every step)
                ec_hits_timed += int(ev.eval_cache_last[0])
        sync()
        dt = time.time() - t0
    finally:                # (the library keeps a raw pointer: never leave it behind, whatever the timed loop did)
        _lib.lib().xq_tower_set_clock_sample(None)
    tm.on(False)
    clk = clock_buf.cpu().numpy()
    trunk_clock_ghz = float(clk[0]) / float(clk[1]) * 0.1 if clk[1] > 0 else None
    prof = eng.profile_read()
    ec_cache_was_on = bool(eng.eval_cache)
    eng.eval_cache_stats(reset=True)
    carry_on = eng._carry_on
    rows_hist, n_rounds = eng.row_history(reset=True) if eng.row_compaction else (None, 0)
    fw_t = np.array([a.elapsed_time(b) for a, b in tm.fw])          # the sampled forwards (tm.idx: which ones)
    tw_t = np.array([a.elapsed_time(b) for a, b in tm.tower])
    n_fw, fw_idx = tm.count, np.array(tm.idx, dtype=np.int64)
    fw_scale = n_fw / max(len(fw_t), 1)                             # sampled -> all forwards
    fw_ms = float(fw_t.sum()) * fw_scale
    # the tree kernel: every args.event_stride-th launch was timed; one k_search_round per forward
    if prof["search_launches"] > 0 and n_fw > 0:
        prof["search_ms"] *= n_fw / prof["search_launches"]
        prof["search_launches_timed"], prof["search_launches"] = prof["search_launches"], n_fw
    outcomes = step.outcomes if args.refill else eng.read_game_outcomes()
    t = torch.tensor([dt], dtype=torch.float64, device="cuda")
    if use_dist:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    dt = float(t.item())

    # ---- beside the headline (N = 1 only, a few steps each, outside the timed region): every root evaluated afresh
    # like the reference (no carry-over), and the full 8,100-column policy head
    aux = {}
    if world == 1 and args.aux_steps > 0 and not args.profile_plies and not args.refill:
        def timed_steps(ev_, base):
            torch.cuda.synchronize()
            ta = time.time()
            for k in range(args.aux_steps):
                step(eng, ev_, base + k * TG)
            torch.cuda.synchronize()
            return TG * args.aux_steps / (time.time() - ta)
        # the headline configuration once more without the per-launch HIP events of the timed region (4 per forward + 2 per
        # tree launch: what the instrumentation itself costs)
        aux["value_without_event_timing"] = timed_steps(ev, 6_100_000)
        # (the no-dedupe and no-carry legs run WITHOUT the evaluation cache too: the cache would answer most of what the other
        # two eliminations leave behind - a re-evaluated root was a leaf one ply earlier - and the legs are meant to show the
        # path with one network row per pending leaf)
        ev_nd = make_ev(args.policy_columns, False, eval_cache=False) if (ev.leaf_dedupe or ev.eval_cache) else ev
        if ev_nd is not ev:
            step(eng, ev_nd, 7_000_000)                                # (untimed: first step after the switch)
            aux["value_no_dedupe"] = timed_steps(ev_nd, 7_100_000)
        if carry_on:
            eng.set_root_eval_carry(False)
            step(eng, ev_nd, 8_000_000)
            aux["value_no_carry"] = timed_steps(ev_nd, 8_100_000)
            eng.set_root_eval_carry(None)
        if args.policy_columns == "reachable":
            ev_all = make_ev("all", ev.leaf_dedupe)
            step(eng, ev_all, 9_000_000)
            aux["value_full_policy_head"] = timed_steps(ev_all, 9_100_000)
            del ev_all
        del ev_nd
        if ev.eval_cache:
            # the evaluation cache (position -> priors + value, two plies).  Random-init priors are 0.35 % apart, a search never
            # descends twice into the same child and nothing below the played move is expanded: what the cache answers in the
            # headline workload are the transpositions of the opening plies (~2 % of the rows).  What it is for is a TRAINED
            # network, whose search follows lines: the same architecture with the policy head's weights and bias scaled by
            # 256 (seeded random-init otherwise: the top move then holds ~0.9 of the prior mass), with the cache and without
            import copy
            ev_off = make_ev(args.policy_columns, ev.leaf_dedupe, eval_cache=False)
            step(eng, ev_off, 10_000_000)
            aux["value_no_eval_cache"] = timed_steps(ev_off, 10_100_000)
            del ev_off
            net_pk = copy.deepcopy(net)
            with torch.no_grad():
                net_pk.policy_fc.weight.mul_(256.0)
                net_pk.policy_fc.bias.mul_(256.0)
            for key, cache in (("value_peaked_priors", True), ("value_peaked_priors_no_reuse", False)):
                ev_pk = make_ev(args.policy_columns, ev.leaf_dedupe, eval_cache=cache, net_=net_pk)
                step(eng, ev_pk, 11_000_000)
                eng.eval_cache_stats(reset=True)
                eng.row_history(cap=0, reset=True)
                aux[key] = timed_steps(ev_pk, 11_100_000)
                rows_pk, _ = eng.row_history(reset=True)
                aux[key + "_rows_per_game"] = float(rows_pk.astype(np.int64).sum()) / (TG * args.aux_steps)
                if cache:
                    hits, fills, _ = eng.eval_cache_stats(reset=True)
                    aux["peaked_priors_cache_hits_per_game"] = hits / float(TG * args.aux_steps)
                del ev_pk
            del net_pk
            aux["peaked_priors_note"] = ("the same workload with the policy head scaled by 256 (policy_fc.weight and .bias of the "
                                         "seeded random-init network x 256: peaked priors, a stand-in for a trained network); "
                                         "value_peaked_priors = everything on, value_peaked_priors_no_reuse = the evaluation cache "
                                         "off; rows_per_game = network rows evaluated per game")
        aux["aux_note"] = ("games/s over %d extra steps each, same engine, outside the timed region: value_no_dedupe = every "
                           "pending leaf has its own network row (no dedupe, no evaluation cache; carry-over on: the round-3a form); value_no_carry = that and "
                           "every root evaluated afresh, i.e. the reference's evaluation count, 7 forwards of G rows per ply "
                           "(rounds 1 and 2 of this project quoted this form); value_full_policy_head = policy FC on all 8,100 "
                           "columns (carry-over and dedupe as in the headline)" % args.aux_steps)

    if rank == 0:
        games = TG * world * args.steps
        rows = eng.n_rows                                   # network rows per forward (one per game; x8 slots with virtual loss)
        fl = net_flops_per_row(args.blocks)
        fl -= 2 * 2880 * (8100 - ev.inet.n_policy_real)        # only the policy columns actually computed count (no padding)
        # rows each forward really evaluated (row compaction: device-side counts of every search round, in launch order)
        rows_known = rows_hist is not None and len(rows_hist) == n_fw
        if rows_known:
            rows_fw = rows_hist.astype(np.int64)
        else:
            # no row compaction (every launch runs every slot), or more search rounds than the engine's row history keeps
            # (65,536): then the row-derived figures below are upper bounds and say so
            rows_fw = np.full(n_fw, rows, np.int64)
        rows_note = None if (rows_known or not eng.row_compaction) else (
            "row history incomplete (%d rounds launched, the engine keeps %d): rows_evaluated / roofline_net / "
            "end_to_end_mfma_frac assume full-size launches and are upper bounds" % (n_rounds, len(rows_hist)))
        net_tflops = float(fl * rows_fw.sum()) / (fw_ms * 1e-3) / 1e12 if fw_ms > 0 else 0.0
        # k_search_round, priced by what each launch does: under the carry-over round 0 of every ply after a game's first has
        # nothing pending and nothing to descend to (the root is expanded, its 8 visits are booked): 64 B per game (board +
        # scalars); every other launch is a descent (tree_bytes_per_descent) plus, with the leaf dedupe, the 13 words of the
        # position written past the L2 and the table entry (52 + 8 B written, 8 + 52 B read on a probe: 120 B)
        bpd = tree_bytes_per_descent() + (120.0 if eng.leaf_dedupe else 0.0)
        n_search = prof["search_launches"]
        n_light = 0
        if carry_on and not args.refill and eng.rounds > 0:
            n_light = max(n_search // eng.rounds - args.steps, 0)          # plies played minus the first ply of every step
        tree_bytes = (n_search - n_light) * bpd * G + n_light * 64.0 * G
        tree_gbs = tree_bytes / (prof["search_ms"] * 1e-3) / 1e9 if prof["search_ms"] > 0 else 0.0
        # dominant kernel: the single-launch trunk (or, with --fused-tower 0, the per-layer conv)
        fused = bool(args.fused_tower) and ev.inet.use_hip_conv
        if fused:       # k_tower: conv1 + 2*blocks convs + both heads per launch
            per_board = 2.0 * 90 * (16 * 9 * 128 + 2 * args.blocks * 128 * 9 * 128 + 128 * 40)
            rows_tw = rows_fw[fw_idx] if (len(tw_t) == len(fw_idx) and len(rows_fw) == n_fw) else np.full(len(tw_t), rows, np.int64)
            # full-size launches: (nearly) every slot has a row - a round in which a few games ended on a terminal leaf or
            # are already over still is one; their flops are counted by the rows they really ran
            full = rows_tw >= 0.98 * rows
            if not full.any():                             # (a short or small run that the leaf dedupe never lets fill up:
                full = rows_tw > 0                         #  every launch with rows, priced by the rows it ran)
            n_conv = int(full.sum())
            conv_ms = float(tw_t[full].sum())
            conv_fl = per_board * float(rows_tw[full].mean()) if n_conv else per_board * rows
            launches = {"full_size": n_conv, "rows_avg_full_size": float(rows_tw[full].mean()) if n_conv else None,
                        "empty": int((rows_tw == 0).sum()), "partial": int(((rows_tw > 0) & ~full).sum()),
                        "empty_ms_avg": float(tw_t[rows_tw == 0].mean()) if (rows_tw == 0).any() else None,
                        "timed_launches": int(len(tw_t)), "timed_every": tm.stride, "all_launches": int(n_fw),
                        "all_ms": float(tw_t.sum()) * fw_scale}
            tv = args.tower_variant if args.tower_variant >= 0 else (60 if rows >= 2048 else 36)
            kbuild = {60: "k_tower1wa (one wave per SIMD, 4 boards per workgroup, the residual tower as one hand-written asm statement)",
                      39: "k_tower16b<NB = 4>", 36: "k_tower16b<NB = 2>", 0: "k_tower (32x32x16)"}.get(tv, "variant %d" % tv)
            kname, kdesc = "k_tower", "%s, hand-written single-launch trunk on v_mfma_f32_16x16x32_bf16: conv3x3(16->128) + %d fused residual convs + 1x1 heads, activations resident in LDS" % (
                kbuild, 2 * args.blocks)
        else:
            n_conv = 2 * args.blocks * len(tw_t)
            conv_ms = float(tw_t.sum())
            conv_fl = 2.0 * rows * 90 * 128 * 9 * 128
            launches = None
            kname, kdesc = "k_conv3x3_b<128>", "hand-written fused conv3x3+bias+residual+ReLU"
        conv_tflops = conv_fl * n_conv / (conv_ms * 1e-3) / 1e12 if conv_ms > 0 else 0.0
        tr_tower = pmc_traffic(kname, G, S, args.blocks, fetch_factor=2.0) if rows == G else (None, None)
        tr_tree = pmc_traffic("k_search_round", G, S, args.blocks)
        tr_fc = pmc_traffic("k_policy_fc", G, S, args.blocks, fetch_factor=2.0) if rows == G else (None, None)
        # counter traffic over algorithmic HBM bytes per launch (the profiled passes run full-size launches without dedupe):
        # trunk: planes in + head activations out + the weight set once; tree: a descent per game; policy FC: activations in
        # + weights + logits out
        alg_tower = rows * (2880.0 + 7200.0) + (2.0 * (16 * 9 * 128 + 2 * args.blocks * 128 * 9 * 128 + 128 * 64) + 4.0 * 128 * (1 + 2 * args.blocks))
        alg_fc = rows * 2880.0 * 2 + 2880.0 * ev.inet.n_policy * 2 + rows * ev.inet.n_policy * 2.0
        ratio = lambda t, a: (t / a) if t else None
        traffic_ratio = {"k_tower": ratio(tr_tower[0], alg_tower), "k_search_round": ratio(tr_tree[0], tree_bytes_per_descent() * G),
                         "k_policy_fc": ratio(tr_fc[0], alg_fc),
                         "note": "HBM bytes from the committed PMC passes (FETCH_SIZE x2 for the 16-byte-per-lane readers + WRITE_SIZE; tree kernel: "
                                 "FETCH_SIZE + WRITE_SIZE) over algorithmic bytes per full-size launch", "source": tr_tower[1] or tr_tree[1]}
        extras = [", policy FC on the %d columns a legal move can index (of 8,100; the others are never read by the search "
                  "and are not counted as work)" % ev.inet.n_policy_real if args.policy_columns == "reachable"
                  else ", full 8,100-column policy FC"]
        if args.root_noise or args.temp_cutoff:
            extras.append(", Dirichlet root noise %s, temperature 1 -> 0 at ply %d (extensions, no reference oracle)"
                          % (args.root_noise, args.temp_cutoff))
        if carry_on:
            extras.append("; root evaluation carry-over ON (result-identical, tested at this size: the new root's priors are the "
                          "ones the played child received during the previous ply's search, so round 0 of every ply after the "
                          "first has no network rows: %d forwards with rows in this run instead of %d; value_no_carry is the "
                          "figure with every root evaluated afresh)" % (int((rows_fw > 0).sum()), n_fw))
        if eng.row_compaction:
            extras.append("; evaluator row compaction (only slots with a pending leaf are network rows)")
        if eng.leaf_dedupe:
            extras.append("; leaf dedupe ON (result-identical, tested at this size: pending leaves of a round that are the same "
                          "position share one network row - every game starts from the same position with the same weights, so "
                          "the first plies of a step repeat across games: %d rows evaluated in this run of %d forwards; value_no_dedupe is "
                          "the figure with one row per pending leaf)" % (int(rows_fw.sum()), n_fw))
        if ev.eval_cache:
            extras.append("; evaluation cache (result-identical, tested: a position's priors and value are kept for two plies and answer "
                          "for any later leaf that is the same position) %s; value_no_eval_cache is the figure without it (taken "
                          "like value_without_event_timing: no per-launch events), value_peaked_priors / _no_reuse show what it is for" % (
                              "ON: %d leaves answered in this run" % ec_hits_timed if ec_cache_was_on else
                              "adaptive: it answered %s in its last step (random-init priors make the search revisit next to "
                              "nothing; an answer saves a row, the probe costs the tree kernel ~10 %%) and suspended itself for the timed steps" % (
                                  "%d, against %d rows evaluated," % (ev.eval_cache_last[0], ev.eval_cache_last[2]) if ev.eval_cache_last else "none")))
        if args.tree_reuse:
            extras.append(", tree reuse (extension)")
        if args.virtual_loss:
            extras.append(", virtual loss: %d slots per game and round (extension)" % (rows // G))
        if args.refill:
            extras.append("; REFILL mode: %d games per GPU and step through the %d slots, a finished game's slot restarted at "
                          "once (steady state; reported beside the headline lock-step number, not instead of it)" % (args.refill, G))
        workload = "%s%d concurrent games/GPU, %d sims, %d-block ResNet %s, random-init weights, start positions, seeds base+g%s" % (
            "BASELINE configs[2]: " if (G, S, args.blocks) == (16384, 50, 6) and not args.refill else "",
            G, S, args.blocks, args.dtype, "".join(extras))
        out = {
            "metric": "self-play games/sec @ 50 MCTS sims" if S == 50 else "self-play games/sec @ %d MCTS sims" % S,
            "value": games / dt, "unit": "games/s", "n_gpus": dist.get_world_size() if use_dist else 1, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": args.dtype, "data": "synthetic" if not args.profile_plies else
            "synthetic; PROFILING RUN truncated to %d plies per step - not a benchmark" % args.profile_plies,
            "config": {"workload": workload,
                       "games_per_gpu": G, "sims": S, "blocks": args.blocks, "max_moves": 70,
                       "parallelism": "games sharded x%d, weights broadcast from rank 0 (%d bytes), all-gather of samples at step end" % (world, bcast_bytes)},
            "ranks": world, "weights_equal": weights_equal,
            "roofline": {"bound": "mfma", "achieved": conv_tflops, "peak": MFMA_PEAK_BF16_TFLOPS, "unit": "TFLOP/s",
                         "frac": conv_tflops / MFMA_PEAK_BF16_TFLOPS,
                         "traffic": tr_tower[0], "traffic_source": tr_tower[1],
                         "kernel": "%s (%s; %d full-size launches timed - HIP events around 1 launch in %d - of %d boards on average, %.4f ms avg; %.0f%% of the step)" % (
                             kname, kdesc, n_conv, tm.stride, int(round(conv_fl / (per_board if fused else conv_fl / rows))), conv_ms / max(n_conv, 1),
                             100.0 * conv_ms * fw_scale / (dt * 1e3)),
                         "flops_per_launch": conv_fl, "launches": launches,
                         "clock_ghz": trunk_clock_ghz,
                         "clock_note": "shader clock the chip held inside the trunk kernel over the timed region (s_memtime / s_memrealtime "
                                       "of one workgroup in 64, k_tower1wa only); 2.4 GHz is the peak's clock: frac = MFMA-pipe efficiency x clock / 2.4"},
            "roofline_net": {"bound": "mfma", "achieved": net_tflops, "peak": MFMA_PEAK_BF16_TFLOPS, "unit": "TFLOP/s",
                             "frac": net_tflops / MFMA_PEAK_BF16_TFLOPS, "traffic": None,
                             "kernel": "whole network forward (%d launches, %d rows evaluated in all, %.3f ms avg over all launches, estimated from the 1 in %d that carry events)" % (
                                 n_fw, int(rows_fw.sum()), fw_ms / max(n_fw, 1), tm.stride),
                             "flops_per_row": fl, "rows_evaluated": int(rows_fw.sum()), "note": rows_note},
            # the whole step against the MFMA peak: every network row evaluated x its flops, over the WALL time of the timed region
            # (per GPU: rank 0's rows over the max-over-ranks time against ONE GPU's peak; every rank plays the same number of games)
            "end_to_end_mfma_frac": float(fl * rows_fw.sum()) / dt / 1e12 / MFMA_PEAK_BF16_TFLOPS,
            "roofline_tree": {"bound": "hbm", "achieved": tree_gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                              "frac": tree_gbs / HBM_PEAK_GBS, "traffic": tr_tree[0], "traffic_source": tr_tree[1],
                              "kernel": "k_search_round (%d launches, %d of them timed, %.3f ms avg)" % (
                                  prof["search_launches"], prof.get("search_launches_timed", prof["search_launches"]),
                                  prof["search_ms"] / max(prof["search_launches"], 1)),
                              "bytes_per_launch": tree_bytes / max(n_search, 1),
                              "bytes_note": "%d launches that descend (%.0f B per game: board + scalars, PUCT reads, leaf moves / board / planes, "
                                            "logit gather, edge init, backup%s) + %d launches of round 0 under the carry-over (64 B per game)" % (
                                                n_search - n_light, bpd, ", dedupe words + table entry" if eng.leaf_dedupe else "", n_light)},
            "traffic_over_algorithmic": traffic_ratio,
            "build_libxq_hip_sha16": lib_sha16(),
            "time_share": {"net_forward_ms": fw_ms, "k_search_round_ms": prof["search_ms"],
                           "k_play_move_ms": prof["play_ms"], "wall_ms": dt * 1e3},
            "games": {"mean_plies": float(outcomes["n_plies"].mean()),
                      "draws_by_cap": int((outcomes["reason"] == 8).sum()), "errors": int(outcomes["error"].sum())},
        }
        out.update(aux)
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(args.blocks, S, plies=70 if args.cpu_baseline_full else 18)
        print(json.dumps(out))
    eng.close()
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
