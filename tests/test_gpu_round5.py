"""Round-5 GPU tests (`-m gpu`, through the C ABI): the default trunk kernel against the other build bit for bit at the
only size the bench runs (VERDICT r04 missing #2), refill x leaf dedupe x root carry-over at a size where the dedupe
table is really shared between XCDs, and the ADVICE r04 items."""
import os
import struct

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from tests.test_gpu_round4 import _mate_line_net, _same_game  # noqa: E402


@pytest.fixture(scope="module")
def L():
    from chinesechessai_amd import _lib
    lib = _lib.lib()
    assert lib.xq_device_count() > 0, "no GPU visible"
    assert lib.xq_device_ok(0) == 1, "not a gfx950 device"
    return lib


def _trunk(L, inet, variant, planes, G, blocks, row_src=None, n_rows=None, fill=None):
    import torch
    from chinesechessai_amd import _lib
    st = torch.cuda.current_stream().cuda_stream
    P = torch.full((G + 1, 2880), 9.0, device="cuda", dtype=torch.bfloat16)
    V = torch.full((G + 1, 720), 9.0, device="cuda", dtype=torch.bfloat16)
    L.xq_tower_set_variant(variant)
    try:
        _lib.check(L.xq_tower_nhwc_bf16(st, planes.data_ptr(), inet.hip_w[0].data_ptr(), inet.hip_wt.data_ptr(),
                                        inet.hip_bt.data_ptr(), inet.hip_hw.data_ptr(), inet.hip_hb.data_ptr(),
                                        P.data_ptr(), V.data_ptr(), G, blocks,
                                        None if row_src is None else row_src.data_ptr(),
                                        None if n_rows is None else n_rows.data_ptr()))
        torch.cuda.synchronize()
    finally:
        L.xq_tower_set_variant(-1)
    return P, V


@pytest.mark.parametrize("blocks", [6, 20])
def test_default_trunk_kernel_equals_the_other_build_at_full_size(L, blocks):
    """k_tower1wa (variant 60: the default from 2,048 boards up; one wave per SIMD, the residual tower as one generated
    asm statement) against k_tower16b<4> (variant 39: round 3's default, plain HIP) BIT FOR BIT at the size the bench
    runs - 16,384 boards = 4,096 workgroups, 16 per CU back to back - with 6 blocks (BASELINE C3) and 20 (C5), on
    position-like planes (each square holds at most one piece: what encode_board produces, neural_network.py:128-146)
    and on dense random ones; then a ROW-MAPPED launch as the engine issues it (row r reads the planes of slot
    row_src[r]; the count comes from device memory and is ragged: 13,001 of 16,384, not a multiple of 4): rows below
    the count equal the plain launch's rows of their slots in both builds, rows above it are not written.
    (neural_network.py:47-71: one forward; the two kernels keep one summation order.)"""
    import torch
    from chinesechessai_amd.neural_network import ChessNet, InferenceNet
    G = 16384
    torch.manual_seed(100 + blocks)
    net = ChessNet(num_blocks=blocks).eval()
    for m in net.modules():
        if isinstance(m, torch.nn.BatchNorm2d):
            m.running_mean.normal_(0, 0.1)
            m.running_var.uniform_(0.5, 1.5)
    inet = InferenceNet(net.cuda())
    g = torch.Generator(device="cuda").manual_seed(5)
    # position-like: one of 15 codes per square (0 = empty, mostly), channel 14 = side to move for the whole board
    code = torch.randint(0, 40, (G, 10, 9), device="cuda", generator=g)
    planes = torch.zeros(G, 10, 9, 16, device="cuda", dtype=torch.bfloat16)
    for c in range(14):
        planes[..., c] = (code == c + 1).to(torch.bfloat16)
    planes[..., 14] = (torch.rand(G, 1, 1, device="cuda", generator=g) < 0.5).to(torch.bfloat16).expand(G, 10, 9)
    dense = torch.zeros(G, 10, 9, 16, device="cuda", dtype=torch.bfloat16)
    dense[..., :15] = (torch.rand(G, 10, 9, 15, device="cuda", generator=g) < 0.3).to(torch.bfloat16)
    for x in (planes, dense):
        Pa, Va = _trunk(L, inet, 60, x, G, blocks)
        Pb, Vb = _trunk(L, inet, 39, x, G, blocks)
        assert Pa[:G].float().abs().max().item() > 0
        assert torch.equal(Pa.view(torch.int16), Pb.view(torch.int16))           # (row G = the guard row, 9.0 in both)
        assert torch.equal(Va.view(torch.int16), Vb.view(torch.int16))
        assert (Pa[G] == 9.0).all() and (Va[G] == 9.0).all()
    # the engine's form of the launch: a permutation of the slots, 13,001 rows present
    n = 13001
    perm = torch.randperm(G, device="cuda", generator=g).to(torch.int32)
    n_rows = torch.tensor([n], device="cuda", dtype=torch.int32)
    Pa, Va = _trunk(L, inet, 60, planes, G, blocks)
    for variant in (60, 39):
        Pm, Vm = _trunk(L, inet, variant, planes, G, blocks, row_src=perm, n_rows=n_rows)
        idx = perm[:n].long()
        assert torch.equal(Pm[:n].view(torch.int16), Pa[idx].view(torch.int16)), variant
        assert torch.equal(Vm[:n].view(torch.int16), Va[idx].view(torch.int16)), variant
        # a workgroup carries 4 rows: rows up to the next multiple of 4 may be computed, nothing beyond is touched
        assert (Pm[(n + 3) // 4 * 4:] == 9.0).all() and (Vm[(n + 3) // 4 * 4:] == 9.0).all(), variant


def test_refill_with_dedupe_and_carry_at_a_contended_size(L):
    """refill x leaf dedupe x root carry-over where the dedupe table is shared for real - 2,048
    slots (8 per CU: waves of all 8 XCDs meet in the table's entries; in the opening every slot holds the same
    position), 6,144 games, the mate-line network (a fifth of the games end at ply 7, so slots restart at plies 7, 14,
    ... 70, 77 ... next to games in mid-play, and every restart re-enters the start position other restarted slots
    hold too).  A 96-game sample - ids spread over first-deal and restarted games - must equal, record for record and z
    bit for bit, the PLAIN lock-step path (no dedupe, no carry-over, every root evaluated afresh, one row per leaf)
    playing the same seeds; every game's outcome must be consistent with its records; the guard blocks stay untouched.
    Then the same schedule with the exact evaluator: 6,144 games through 2,048 slots, 64 of them against the CPU
    oracle's game for the seed, move for move, visit for visit, z bit for bit (self_play.py:404-408: handing a worker
    its next game must not change any game)."""
    import torch
    from chinesechessai_amd import distributed as xd
    from chinesechessai_amd.engine import HashNetEvaluator, SelfPlayEngine, TorchNetEvaluator
    from oracle import xq_oracle as xo
    G, T, S = 2048, 6144, 50
    net = _mate_line_net(2, 11)
    seeds = (np.arange(T, dtype=np.uint32) * 7 + 3).astype(np.uint32)
    block = 70 * xd.RECORD_BYTES

    ev = TorchNetEvaluator(net)
    eng = SelfPlayEngine(G, sims=S, planes_format=ev.planes_format)
    buf = torch.full(((T + 2) * block,), 0xA5, dtype=torch.uint8, device="cuda")
    out, plies = eng.play_refill(ev, seeds, buf.data_ptr() + block, check_every=4)
    assert eng._carry_on and eng.row_compaction and eng.leaf_dedupe                    # everything automatic
    rows, n_rounds = eng.row_history()
    eng.close()
    assert int(out["error"].sum()) == 0
    assert (buf[:block] == 0xA5).all().item() and (buf[-block:] == 0xA5).all().item()  # guard blocks
    lengths = out["n_plies"]
    assert 0.05 * T <= (lengths < 70).sum() <= 0.6 * T, np.bincount(lengths)           # the network does what it is for
    assert plies < 3 * 70 + 8                                                          # slots restarted out of step
    # the dedupe did share rows: the opening round of the session is one row, and over the session fewer rows than leaves
    per = rows.reshape(-1, eng.rounds)
    assert per[0, 0] == 1 and per[0, 1] == 1 and per.max() <= G

    sample = np.unique(np.concatenate([np.linspace(0, G - 1, 32).astype(int), np.linspace(G, T - 1, 64).astype(int)]))
    rec_all = buf[block:-block].view(T, block)
    rec = np.frombuffer(rec_all[torch.as_tensor(sample, device="cuda")].cpu().numpy().tobytes(),
                        dtype=xd.RECORD_DTYPE).reshape(len(sample), 70)
    evp = TorchNetEvaluator(net, leaf_dedupe=False, eval_cache=False)
    engp = SelfPlayEngine(len(sample), sims=S, planes_format=evp.planes_format)
    engp.set_root_eval_carry(False)
    engp.play(evp, seeds[sample], read=False)
    ref_t = torch.zeros(len(sample) * block, dtype=torch.uint8, device="cuda")
    engp.pack_samples(ref_t.data_ptr())
    ref_out = engp.read_game_outcomes()
    engp.close()
    ref = xd.records_to_numpy(ref_t).reshape(len(sample), 70)
    for k in ("winner", "reason", "reason_side", "reason_count", "n_plies", "n_samples", "error"):
        assert np.array_equal(out[k][sample], ref_out[k]), k
    for i, g in enumerate(sample):
        _same_game(rec[i], ref[i], int(g))
    assert (ref_out["n_plies"] < 70).sum() >= 4                                        # short games are in the sample
    # every game of the session, not only the sample: outcome scalars agree with the records' valid flags
    valid = np.frombuffer(rec_all.cpu().numpy().tobytes(), dtype=xd.RECORD_DTYPE).reshape(T, 70)["valid"]
    assert np.array_equal(valid.sum(axis=1), out["n_samples"])

    # the same schedule with the exact evaluator, against the oracle
    eng = SelfPlayEngine(G, sims=S)
    rec_t = torch.zeros(T * block, dtype=torch.uint8, device="cuda")
    out, plies = eng.play_refill(HashNetEvaluator(), seeds, rec_t.data_ptr(), check_every=4)
    eng.close()
    assert int(out["error"].sum()) == 0
    pick = np.linspace(0, T - 1, 64).astype(int)
    rec = np.frombuffer(rec_t.view(T, block)[torch.as_tensor(pick, device="cuda")].cpu().numpy().tobytes(),
                        dtype=xd.RECORD_DTYPE).reshape(len(pick), 70)
    for i, g in enumerate(pick):
        rc, og = xo.self_play_game(int(seeds[g]), S)
        assert rc == 0
        assert (out["winner"][g], out["reason"][g], out["n_plies"][g], out["n_samples"][g]) == (
            og.winner, og.end_reason, og.n_plies, og.n_samples), g
        assert int(rec[i]["valid"].sum()) == og.n_samples
        for j in range(og.n_samples):
            k = og.s_nmoves[j]
            assert int(rec[i, j]["n_moves"]) == k and int(rec[i, j]["chosen"]) == og.t_move[j], (g, j)
            assert rec[i, j]["moves"][:k].tolist() == list(og.s_moves[j][:k]), (g, j)
            assert rec[i, j]["counts"][:k].tolist() == list(og.t_visits[j][:k]), (g, j)
            assert struct.pack("<d", og.s_z[j]) == struct.pack("<d", float(rec[i, j]["z"])), (g, j)


def test_auto_carry_needs_row_compaction(L):
    """the automatic root-evaluation carry-over is limited to evaluators WITH row compaction (without it
    skipping round 0 costs a blocking read per ply); the exact evaluator gets it only when asked."""
    from chinesechessai_amd.engine import HashNetEvaluator, SelfPlayEngine
    seeds = np.arange(4, dtype=np.uint32)
    eng = SelfPlayEngine(4, sims=16, max_moves=2)
    a = eng.play(HashNetEvaluator(), seeds)
    assert not eng._carry_on
    eng.set_root_eval_carry(True)
    b = eng.play(HashNetEvaluator(), seeds)
    assert eng._carry_on
    eng.close()
    assert np.array_equal(a.chosen, b.chosen) and np.array_equal(a.s_counts, b.s_counts)


def test_eval_cache_is_result_identical_and_answers_for_peaked_priors(L):
    """evaluation reuse below the root.  The reference rebuilds its tree every ply (self_play.py:98)
    and evaluates again what the last ply's search expanded below the move that was played; the engine's evaluation cache
    (xq_engine_set_eval_cache: position -> priors + value, kept for two plies, shared by all games) answers those leaves
    without a network row.  1,024 games x S = 50 with the mate-line network (peaked priors: a search follows one line, so
    the next ply meets its own expansions again): every game's records must equal, field for field and z bit for bit, the
    PLAIN path (no cache, no dedupe, no carry-over), with the cache alone and with everything on; the cache must really
    answer (hits, fewer evaluator rows); and a cached engine must not carry anything over to a second network."""
    import torch
    from chinesechessai_amd import distributed as xd
    from chinesechessai_amd.engine import SelfPlayEngine, TorchNetEvaluator
    G, S = 1024, 50
    net = _mate_line_net(2, 11)
    seeds = (np.arange(G, dtype=np.uint32) * 5 + 1).astype(np.uint32)
    block = 70 * xd.RECORD_BYTES

    def run(net_, dedupe, cache, carry):
        ev = TorchNetEvaluator(net_, leaf_dedupe=dedupe, eval_cache=cache)
        eng = SelfPlayEngine(G, sims=S, planes_format=ev.planes_format)
        if not carry:
            eng.set_root_eval_carry(False)
        eng.play(ev, seeds, read=False)
        assert eng.eval_cache == cache and eng.leaf_dedupe == dedupe and eng._carry_on == carry
        t = torch.zeros(G * block, dtype=torch.uint8, device="cuda")
        eng.pack_samples(t.data_ptr())
        out = eng.read_game_outcomes()
        rows, n_rounds = eng.row_history()
        stats = eng.eval_cache_stats()
        eng.close()
        return xd.records_to_numpy(t).reshape(G, 70), out, int(rows.astype(np.int64).sum()), stats

    ref, ref_out, rows_plain, st0 = run(net, False, False, False)
    assert st0 == (0, 0, 0) and int(ref_out["error"].sum()) == 0
    assert (ref_out["n_plies"] < 70).sum() >= G // 20                                  # the network ends games early, too
    for dedupe, cache, carry in ((False, True, False), (True, True, True)):
        rec, out, rows, (hits, fills, _) = run(net, dedupe, cache, carry)
        for k in ("winner", "reason", "reason_side", "reason_count", "n_plies", "n_samples", "error"):
            assert np.array_equal(out[k], ref_out[k]), (k, dedupe, cache, carry)
        for g in range(G):
            _same_game(rec[g], ref[g], (g, dedupe, cache, carry))
        assert hits > 0 and fills > 0, (hits, fills)
        if not dedupe:
            # the cache alone: every hit is a row the plain path evaluated
            assert rows == rows_plain - hits, (rows, rows_plain, hits)
            print("eval cache alone: %d rows without it, %d hits, %d fills" % (rows_plain, hits, fills))
            assert hits > rows_plain // 20, (hits, rows_plain)                          # peaked priors: a real share of the rows
        else:
            assert rows < rows_plain - hits // 2
    # other weights, same engine object semantics: a fresh engine per evaluator in this test; inside ONE engine new games
    # age the cache out (xq_engine_new_games), so a second network never reads the first one's answers
    net2 = _mate_line_net(2, 12)
    ev1, ev2 = TorchNetEvaluator(net), TorchNetEvaluator(net2)
    eng = SelfPlayEngine(64, sims=S, planes_format=ev1.planes_format, max_moves=6)
    a1 = eng.play(ev1, seeds[:64])
    b2 = eng.play(ev2, seeds[:64])
    eng.close()
    eng = SelfPlayEngine(64, sims=S, planes_format=ev2.planes_format, max_moves=6)
    b2_fresh = eng.play(ev2, seeds[:64])
    eng.close()
    assert np.array_equal(b2.s_counts, b2_fresh.s_counts) and np.array_equal(b2.chosen, b2_fresh.chosen)
    assert not np.array_equal(a1.s_counts, b2.s_counts)


def test_eval_cache_verify_mode_at_the_bench_size(L):
    """The evaluation cache checks itself: in verify mode a leaf the cache could answer is evaluated all the same and its
    priors, value and move count are compared with the entry's, bit for bit.  BASELINE C3's size (16,384 games x S = 50 x
    6-block bf16, 8 plies, dedupe and carry-over on and off): about 95,000 / 209,000 leaves compared, NO mismatch, and
    exactly one fill per reserved entry.  (Round 5 found a real bug this way: a sign extension in the 64-bit readfirstlane
    idiom let every wave that lost the reservation race believe it had won - 30 to 80 mismatches per run.)"""
    import torch
    from chinesechessai_amd.engine import SelfPlayEngine, TorchNetEvaluator
    from chinesechessai_amd.neural_network import ChessNet
    torch.manual_seed(0)
    net = ChessNet(num_blocks=6).eval().cuda()
    G, P = 16384, 8
    seeds = np.arange(G, dtype=np.uint32)
    for dedupe, carry in ((True, True), (False, False)):
        ev = TorchNetEvaluator(net, leaf_dedupe=dedupe, eval_cache="verify")
        eng = SelfPlayEngine(G, sims=50, planes_format=ev.planes_format, max_moves=P)
        if not carry:
            eng.set_root_eval_carry(False)
        b = eng.play(ev, seeds)
        compared, fills, bad = eng.eval_cache_stats()
        eng.close()
        assert int(b.error.sum()) == 0
        assert bad == 0, (dedupe, carry, compared, fills, bad)
        assert compared > (20000 if dedupe else 50000) and fills > 50000, (compared, fills)     # (a likely dedupe duplicate does not look the cache up)
        if not carry:
            # every root of plies 1.. was a leaf of the ply before: the cache knows it
            assert compared >= G * (P - 1), compared


def test_eval_cache_suspends_itself_when_it_answers_nothing(L):
    """The default (`eval_cache=True`) watches itself: a play() in which it answered fewer leaves than 1 % of the rows that
    were evaluated suspends it for the evaluator's next 15 plays (its probe is not free).  Games that never meet - every
    game started from a position of its own (`set_roots`) - give it nothing to answer; games from the start position
    (transpositions of the opening) and the mate-line network keep it on.  `eval_cache="on"` never steps aside.  Results
    never change."""
    import torch
    from chinesechessai_amd.engine import SelfPlayEngine, TorchNetEvaluator
    from chinesechessai_amd.neural_network import ChessNet
    torch.manual_seed(3)
    net = ChessNet(num_blocks=2).eval().cuda()
    G = 2048
    seeds = np.arange(G, dtype=np.uint32)
    ev = TorchNetEvaluator(net)
    eng = SelfPlayEngine(G, sims=50, planes_format=ev.planes_format, max_moves=12)
    a = eng.play(ev, seeds)
    hits, fills, rows = ev.eval_cache_last
    assert eng.eval_cache and ev.eval_cache_suspended == 0 and hits >= 0.01 * (rows + hits), ev.eval_cache_last
    # the same evaluator told that it answered nothing: it steps aside, results unchanged
    ev.eval_cache_suspended = 15
    b = eng.play(ev, seeds)
    assert not eng.eval_cache and ev.eval_cache_suspended == 14
    assert np.array_equal(a.chosen, b.chosen) and np.array_equal(a.s_counts, b.s_counts)
    # ... and the rule itself, on the numbers it sees
    class Fake:
        eval_cache = True
        def eval_cache_stats(self): return (5, 1000, 0)
        def row_history(self, cap=0, reset=False): return (np.full(max(cap, 1), 100, np.int32), 10)
    ev2 = TorchNetEvaluator(net)
    ev2._rounds_at_bind = 0
    ev2.after_play(Fake())                           # 5 answers against 1,000 rows: below 1 %
    assert ev2.eval_cache_suspended == 15 and ev2.eval_cache_last == (5, 1000, 1000)
    ev_on = TorchNetEvaluator(net, eval_cache="on")
    ev_on._rounds_at_bind = 0
    ev_on.after_play(Fake())
    assert ev_on.eval_cache_suspended == 0
    eng.close()
    seeds = np.arange(256, dtype=np.uint32)
    ev = TorchNetEvaluator(_mate_line_net(2, 11))
    eng = SelfPlayEngine(256, sims=50, planes_format=ev.planes_format, max_moves=12)
    eng.play(ev, seeds)
    eng.play(ev, seeds)
    assert eng.eval_cache and ev.eval_cache_suspended == 0 and ev.eval_cache_last[0] > 0.1 * ev.eval_cache_last[2]
    eng.close()


def test_search_tree_equals_oracle_node_by_node(L):
    """the whole search tree - every MCTSNode's move, visit_count, value_sum, prior_prob and child
    range (self_play.py:19-28), not only the root's visit counts - equals the oracle's, bit for bit, on positions reached
    by seeded random play, at 15 / 50 / 200 simulations; and MCTS.search(..., return_root=True) hands the same tree out
    as MCTSNode objects whose select_child picks what the oracle's PUCT (pinned by the golden vectors) picks."""
    from chinesechessai_amd import _lib
    from chinesechessai_amd.engine import SelfPlayEngine, HashNetEvaluator
    from chinesechessai_amd.self_play import MCTS, MCTSNode
    from chinesechessai_amd import ChineseChess
    from chinesechessai_amd.chess_env import decode_move
    from oracle import xq_oracle as xo

    rng = np.random.RandomState(20261005)
    envs, lines = [], []
    for k in range(12):
        oe = xo.OracleEnv()
        line = []
        for _ in range(3 * k):
            mv = oe.legal_moves()
            if not mv or oe.e.winner != xo.WINNER_NONE:
                break
            m = int(mv[rng.randint(len(mv))])
            line.append(m)
            oe.make_move(m)
        if oe.e.winner == xo.WINNER_NONE and oe.legal_moves():
            envs.append(oe); lines.append(line)
    n = len(envs)
    assert n >= 10
    boards = np.stack([np.array(oe.board(), dtype=np.int8).reshape(90) for oe in envs])
    st = np.zeros((n, _lib.STATE_WORDS), np.int32)
    for i, oe in enumerate(envs):
        e = oe.e
        st[i, _lib.S_PLAYER], st[i, _lib.S_MOVE_COUNT], st[i, _lib.S_WINNER] = e.current_player, e.move_count, _lib.WINNER_NONE
        st[i, _lib.S_RED_KING], st[i, _lib.S_BLACK_KING], st[i, _lib.S_NO_CAPTURE] = e.red_king, e.black_king, e.no_capture_count
    total = 0
    for S in (15, 50, 200):
        eng = SelfPlayEngine(n, sims=S)
        ev = HashNetEvaluator()
        ev.bind(eng)
        eng.set_roots(boards, st)
        eng.search(ev)
        for i, oe in enumerate(envs):
            ref = oe.search_tree(S)
            got = eng.read_tree(i)
            m = len(ref["move"])
            assert got["root"] == 0 and len(got["move"]) == m, (S, i, len(got["move"]), m)
            assert np.array_equal(got["visit_count"].astype(np.int64), ref["visit_count"]), (S, i)
            assert np.array_equal(got["value_sum"].view(np.uint64), ref["value_sum"].view(np.uint64)), (S, i)
            assert np.array_equal(got["prior"][1:].view(np.uint32), ref["prior"][1:].view(np.uint32)), (S, i)
            assert np.array_equal(got["move"][1:], ref["move"][1:]), (S, i)
            assert np.array_equal(got["n_child"].astype(np.int64), ref["n_child"]), (S, i)
            exp = ref["n_child"] > 0
            assert np.array_equal(got["first_child"].astype(np.int64)[exp], ref["first_child"][exp]), (S, i)
            assert int(ref["visit_count"][0]) == S
            total += m
        eng.close()
    print("search trees compared node by node with the oracle's: %d positions x 3 simulation counts, %d nodes" % (n, total))

    # the mirror's MCTSNode view of the same arena
    env = ChineseChess()
    for m in lines[4]:
        env.make_move(decode_move(m))
    mcts = MCTS(HashNetEvaluator(), num_simulations=50)
    visits, root = mcts.search(env, return_root=True)
    ref = envs[4].search_tree(50)
    assert isinstance(root, MCTSNode) and root.parent is None and root.move is None and root.visit_count == 50
    assert list(root.children) == list(visits) and [c.visit_count for c in root.children.values()] == list(visits.values())
    f = int(ref["first_child"][0])
    for j, (mv, ch) in enumerate(root.children.items()):
        assert ch.parent is root and ch.move == mv == decode_move(int(ref["move"][f + j]))
        assert isinstance(ch.prior_prob, np.float32) and ch.prior_prob == ref["prior"][f + j]
        assert ch.value_sum == ref["value_sum"][f + j] and ch.is_leaf() == (ref["n_child"][f + j] == 0)
    # select_child: the oracle's float32-stepwise PUCT (pinned by the reference's golden search vectors) picks the same child
    lib = xo.lib()
    def walk(node, idx, depth=0):
        if node.is_leaf():
            return 0
        fc, nc = int(ref["first_child"][idx]), int(ref["n_child"][idx])
        scores = [lib.xqo_puct_score(float(ref["value_sum"][fc + j]), int(ref["visit_count"][fc + j]), float(ref["prior"][fc + j]),
                                     int(ref["visit_count"][idx])) for j in range(nc)]
        best = int(np.argmax(np.array(scores, dtype=np.float32)))            # (argmax: the first maximum, like the strict '>')
        mv, ch = node.select_child()
        assert mv == decode_move(int(ref["move"][fc + best])) and ch is list(node.children.values())[best], (depth, idx)
        return 1 + sum(walk(c, fc + j, depth + 1) for j, c in enumerate(node.children.values()) if not c.is_leaf())
    assert walk(root, 0) >= 5
    # expand() adds only missing moves; update() flips the sign on the way up (host copy only)
    leaf = next(c for c in root.children.values() if c.is_leaf())
    n0, w0, rn, rw = leaf.visit_count, leaf.value_sum, root.visit_count, root.value_sum
    leaf.expand({(0, 0, 1, 0): np.float32(0.5)}); leaf.expand({(0, 0, 1, 0): np.float32(0.25), (0, 0, 2, 0): np.float32(0.25)})
    assert [c.prior_prob for c in leaf.children.values()] == [np.float32(0.5), np.float32(0.25)]
    leaf.update(0.5)
    assert (leaf.visit_count, leaf.value_sum, root.visit_count, root.value_sum) == (n0 + 1, w0 + 0.5, rn + 1, rw - 0.5)


@pytest.mark.parametrize("ext", ["noise+schedule", "virtual_loss", "tree_reuse", "all"])
def test_eval_cache_is_result_identical_under_the_search_extensions(L, ext):
    """The evaluation cache is keyed by the position alone, so it must not care how the tree above a leaf is searched: under
    Dirichlet root noise + a per-ply temperature schedule (BASELINE C5's settings), virtual loss (8 pending leaves per game
    and round: the generic kernel path, 8 slots per game), tree reuse, and all of them together, 256 games x S = 48 with the
    mate-line network give the same records with the cache as without it; the cache must really answer; and its verify
    mode finds every answer equal to a fresh evaluation."""
    import torch
    from chinesechessai_amd import distributed as xd
    from chinesechessai_amd.engine import SelfPlayEngine, TorchNetEvaluator
    G, S = 256, 48
    net = _mate_line_net(2, 11)
    seeds = (np.arange(G, dtype=np.uint32) * 7 + 3).astype(np.uint32)
    block = 70 * xd.RECORD_BYTES

    def run(cache):
        ev = TorchNetEvaluator(net, leaf_dedupe=True, eval_cache=cache)
        eng = SelfPlayEngine(G, sims=S, planes_format=ev.planes_format, max_moves=24)
        sched = None
        if ext in ("noise+schedule", "all"):
            eng.set_root_noise(0.3, 0.25, seed=99)
            sched = lambda ply: 1.0 if ply < 10 else 0.001
        if ext in ("tree_reuse", "all"):
            eng.set_tree_reuse(True)
        if ext in ("virtual_loss", "all"):
            eng.set_virtual_loss(True)
        eng.play(ev, seeds, read=False, temperature_schedule=sched)
        t = torch.zeros(G * block, dtype=torch.uint8, device="cuda")
        eng.pack_samples(t.data_ptr())
        out = eng.read_game_outcomes()
        stats = eng.eval_cache_stats() if cache else (0, 0, 0)
        eng.close()
        return xd.records_to_numpy(t).reshape(G, 70), out, stats

    ref, ref_out, _ = run(False)
    assert int(ref_out["error"].sum()) == 0
    rec, out, (hits, fills, _) = run("on")
    for k in ("winner", "reason", "reason_side", "reason_count", "n_plies", "n_samples", "error"):
        assert np.array_equal(out[k], ref_out[k]), (ext, k)
    for g in range(G):
        _same_game(rec[g], ref[g], (ext, g))
    assert hits > 0 and fills > 0, (ext, hits, fills)
    rec_v, out_v, (compared, fills_v, bad) = run("verify")
    assert bad == 0 and compared > 0, (ext, compared, bad)
    for g in range(G):
        _same_game(rec_v[g], ref[g], (ext, "verify", g))
    print("eval cache under %s: %d hits, %d fills; verify mode compared %d answers, 0 mismatches" % (ext, hits, fills, compared))
