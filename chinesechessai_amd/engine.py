"""Host driver of the batched self-play engine (one process per GPU, G games in lock-step).

Per ply:  R = ceil(sims / 8) rounds of { tree kernel (select / rules / terminal backups, leaf
planes) -> evaluator }  then  { consume, pi, np.random.choice, make_move }  — the loop of
MCTS.search (self_play.py:103-148) and self_play_game (self_play.py:203-256) for all games at
once.  Everything stays on the device and on one HIP stream; Python only enqueues.

Evaluators:
  HashNetEvaluator   exact dyadic priors/values computed by a HIP kernel (parity tests)
  TorchNetEvaluator  InferenceNet under PyTorch-ROCm (the production leaf evaluator)
  CallbackEvaluator  any object with the reference's predict_batch (duck-typed, self_play.py:143)
"""
import ctypes as C

import numpy as np

from . import _lib
from .chess_env import decode_move, format_end_reason


class HashNetEvaluator:
    """SURVEY.md Appendix B evaluator, on the GPU."""
    planes_format = _lib.PLANES_NONE
    deterministic = True            # same position -> same priors, whatever the row: root evaluations may be carried over

    def __init__(self, salt=0):
        self.salt = salt

    def bind(self, engine):
        pass

    def planes_ptr(self):
        return None

    def evaluate(self, engine):
        _lib.check(engine.L.xq_engine_eval_hashnet(engine.h, self.salt))
        return _lib.EVAL_PRIORS, engine.priors_ptr, engine.values_ptr


class CallbackEvaluator:
    """Reference-style network object: predict_batch([(board, player, legal_moves)]) ->
    [(dict move->prior, value)] (neural_network.py:96-126).  Rows are passed with the same
    multiplicity the reference passes them (self_play.py:139-143)."""
    planes_format = _lib.PLANES_NONE
    deterministic = False           # a caller's network may be stochastic: every root is evaluated afresh, as the reference does

    def __init__(self, network):
        self.network = network

    def bind(self, engine):
        G = engine.n_games
        self.boards = np.zeros((G, 90), np.int8)
        self.player = np.zeros(G, np.int32)
        self.moves = np.zeros((G, _lib.MAX_MOVES), np.uint16)
        self.n_moves = np.zeros(G, np.int32)
        self.mult = np.zeros(G, np.int32)
        self.priors = np.zeros((G, _lib.MAX_MOVES), np.float32)
        self.values = np.zeros(G, np.float64)

    def planes_ptr(self):
        return None

    def evaluate(self, engine):
        L = engine.L
        _lib.check(L.xq_engine_read_leaves(engine.h, _lib.ptr(self.boards), _lib.ptr(self.player), _lib.ptr(self.moves),
                                            _lib.ptr(self.n_moves), _lib.ptr(self.mult)))
        for g in range(engine.n_games):
            m = int(self.mult[g])
            if m == 0:
                continue
            legal = [decode_move(x) for x in self.moves[g, :self.n_moves[g]]]
            row = (self.boards[g].reshape(10, 9).copy(), int(self.player[g]), legal)
            res = self.network.predict_batch([row] * m)
            probs, value = res[0]
            self.priors[g, :len(legal)] = [np.float32(probs[mv]) for mv in legal]
            self.values[g] = float(value)
        _lib.check(L.xq_engine_write_priors(engine.h, _lib.ptr(self.priors), _lib.ptr(self.values)))
        return _lib.EVAL_PRIORS, engine.priors_ptr, engine.values_ptr


class TorchNetEvaluator:
    """InferenceNet under PyTorch-ROCm; the search kernel writes its input planes in place.  On the hand-written
    single-launch path the evaluator runs with row compaction: only slots with a pending leaf are network rows
    (SelfPlayEngine.set_row_compaction), and pending leaves that are the same position share one row
    (SelfPlayEngine.set_leaf_dedupe; `leaf_dedupe=False` evaluates every pending leaf like the reference does)."""
    deterministic = True

    def __init__(self, net, dtype=None, channels_last=True, chunk=None, policy_columns="reachable", fused_tower=True,
                 leaf_dedupe=True, eval_cache=True):
        import torch
        from .neural_network import InferenceNet
        self.torch = torch
        dtype = dtype or torch.bfloat16
        self.dtype = dtype
        if dtype == torch.bfloat16:
            self.planes_format = _lib.PLANES_NHWC16_BF16 if channels_last else _lib.PLANES_NCHW_BF16
        elif dtype == torch.float32:
            self.planes_format = _lib.PLANES_NCHW_F32
            channels_last = False
        else:
            raise ValueError("dtype must be bfloat16 or float32")
        self.channels_last = channels_last
        # bf16 channels-last = the hand-written kernels; float32 (or NCHW) = PyTorch's library kernels, asked
        # for explicitly by choosing that dtype / layout (parity runs against the fp32 reference)
        self.inet = net if isinstance(net, InferenceNet) else InferenceNet(
            net, dtype=dtype, c_in=16 if channels_last else 15, device="cuda", policy_columns=policy_columns,
            fused_tower=fused_tower, allow_library_fallback=(dtype != torch.bfloat16 or not channels_last))
        self.chunk = chunk
        self.kind = _lib.EVAL_LOGITS_BF16 if dtype == torch.bfloat16 else _lib.EVAL_LOGITS_F32
        self.row_compaction = bool(self.inet.supports_row_map and not chunk)
        # the hand-written kernels' output for a position is one fixed fp32 chain per element, whatever the row,
        # the batch size and the launch: equal positions may share a row
        self.leaf_dedupe = bool(leaf_dedupe and self.row_compaction)
        # ... and a position evaluated during the last two plies - by any game - need not be evaluated again
        # (SelfPlayEngine.set_eval_cache; `eval_cache=False` evaluates every leaf the reference would)
        # eval_cache: True = on, and it watches itself: its probe costs the tree kernel a few per cent and an answered leaf
        # saves one network row.  With random-init weights at 16,384 games it answers ~2 % of the rows (transpositions of the
        # opening plies) and the step gets ~1 % faster (measured, round 5, once likely dedupe duplicates stopped queueing on
        # one cache entry); a trained network's line-following search: 40-60 % of the rows.  A whole play() in which it
        # answered fewer leaves than 1 % of the rows that were evaluated suspends it for the next 15 plays with this
        # evaluator, then it tries again.  "on" = always on; False = off; "verify" = leaves the cache could answer are
        # evaluated all the same and compared with its entry
        self.eval_cache_verify = eval_cache == "verify"
        self.eval_cache = bool(eval_cache and self.row_compaction)
        self.eval_cache_adaptive = eval_cache is True
        self.eval_cache_suspended = 0            # plays left without the cache
        self.eval_cache_last = None              # (hits, fills, rows evaluated) of the last play that used it
        self._rounds_at_bind = 0
        self.row_src = self.n_rows_dev = None

    def bind(self, engine):
        torch = self.torch
        engine.set_row_compaction(self.row_compaction)
        if self.row_compaction:
            engine.set_leaf_dedupe(self.leaf_dedupe)
            use = self.eval_cache and self.eval_cache_suspended == 0
            engine.set_eval_cache(use, verify=self.eval_cache_verify)
            if use:
                engine.eval_cache_stats(reset=True)
                self._rounds_at_bind = engine.row_history(cap=0)[1]
        self.row_src, self.n_rows_dev = engine.row_map()
        G = engine.n_rows                                       # one row per pending-leaf slot
        if self.channels_last:
            self.storage = torch.zeros((G, 10, 9, 16), dtype=self.dtype, device="cuda")
            self.x = self.storage.permute(0, 3, 1, 2)            # logical NCHW, channels-last strides
        else:
            self.storage = torch.zeros((G, 15, 10, 9), dtype=self.dtype, device="cuda")
            self.x = self.storage
        self.logits = torch.empty((G, self.inet.n_policy), dtype=self.dtype, device="cuda")
        cm = self.inet.column_map
        _lib.check(engine.L.xq_engine_set_logit_columns(engine.h, _lib.ptr(cm) if cm is not None else None,
                                                        self.inet.n_policy if cm is not None else 0))
        self.values = torch.empty((G,), dtype=self.dtype, device="cuda")

    def planes_ptr(self):
        return self.storage.data_ptr()

    def after_play(self, engine):
        """called by SelfPlayEngine.play / play_refill when the games are over: the evaluation cache's self-assessment"""
        if not self.eval_cache:
            return
        if engine.eval_cache:
            hits, fills, _ = engine.eval_cache_stats()
            n_now = engine.row_history(cap=0)[1]
            played = n_now - self._rounds_at_bind if n_now >= self._rounds_at_bind else n_now     # (the caller may have reset the count)
            rows = int(engine.row_history(cap=max(played, 1))[0].astype(np.int64).sum()) if played > 0 else 0
            self.eval_cache_last = (hits, fills, rows)
            if self.eval_cache_adaptive and hits < 0.01 * (rows + hits):
                self.eval_cache_suspended = 15
        elif self.eval_cache_suspended > 0:
            self.eval_cache_suspended -= 1

    def evaluate(self, engine):
        G = engine.n_rows
        if self.row_compaction:
            self.inet(self.x, out_logits=self.logits, out_values=self.values, row_src=self.row_src, n_rows=self.n_rows_dev)
            return self.kind, self.logits.data_ptr(), self.values.data_ptr()
        step = self.chunk or G
        for s in range(0, G, step):
            self.inet(self.x[s:s + step], out_logits=self.logits[s:s + step], out_values=self.values[s:s + step])
        return self.kind, self.logits.data_ptr(), self.values.data_ptr()


class GameBatch:
    """Results of one batch of games, in the reference's vocabulary."""

    def __init__(self, n_games, temperature):
        self.n_games = n_games
        self.temperature = temperature

    def game_data(self, g):
        """[(board int8[10,9], {move: np.float64 prob}, z float)] as self_play_game returns it
        (self_play.py:234-239, 310)."""
        out = []
        for i in range(int(self.n_samples[g])):
            n = int(self.s_n[g, i])
            moves = [decode_move(m) for m in self.s_moves[g, i, :n]]
            counts = self.s_counts[g, i, :n].astype(np.int64)
            temp = self.temperature
            pt = getattr(self, "ply_temperature", None)
            if pt:                                            # per-ply schedule (extension)
                ply = 2 * i if getattr(self, "opponent_mode", False) else i
                if ply < len(pt):
                    temp = pt[ply]
            if temp < 0.01:                                   # self_play.py:224-227
                probs = np.zeros(n)
                probs[np.argmax(counts)] = 1
            else:                                             # self_play.py:230-231
                c = counts ** (1.0 / temp)
                probs = c / c.sum()
            out.append((self.s_board[g, i].reshape(10, 9).copy(), {m: p for m, p in zip(moves, probs)},
                        float(self.s_z[g, i])))
        return out

    def end_reason(self, g):
        s = format_end_reason(int(self.reason[g]), int(self.reason_side[g]), int(self.reason_count[g]))
        return s if s else "未知原因"                           # self_play.py:260

    def results(self):
        """[(game_data, winner, end_reason)] — the return value of parallel_self_play
        (self_play.py:469); failed games (np.random.choice ValueError) are dropped like the
        reference drops (None, None, None) results (self_play.py:415-416)."""
        return [(self.game_data(g), int(self.winner[g]), self.end_reason(g))
                for g in range(self.n_games) if self.error[g] == 0]


class SelfPlayEngine:
    def __init__(self, n_games, sims=50, temperature=1.0, max_moves=70, opponent_mode=False,
                 planes_format=_lib.PLANES_NONE, device=0, leaf_batch=8, stream=None):
        self.L = _lib.lib()
        if self.L.xq_device_count() <= 0:
            raise _lib.XqError("no HIP device visible: the engine has no CPU fallback")
        self.n_games, self.sims, self.temperature, self.max_moves = n_games, sims, temperature, max_moves
        self.opponent_mode = opponent_mode
        cfg = _lib.Config(n_games, sims, leaf_batch, max_moves, float(temperature), 1 if opponent_mode else 0,
                          planes_format, device, 0)
        h = C.c_void_p()
        _lib.check(self.L.xq_engine_create(C.byref(cfg), C.byref(h)))
        self.h = h
        self.rounds = self.L.xq_engine_rounds_per_move(h)
        self.priors_ptr = self.L.xq_engine_priors_ptr(h)
        self.values_ptr = self.L.xq_engine_values_ptr(h)
        self.n_rows = n_games                                   # evaluator rows: n_games * leaf slots
        # root evaluation carry-over: None = automatic (on whenever it is result-identical: one deterministic
        # evaluator, no noise / virtual loss / tree reuse / arena mode), True / False = the caller's choice
        self.root_eval_carry = None
        self._carry_on = False
        self.row_compaction = False
        self.leaf_dedupe = False
        self.eval_cache = False
        self._noise = self._vloss = False
        self.tree_reuse = False
        if stream is not None:
            _lib.check(self.L.xq_engine_set_stream(h, C.c_void_p(stream)))
        if temperature >= 0.01 and temperature != 1.0:
            # counts ** (1/T) exactly as the host's numpy evaluates it (self_play.py:230)
            tab = np.arange(sims + 1, dtype=np.int64) ** (1.0 / temperature)
            tab = np.ascontiguousarray(tab, dtype=np.float64)
            _lib.check(self.L.xq_engine_set_pow_table(h, _lib.ptr(tab), len(tab)))

    def close(self):
        if getattr(self, "h", None):
            self.L.xq_engine_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- low-level steps ------------------------------------------------------------------
    def new_games(self, seeds):
        seeds = np.ascontiguousarray(seeds, dtype=np.uint32)
        assert seeds.shape == (self.n_games,)
        _lib.check(self.L.xq_engine_new_games(self.h, _lib.ptr(seeds)))

    def set_uniforms(self, u):
        u = np.ascontiguousarray(u, dtype=np.float64)
        assert u.shape == (self.n_games, _lib.MAX_PLIES)
        _lib.check(self.L.xq_engine_set_uniforms(self.h, _lib.ptr(u)))

    def set_roots(self, boards, states):
        boards = np.ascontiguousarray(boards, dtype=np.int8).reshape(self.n_games, 90)
        states = np.ascontiguousarray(states, dtype=np.int32).reshape(self.n_games, _lib.STATE_WORDS)
        _lib.check(self.L.xq_engine_set_roots(self.h, _lib.ptr(boards), _lib.ptr(states)))

    def set_root_eval_carry(self, enable=True):
        """Result-identical work elimination: the played child's network evaluation becomes the next root's instead
        of being computed a second time (the reference rebuilds its tree every ply, self_play.py:98), so round 0 of
        a ply has nothing to evaluate.  play() / play_refill() switch it on by themselves whenever it is
        result-identical AND free: one deterministic evaluator that runs with row compaction (TorchNetEvaluator on the
        hand-written kernels: a carried root simply has no row in round 0), no root noise / virtual loss / tree reuse /
        arena mode.  Evaluators without row compaction (HashNetEvaluator, CallbackEvaluator, a chunked TorchNetEvaluator)
        do NOT get it automatically - skipping round 0 would cost them a blocking read per ply - only by this call;
        this call fixes the choice: True insists (and raises for a combination it cannot serve), False evaluates
        every root afresh like the reference.  None returns to automatic.  Call before new games start."""
        if enable is None:
            self.root_eval_carry = None
            return
        _lib.check(self.L.xq_engine_set_root_eval_carry(self.h, 1 if enable else 0))
        self.root_eval_carry = self._carry_on = bool(enable)

    def _auto_carry(self, evaluator, opponent_evaluator):
        """Decide the carry-over for the games about to start (play / play_refill)."""
        if self.root_eval_carry is not None:
            if self.root_eval_carry and opponent_evaluator is not None:
                raise _lib.XqError("root evaluation carry-over needs one network for both sides (the carried priors are the mover's network's)")
            return self._carry_on
        # automatic only where a carried root costs nothing by itself: with row compaction it simply has no row in round
        # 0.  Without compaction skipping round 0 takes a blocking read of roots_not_ready every ply (the host would stop
        # running ahead of the GPU), which nobody asked for: those evaluators get the carry-over by set_root_eval_carry(True)
        want = bool(getattr(evaluator, "deterministic", False) and getattr(evaluator, "row_compaction", False)
                    and opponent_evaluator is None and not self.opponent_mode
                    and not self._noise and not self._vloss and not self.tree_reuse)
        if want != self._carry_on:
            _lib.check(self.L.xq_engine_set_root_eval_carry(self.h, 1 if want else 0))
            self._carry_on = want
        return want

    def _drop_auto_carry(self):
        # an extension is being switched on: the automatic carry-over steps aside (an explicit one refuses in C)
        if self._carry_on and self.root_eval_carry is None:
            _lib.check(self.L.xq_engine_set_root_eval_carry(self.h, 0))
            self._carry_on = False

    def set_row_compaction(self, enable=True):
        """Only slots with a pending leaf become evaluator rows (xq_engine_set_row_compaction); evaluators whose
        kernels take the row map switch it on when they are bound (TorchNetEvaluator on the hand-written path)."""
        _lib.check(self.L.xq_engine_set_row_compaction(self.h, 1 if enable else 0))
        self.row_compaction = bool(enable)
        if not enable:
            self.leaf_dedupe = False

    def set_leaf_dedupe(self, enable=True):
        """Pending leaves of a round that are the same position (board + side to move) share one evaluator row
        (xq_engine_set_leaf_dedupe; needs row compaction).  Result-identical for an evaluator whose output depends
        on the position only; the reference evaluates every leaf of every game (self_play.py:137-143)."""
        _lib.check(self.L.xq_engine_set_leaf_dedupe(self.h, 1 if enable else 0))
        self.leaf_dedupe = bool(enable)

    def set_eval_cache(self, enable=True, log2_entries=None, verify=False):
        """Evaluation cache (xq_engine_set_eval_cache): the evaluator's priors and value for a position are kept in HBM for
        two plies; a later pending leaf that is the same position - in any game - takes them from there instead of
        becoming an evaluator row.  What it removes: the reference rebuilds its tree every ply (self_play.py:98), so
        whatever the last ply's search expanded below the move that was played is evaluated again; with a trained
        (peaked) network that is most of a ply's rows, with random-init weights next to nothing.  Result-identical for an
        evaluator whose output depends on the position only; evaluators on the hand-written kernels switch it on when
        they are bound (`TorchNetEvaluator(eval_cache=False)` keeps it off).  The default size holds four plies of
        evaluations."""
        if not enable:
            _lib.check(self.L.xq_engine_set_eval_cache(self.h, 0))
            self.eval_cache = False
            return
        if log2_entries is None:
            want = 4 * self.rounds * self.n_rows
            log2_entries = min(22, max(12, int(np.ceil(np.log2(max(want, 2))))))
        # verify: leaves the cache could answer are evaluated all the same and compared with its entry (eval_cache_stats()[2])
        _lib.check(self.L.xq_engine_set_eval_cache(self.h, -int(log2_entries) if verify else int(log2_entries)))
        self.eval_cache = True

    def eval_cache_stats(self, reset=False):
        """(hits, fills, mismatches found in verify mode) of the evaluation cache since the last reset."""
        out = np.zeros(3, np.uint64)
        _lib.check(self.L.xq_engine_eval_cache_stats(self.h, _lib.ptr(out), 1 if reset else 0))
        return int(out[0]), int(out[1]), int(out[2])

    def row_map(self):
        """(row_src, row_count) device pointers as ints, or (None, None) without compaction."""
        a, b = C.c_void_p(), C.c_void_p()
        _lib.check(self.L.xq_engine_row_map(self.h, C.byref(a), C.byref(b)))
        return a.value, b.value

    def row_history(self, cap=65536, reset=False):
        """(rows, n): rows the evaluator had to run in each of the last min(cap, n, _lib.ROW_HISTORY) search rounds
        (oldest first; the engine keeps the last 65,536 rounds only) and n = rounds launched since the last reset.
        len(rows) < n means the history is incomplete."""
        cap = min(int(cap), _lib.ROW_HISTORY)
        rows = np.zeros(cap, np.int32)
        n = C.c_int64()
        _lib.check(self.L.xq_engine_read_row_history(self.h, _lib.ptr(rows), cap, C.byref(n), 1 if reset else 0))
        return rows[:min(cap, n.value)], n.value

    def leaf_rows(self):
        rows = np.zeros(self.n_rows, np.int32)
        _lib.check(self.L.xq_engine_read_leaf_rows(self.h, _lib.ptr(rows)))
        return rows

    def roots_not_ready(self):
        n = np.zeros(1, np.int32)
        _lib.check(self.L.xq_engine_roots_not_ready(self.h, _lib.ptr(n)))
        return int(n[0])

    def search(self, evaluator, skip_round0=False):
        """One MCTS.search for every game (self_play.py:89-154); root visits are final after it.
        skip_round0: every root already holds its expansion (set_root_eval_carry + roots_not_ready() == 0); with row
        compaction nobody needs to know: a carried root simply has no row in round 0."""
        kind, a, v = _lib.EVAL_PRIORS, None, None
        if skip_round0:
            a, v = self.priors_ptr, self.values_ptr          # nothing is pending: round 1 has nothing to consume
        pp = evaluator.planes_ptr()
        for r in range(1 if skip_round0 else 0, self.rounds):
            _lib.check(self.L.xq_engine_search_round(self.h, r, kind, a, v, pp))
            kind, a, v = evaluator.evaluate(self)
        _lib.check(self.L.xq_engine_end_search(self.h, kind, a, v))

    def root_visits(self):
        G = self.n_games
        moves = np.zeros((G, _lib.MAX_MOVES), np.uint16)
        visits = np.zeros((G, _lib.MAX_MOVES), np.int32)
        n = np.zeros(G, np.int32)
        _lib.check(self.L.xq_engine_read_root_visits(self.h, _lib.ptr(moves), _lib.ptr(visits), _lib.ptr(n)))
        return moves, visits, n

    def read_tree(self, game=0):
        """The search tree of `game` as it stands (after search(): the finished tree of the ply), node by node in creation
        order: dict of arrays visit_count, value_sum, prior, move, first_child, n_child (+ "root": the root's index) -
        MCTSNode's fields, self_play.py:19-28; a node's children are [first_child, first_child + n_child)."""
        nn, root = np.zeros(1, np.int32), np.zeros(1, np.int32)
        _lib.check(self.L.xq_engine_read_tree(self.h, int(game), 0, _lib.ptr(nn), _lib.ptr(root), *([None] * 6)))
        n = int(nn[0])
        t = {"visit_count": np.zeros(n, np.uint32), "value_sum": np.zeros(n, np.float64), "prior": np.zeros(n, np.float32),
             "move": np.zeros(n, np.uint16), "first_child": np.zeros(n, np.uint16), "n_child": np.zeros(n, np.uint8)}
        _lib.check(self.L.xq_engine_read_tree(self.h, int(game), n, _lib.ptr(nn), _lib.ptr(root),
                                              *[_lib.ptr(t[k]) for k in ("visit_count", "value_sum", "prior", "move", "first_child", "n_child")]))
        t["root"] = int(root[0])
        return t

    def active_games(self):
        n = np.zeros(1, np.int32)
        _lib.check(self.L.xq_engine_active_games(self.h, _lib.ptr(n)))
        return int(n[0])

    def active_games_post(self):
        """enqueue the count of games still playing (non-blocking; see active_games_poll)"""
        _lib.check(self.L.xq_engine_active_games_post(self.h))

    def active_games_poll(self):
        """the newest posted count if it has arrived, else None - never waits for the GPU"""
        n, ready = np.zeros(1, np.int32), np.zeros(1, np.int32)
        _lib.check(self.L.xq_engine_active_games_poll(self.h, _lib.ptr(n), _lib.ptr(ready)))
        return int(n[0]) if ready[0] else None

    # ---- whole games ------------------------------------------------------------------------
    def set_temperature(self, temperature):
        """Per-ply temperature (extension; the reference uses one temperature per game)."""
        tab = None
        if temperature >= 0.01 and temperature != 1.0:
            top = 65536 if self.tree_reuse else self.sims + 1                        # carried visits exceed sims
            tab = np.ascontiguousarray(np.arange(top, dtype=np.int64) ** (1.0 / temperature), dtype=np.float64)
        _lib.check(self.L.xq_engine_set_temperature(self.h, float(temperature), _lib.ptr(tab), 0 if tab is None else len(tab)))

    def set_root_noise(self, alpha, epsilon, seed=0):
        """Dirichlet root noise (extension, BASELINE C5): root priors (1-eps) P + eps Dir(alpha)."""
        if epsilon > 0:
            self._drop_auto_carry()
        _lib.check(self.L.xq_engine_set_root_noise(self.h, float(alpha), float(epsilon), int(seed)))
        self._noise = epsilon > 0

    def set_tree_reuse(self, enable=True):
        """Keep the played move's subtree as the next ply's tree (extension; the reference builds a
        fresh tree every ply).  Call before new games are started."""
        if enable:
            self._drop_auto_carry()
        _lib.check(self.L.xq_engine_set_tree_reuse(self.h, 1 if enable else 0))
        self.tree_reuse = bool(enable)
        self.set_temperature(self.temperature)             # counts ** (1/T) table must cover carried visits

    def set_virtual_loss(self, enable=True):
        """A round's simulations spread over up to leaf_batch distinct leaves, each evaluated once
        (extension; the reference's rounds all reach one leaf).  The evaluator then works on
        n_rows = n_games * leaf_batch rows.  Call before new games are started and before an
        evaluator is bound."""
        if enable:
            self._drop_auto_carry()
        _lib.check(self.L.xq_engine_set_virtual_loss(self.h, 1 if enable else 0))
        self._vloss = bool(enable)
        self.n_rows = self.n_games * self.L.xq_engine_leaf_slots(self.h)
        self.priors_ptr = self.L.xq_engine_priors_ptr(self.h)
        self.values_ptr = self.L.xq_engine_values_ptr(self.h)

    def tree_stats(self):
        """(nodes in each game's arena, pending virtual-loss visits) — xq_engine_tree_stats."""
        n, v = np.zeros(self.n_games, np.int32), np.zeros(self.n_games, np.int32)
        _lib.check(self.L.xq_engine_tree_stats(self.h, _lib.ptr(n), _lib.ptr(v)))
        return n, v

    def refill_slots(self):
        """Game id every slot holds during a refill session (-1: retired) — xq_engine_refill_read_slots."""
        s = np.zeros(self.n_games, np.int32)
        _lib.check(self.L.xq_engine_refill_read_slots(self.h, _lib.ptr(s)))
        return s

    def root_priors(self):
        p = np.zeros((self.n_games, _lib.MAX_MOVES), np.float32)
        _lib.check(self.L.xq_engine_read_root_priors(self.h, _lib.ptr(p)))
        return p

    @staticmethod
    def _row_layout(ev):
        """(row compaction, leaf dedupe, policy columns) an evaluator hands its logits in; None = it fills priors by slot
        (XQ_EVAL_PRIORS), which every layout serves."""
        if getattr(ev, "planes_format", _lib.PLANES_NONE) == _lib.PLANES_NONE and not hasattr(ev, "row_compaction"):
            return None
        inet = getattr(ev, "inet", None)
        return (bool(getattr(ev, "row_compaction", False)), bool(getattr(ev, "leaf_dedupe", False)),
                getattr(inet, "n_policy", None))

    def _bind(self, evaluator, opponent_evaluator=None):
        """Bind the evaluator(s) of the games about to start.  The engine's row layout is reset first: an evaluator that
        hands its logits in by slot (no `row_compaction` attribute) must never meet the compaction an earlier
        TorchNetEvaluator.bind left behind (the C side refuses that combination too); those that want compaction switch
        it on in their bind.  Both evaluators of a match work on ONE engine, so two network evaluators must ask for
        the same layout and policy columns."""
        la, lb = self._row_layout(evaluator), None if opponent_evaluator is None else self._row_layout(opponent_evaluator)
        if la is not None and lb is not None and la != lb:
            raise _lib.XqError("the two evaluators of a match must use the same row layout and policy columns "
                               "(same dtype / layout / chunk / policy_columns / leaf_dedupe)")
        self.set_row_compaction(False)
        self.set_eval_cache(False)
        for ev in (evaluator, opponent_evaluator):
            if ev is not None and self._row_layout(ev) is None:
                ev.bind(self)                 # (by-slot priors first: they do not touch the layout)
        for ev in (evaluator, opponent_evaluator):
            if ev is not None and self._row_layout(ev) is not None:
                ev.bind(self)
        if opponent_evaluator is not None:
            self.set_eval_cache(False)        # two networks answer differently for one position

    def play(self, evaluator, seeds, opponent_evaluator=None, uniforms=None, check_every=8, read=True,
             temperature_schedule=None):
        """Play G games to the end (self_play_game for every game).  `seeds[g]` seeds game g's
        MT19937 stream like np.random.seed; `uniforms` overrides the streams.
        `temperature_schedule(ply) -> T` (extension) overrides the constant temperature per ply."""
        carry = self._auto_carry(evaluator, opponent_evaluator)
        self._bind(evaluator, opponent_evaluator)
        self.new_games(seeds)
        if uniforms is not None:
            self.set_uniforms(uniforms)
        self.ply_temperature = []
        cur_t = None
        skip0 = False
        for ply in range(min(self.max_moves, _lib.MAX_PLIES)):
            t = self.temperature if temperature_schedule is None else float(temperature_schedule(ply))
            if temperature_schedule is not None and t != cur_t:
                self.set_temperature(t)
                cur_t = t
            self.ply_temperature.append(t)
            ev = evaluator if (ply % 2 == 0 or opponent_evaluator is None) else opponent_evaluator   # self_play.py:211
            self.search(ev, skip_round0=skip0)
            _lib.check(self.L.xq_engine_play_move(self.h))
            skip0 = carry and not self.row_compaction and self.roots_not_ready() == 0
            if check_every and ply % check_every == check_every - 1:
                # "are all games over?" without stopping for the answer: the count posted a few plies ago, if it has arrived
                # (plies of finished games cost next to nothing; the blocking form idled the GPU ~0.3 % of an epoch)
                if self.active_games_poll() == 0:
                    break
                self.active_games_post()
        _lib.check(self.L.xq_engine_finalize(self.h))
        for ev in (evaluator, opponent_evaluator):
            if ev is not None and hasattr(ev, "after_play"):
                ev.after_play(self)
        return self.read_results() if read else None

    def play_refill(self, evaluator, seeds, records_ptr, check_every=4, max_plies=None, on_ply=None):
        """`len(seeds)` games through the engine's G slots with refill (xq_engine_refill_*): a finished game's
        slot restarts on the next unplayed seed at once, like the reference's pool (self_play.py:404-408).
        `records_ptr`: device buffer of len(seeds) * 70 sample records (distributed.RECORD_BYTES each), filled
        by game id.  Returns the per-game outcome arrays (by game id) and the number of plies stepped.
        `on_ply(ply)`: called after every refill step (tests / monitoring; reading engine state there synchronises)."""
        seeds = np.ascontiguousarray(seeds, dtype=np.uint32)
        total = len(seeds)
        carry = self._auto_carry(evaluator, None)
        self._bind(evaluator)
        _lib.check(self.L.xq_engine_refill_begin(self.h, _lib.ptr(seeds), total))
        active = np.zeros(1, np.int32)
        plies = 0
        cap = max_plies or (total // self.n_games + 2) * _lib.MAX_PLIES + 8
        skip0 = False
        while plies < cap:
            self.search(evaluator, skip_round0=skip0)
            _lib.check(self.L.xq_engine_play_move(self.h))
            poll = plies % check_every == check_every - 1
            _lib.check(self.L.xq_engine_refill_step(self.h, C.c_void_p(records_ptr), _lib.ptr(active) if poll else None))
            skip0 = carry and not self.row_compaction and self.roots_not_ready() == 0   # (after the refill: restarted slots need their round 0)
            if on_ply is not None:
                on_ply(plies)
            plies += 1
            if poll and int(active[0]) == 0:
                break
        else:
            raise _lib.XqError("play_refill: games still active after %d plies" % cap)
        if hasattr(evaluator, "after_play"):
            evaluator.after_play(self)
        out = {k: np.zeros(total, np.int32) for k in ("winner", "reason", "reason_side", "reason_count", "n_plies",
                                                      "n_samples", "error")}
        _lib.check(self.L.xq_engine_refill_read_games(self.h, *[_lib.ptr(out[k]) for k in (
            "winner", "reason", "reason_side", "reason_count", "n_plies", "n_samples", "error")]))
        return out, plies

    def read_results(self):
        G, P, M = self.n_games, _lib.MAX_PLIES, _lib.MAX_MOVES
        b = GameBatch(G, self.temperature)
        b.ply_temperature = list(getattr(self, "ply_temperature", []))
        b.opponent_mode = self.opponent_mode
        for name in ("winner", "reason", "reason_side", "reason_count", "n_plies", "n_samples", "error"):
            setattr(b, name, np.zeros(G, np.int32))
        _lib.check(self.L.xq_engine_read_games(self.h, _lib.ptr(b.winner), _lib.ptr(b.reason), _lib.ptr(b.reason_side),
                                               _lib.ptr(b.reason_count), _lib.ptr(b.n_plies), _lib.ptr(b.n_samples),
                                               _lib.ptr(b.error)))
        b.s_board = np.zeros((G, P, 90), np.int8)
        b.s_player = np.zeros((G, P), np.int8)
        b.s_n = np.zeros((G, P), np.uint8)
        b.s_moves = np.zeros((G, P, M), np.uint16)
        b.s_counts = np.zeros((G, P, M), np.uint16)
        b.s_z = np.zeros((G, P), np.float64)
        b.chosen = np.zeros((G, P), np.uint16)
        b.step_reward = np.zeros((G, P), np.float64)
        _lib.check(self.L.xq_engine_read_samples(self.h, _lib.ptr(b.s_board), _lib.ptr(b.s_player), _lib.ptr(b.s_n),
                                                 _lib.ptr(b.s_moves), _lib.ptr(b.s_counts), _lib.ptr(b.s_z),
                                                 _lib.ptr(b.chosen), _lib.ptr(b.step_reward)))
        return b

    def read_game_outcomes(self):
        """Only the per-game scalars (cheap)."""
        G = self.n_games
        out = {k: np.zeros(G, np.int32) for k in ("winner", "reason", "reason_side", "reason_count", "n_plies",
                                                  "n_samples", "error")}
        _lib.check(self.L.xq_engine_read_games(self.h, *[_lib.ptr(out[k]) for k in (
            "winner", "reason", "reason_side", "reason_count", "n_plies", "n_samples", "error")]))
        return out

    def pack_samples(self, records_ptr):
        """Fixed-size device-resident sample records for the all-gather (distributed.py)."""
        _lib.check(self.L.xq_engine_pack_samples(self.h, C.c_void_p(records_ptr)))

    def profile(self, enable):
        """HIP events around the tree-kernel launches: False / 0 off, True / 1 every launch, N every N-th search round"""
        _lib.check(self.L.xq_engine_profile(self.h, int(enable)))

    def profile_read(self):
        sm, pm = C.c_double(), C.c_double()
        sn, pn = C.c_int64(), C.c_int64()
        _lib.check(self.L.xq_engine_profile_read(self.h, C.byref(sm), C.byref(sn), C.byref(pm), C.byref(pn)))
        return dict(search_ms=sm.value, search_launches=sn.value, play_ms=pm.value, play_launches=pn.value)
