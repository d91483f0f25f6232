"""Host mirror of the reference's self_play.py: MCTSNode, MCTS, self_play_game, parallel_self_play,
InterruptedWithResults — same names, arguments, return types and error behaviour
(SURVEY.md §8b), backed by the HIP engine (engine.py / libxq_hip.so).

`network` is anything with the reference's predict_batch (duck-typed, self_play.py:143); a
`chinesechessai_amd.neural_network.ChessNet` living on the GPU takes the fast path
(TorchNetEvaluator: planes written by the search kernel, logits consumed by the tree kernel).
"""
import math

import numpy as np

from . import _lib
from .chess_env import ChineseChess, decode_move, _sq
from .config import MAX_MOVES, MCTS_SIMULATIONS
from .engine import CallbackEvaluator, HashNetEvaluator, SelfPlayEngine, TorchNetEvaluator


class InterruptedWithResults(Exception):
    """self_play.py:12-16"""

    def __init__(self, results):
        self.results = results
        super().__init__("Training interrupted by user")


# Precision of the leaf evaluator when a CUDA eval-mode ChessNet is handed to the mirror API.  The reference
# evaluates in fp32 (neural_network.py:112-115); "bf16" (default) runs the hand-written MFMA kernels with
# eval-mode BatchNorm folded (priors within the tolerance tests/test_gpu_parity.py states next to its
# measured error), "f32" runs PyTorch's fp32 library kernels on the same device (parity runs).
INFERENCE_DTYPE = "bf16"
_warned_f32 = False


def _note_f32_once():
    """Said once per process, when the mirror API is first sent to fp32: that precision is the reference's
    (neural_network.py:112-115) but it is served by PyTorch's library kernels, not by the hand-written gfx950 path."""
    global _warned_f32
    if not _warned_f32:
        _warned_f32 = True
        import warnings
        warnings.warn("inference_dtype='f32': the leaf evaluator runs on PyTorch-ROCm library kernels (MIOpen / hipBLASLt), "
                      "not on the hand-written MI355X kernels (bf16 only); meant for parity runs against the reference's fp32",
                      RuntimeWarning, stacklevel=3)


def _evaluator_for(network, fast=True, inference_dtype=None):
    if isinstance(network, (HashNetEvaluator, CallbackEvaluator, TorchNetEvaluator)):
        return network
    if fast:
        try:
            import torch
            from .neural_network import ChessNet, InferenceNet
            if isinstance(network, InferenceNet):
                return TorchNetEvaluator(network, dtype=network.dtype, channels_last=network.c_in == 16)
            if isinstance(network, ChessNet) and next(network.parameters()).is_cuda and not network.training:
                name = inference_dtype or INFERENCE_DTYPE
                if name not in ("bf16", "f32"):
                    raise ValueError("inference_dtype must be 'bf16' or 'f32'")
                if name == "f32":
                    _note_f32_once()
                return TorchNetEvaluator(network, dtype=torch.bfloat16 if name == "bf16" else torch.float32)
        except ImportError:
            pass
    return CallbackEvaluator(network)


class MCTSNode:
    """self_play.py:19-80 - a node of the search tree with the reference's attribute names: `parent`, `move`, `children`
    ({move: MCTSNode} in legal-move order), `visit_count`, `value_sum`, `prior_prob`.  The engine keeps its trees in a
    device arena (csrc/xq_engine.hip: nN / nW / nP / nMove, one wavefront per game); MCTS.search(..., return_root=True) and
    MCTS.root hand out the finished tree of the last search as a host COPY made of these objects, so tools written against
    the reference's nodes can walk it.  expand() / update() act on that copy only."""
    __slots__ = ("parent", "move", "children", "visit_count", "value_sum", "prior_prob")

    def __init__(self, parent=None, move=None, prior_prob=0):
        self.parent, self.move, self.prior_prob = parent, move, prior_prob
        self.children = {}
        self.visit_count, self.value_sum = 0, 0

    def value(self):                                            # :30-34
        return self.value_sum / self.visit_count if self.visit_count else 0

    def is_leaf(self):                                          # :36-38
        return not self.children

    def select_child(self, c_puct=1.5):
        """(move, child) with the highest PUCT score, the first one on ties (:40-59).  prior_prob is an np.float32, so
        NumPy rounds every step to float32 exactly as it does in the reference."""
        best = (None, None)
        top = -float("inf")
        root_n = math.sqrt(self.visit_count)
        for mv, ch in self.children.items():
            u = ch.value() + c_puct * ch.prior_prob * root_n / (1 + ch.visit_count)
            if u > top:
                top, best = u, (mv, ch)
        return best

    def expand(self, move_probs):                               # :61-68: only moves that are not children yet
        for mv, p in move_probs.items():
            self.children.setdefault(mv, MCTSNode(self, mv, p))

    def update(self, value):                                    # :70-80: the sign flips at every level up
        node = self
        while node is not None:
            node.visit_count += 1
            node.value_sum += value
            value, node = -value, node.parent

    @staticmethod
    def from_arena(tree):
        """the host copy of a device tree (SelfPlayEngine.read_tree)"""
        n = len(tree["move"])
        nodes = [MCTSNode() for _ in range(n)]
        for i, nd in enumerate(nodes):
            nd.visit_count = int(tree["visit_count"][i])
            # (the reference's value_sum is the int 0 until the first update, a Python float afterwards)
            nd.value_sum = float(tree["value_sum"][i]) if nd.visit_count else 0
            f, c = int(tree["first_child"][i]), int(tree["n_child"][i])
            for j in range(f, f + c):
                ch = nodes[j]
                ch.parent, ch.move, ch.prior_prob = nd, decode_move(int(tree["move"][j])), np.float32(tree["prior"][j])
                nd.children[ch.move] = ch
        return nodes[tree["root"]] if n else MCTSNode()


class MCTS:
    """self_play.py:83-175.  search(env) returns {move: visit_count} over ALL root children in
    legal-move order (Appendix A13); the tree is rebuilt on every call (self_play.py:98)."""

    def __init__(self, network, num_simulations=None):
        self.network = network
        self.default_simulations = num_simulations if num_simulations else MCTS_SIMULATIONS
        self._engines = {}

    def _engine(self, sims):
        if sims not in self._engines:
            self._engines[sims] = SelfPlayEngine(1, sims=sims)
        return self._engines[sims]

    def search(self, env, num_simulations=None, return_root=False):
        """{move: visit_count} (self_play.py:151-154); return_root=True: (that, the finished tree's root as an MCTSNode)"""
        if num_simulations is None:
            num_simulations = self.default_simulations
        eng = self._engine(num_simulations)
        ev = CallbackEvaluator(self.network) if not isinstance(self.network, HashNetEvaluator) else self.network
        ev.bind(eng)
        st = np.zeros((1, _lib.STATE_WORDS), np.int32)          # _copy_env (self_play.py:156-175)
        st[0, _lib.S_PLAYER] = env.current_player
        st[0, _lib.S_MOVE_COUNT] = env.move_count
        st[0, _lib.S_WINNER] = _lib.WINNER_NONE if env.winner is None else env.winner
        st[0, _lib.S_RED_KING] = _sq(env.red_king_pos)
        st[0, _lib.S_BLACK_KING] = _sq(env.black_king_pos)
        st[0, _lib.S_NO_CAPTURE] = env.no_capture_count
        eng.set_roots(np.ascontiguousarray(env.board, dtype=np.int8).reshape(1, 90), st)
        eng.search(ev)
        moves, visits, n = eng.root_visits()
        self._last = eng
        out = {decode_move(moves[0, j]): int(visits[0, j]) for j in range(int(n[0]))}
        return (out, self.root) if return_root else out

    @property
    def root(self):
        """the finished tree of the last search() as MCTSNode objects (a host copy of the device arena)"""
        if getattr(self, "_last", None) is None:
            raise RuntimeError("MCTS.root: no search has run yet")
        return MCTSNode.from_arena(self._last.read_tree(0))


def _play_batch(network, num_games, temperature, num_simulations, opponent_network, seeds=None, uniforms=None,
                inference_dtype=None, partial_on_interrupt=False):
    sims = num_simulations if num_simulations else MCTS_SIMULATIONS
    ev = _evaluator_for(network, inference_dtype=inference_dtype)
    ev_b = _evaluator_for(opponent_network, inference_dtype=inference_dtype) if opponent_network is not None else None
    if ev_b is not None and getattr(ev_b, "planes_format", 0) != getattr(ev, "planes_format", 0):
        raise ValueError("both networks must use the same planes format")      # (before any engine exists)
    eng = SelfPlayEngine(num_games, sims=sims, temperature=temperature, max_moves=MAX_MOVES,
                         opponent_mode=opponent_network is not None,
                         planes_format=getattr(ev, "planes_format", _lib.PLANES_NONE))
    if seeds is None:
        seeds = np.random.randint(0, 2 ** 31 - 1, size=num_games).astype(np.uint32)
    try:
        return eng.play(ev, seeds, opponent_evaluator=ev_b, uniforms=uniforms)
    except KeyboardInterrupt:
        if not partial_on_interrupt:
            raise
        # self_play.py:436-452: hand back the games that had finished when Ctrl-C arrived
        raise InterruptedWithResults(_finished_games(eng))
    finally:
        eng.close()


def _finished_games(eng):
    """Results of the games of a batch in flight that are already over (terminal position or move cap): the
    z table is applied to every slot, unfinished games are filtered out."""
    try:
        _lib.check(eng.L.xq_engine_finalize(eng.h))
        b = eng.read_results()
    except Exception:
        return []
    over = (b.reason != 0) | (b.n_plies >= eng.max_moves)
    return [(b.game_data(g), int(b.winner[g]), b.end_reason(g)) for g in range(b.n_games) if over[g] and b.error[g] == 0]


def self_play_game(network, temperature=1.0, render=False, num_simulations=None, opponent_network=None):
    """self_play.py:178-312.  Returns ([(board, {move: prob}, z)], winner, end_reason).

    The reference draws one double per ply from NumPy's GLOBAL stream (np.random.choice,
    self_play.py:242).  The mirror hands the engine the next 70 doubles of that stream and then
    rewinds it to just after the plies actually played, so a caller that seeds np.random sees
    the same moves and the same stream position as with the reference."""
    state = np.random.get_state()
    uniforms = np.random.random_sample(_lib.MAX_PLIES).reshape(1, _lib.MAX_PLIES)
    batch = _play_batch(network, 1, temperature, num_simulations, opponent_network,
                        seeds=np.zeros(1, np.uint32), uniforms=uniforms)
    np.random.set_state(state)
    if batch.error[0]:
        # S <= 8: all root visits are zero -> NaN probabilities
        raise ValueError("probabilities contain NaN")
    if batch.n_plies[0]:
        np.random.random_sample(int(batch.n_plies[0]))
    if render:
        env = ChineseChess()
        for i in range(int(batch.n_plies[0])):
            mv = decode_move(batch.chosen[0, i])
            env.make_move(mv)
            env.render()
            print(f"走法: {mv}, 即时奖励: {batch.step_reward[0, i]:.2f}")
    return batch.game_data(0), int(batch.winner[0]), batch.end_reason(0)


def parallel_self_play(network, num_games, temperature=1.0, num_simulations=None, num_workers=4,
                       opponent_network=None, seeds=None, inference_dtype=None):
    """self_play.py:368-469.  The reference's process pool becomes G concurrent games on the GPU;
    `num_workers` is accepted and ignored.  Results come back in game order (the reference's
    order is arbitrary: imap_unordered).  Ctrl-C raises InterruptedWithResults(results) with the
    games that had already finished (self_play.py:436-452); `inference_dtype` ("bf16" | "f32", default
    INFERENCE_DTYPE) picks the precision a CUDA ChessNet is evaluated in."""
    batch = _play_batch(network, num_games, temperature, num_simulations, opponent_network, seeds=seeds,
                        inference_dtype=inference_dtype, partial_on_interrupt=True)
    return batch.results()


class SelfPlay:
    """BASELINE north_star's name for this surface (`SelfPlay.play_game()`): an object holding the arguments of
    self_play_game (self_play.py:178).  play_game() is self_play_game(...), play_games(n) is parallel_self_play(...)."""

    def __init__(self, network, temperature=1.0, num_simulations=None, opponent_network=None):
        self.network, self.temperature = network, temperature
        self.num_simulations, self.opponent_network = num_simulations, opponent_network

    def play_game(self, render=False):
        return self_play_game(self.network, temperature=self.temperature, render=render,
                              num_simulations=self.num_simulations, opponent_network=self.opponent_network)

    def play_games(self, num_games, num_workers=4, seeds=None, inference_dtype=None):
        return parallel_self_play(self.network, num_games, temperature=self.temperature,
                                  num_simulations=self.num_simulations, num_workers=num_workers,
                                  opponent_network=self.opponent_network, seeds=seeds, inference_dtype=inference_dtype)
