"""chinesechessai_amd — MI355X-native batched self-play engine for Chinese Chess.

One data-parallel hot path of hpy666666/ChineseChessAI (rules -> MCTS -> leaf evaluation ->
(state, pi, z) samples), rebuilt as hand-written HIP kernels for gfx950 behind the reference's
own Python surface.  See DESIGN.md / INTEGRATION.md.
"""
from . import _lib, config  # noqa: F401
from ._lib import XqError  # noqa: F401
from .chess_env import ChineseChess  # noqa: F401

ChessEnv = ChineseChess          # BASELINE north_star's name for chess_env.py:9


def __getattr__(name):
    # torch-dependent parts load lazily so that rules-only users do not pay the torch import
    if name in ("MCTS", "MCTSNode", "self_play_game", "parallel_self_play", "InterruptedWithResults", "SelfPlay"):
        from . import self_play
        return getattr(self_play, name)
    if name in ("SelfPlayEngine", "HashNetEvaluator", "TorchNetEvaluator", "CallbackEvaluator"):
        from . import engine
        return getattr(engine, name)
    if name in ("ChessNet", "InferenceNet"):
        from . import neural_network
        return getattr(neural_network, name)
    raise AttributeError(name)
