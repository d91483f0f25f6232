"""Arena / evaluation callers of the self-play path: the game loops of
evaluate.py:63-86 (`evaluate_model`: N self-play games at temperature 0.1) and
compare_models.py:13-92 (`play_match`: network1 as red vs network2 as black, temperature 0.3),
batched on the GPU engine instead of one Python game at a time.  Return dicts keep the reference's
keys; loading checkpoints, printing and the heuristic "skill level" strings stay with the caller.
"""
import numpy as np

from . import _lib
from .config import MAX_MOVES, MCTS_SIMULATIONS
from .engine import SelfPlayEngine
from .self_play import _evaluator_for


def _play(network, opponent, num_games, temperature, num_simulations, seeds):
    sims = num_simulations if num_simulations else MCTS_SIMULATIONS
    ev = _evaluator_for(network)
    ev_b = _evaluator_for(opponent) if opponent is not None else None
    eng = SelfPlayEngine(num_games, sims=sims, temperature=temperature, max_moves=MAX_MOVES,
                         opponent_mode=opponent is not None,
                         planes_format=getattr(ev, "planes_format", _lib.PLANES_NONE))
    if seeds is None:
        seeds = np.random.randint(0, 2 ** 31 - 1, size=num_games).astype(np.uint32)
    try:
        eng.play(ev, np.asarray(seeds, dtype=np.uint32), opponent_evaluator=ev_b, read=False)
        out = eng.read_game_outcomes()
        batch = eng.read_results()
    finally:
        eng.close()
    return out, batch


def evaluate_games(network, num_games=10, temperature=0.1, num_simulations=None, seeds=None):
    """The statistics block of evaluate.py:56-100 for `num_games` self-play games."""
    out, batch = _play(network, None, num_games, temperature, num_simulations, seeds)
    ok = out["error"] == 0
    winners = out["winner"][ok]
    moves = out["n_samples"][ok]                      # len(game_data) (evaluate.py:70)
    n = int(ok.sum())
    return {
        "red_wins": int((winners == 1).sum()), "black_wins": int((winners == -1).sum()),
        "draws": int((winners == 0).sum()), "num_games": n,
        "avg_moves": float(moves.mean()) if n else 0.0,
        "min_moves": int(moves.min()) if n else 0, "max_moves": int(moves.max()) if n else 0,
        "red_rate": float((winners == 1).mean() * 100) if n else 0.0,
        "black_rate": float((winners == -1).mean() * 100) if n else 0.0,
        "draw_rate": float((winners == 0).mean() * 100) if n else 0.0,
        "end_reasons": [batch.end_reason(g) for g in range(num_games) if ok[g]],
    }


def play_match(network1, network2, num_games=20, verbose=True, num_simulations=None, seeds=None, temperature=0.3):
    """compare_models.py:13-92: network1 plays red, network2 black, moves sampled at temperature
    0.3 from the root visits; same result keys."""
    out, _ = _play(network1, network2, num_games, temperature, num_simulations, seeds)
    ok = out["error"] == 0
    winners = out["winner"][ok]
    n = max(int(ok.sum()), 1)
    res = {
        "model1_wins": int((winners == 1).sum()), "model2_wins": int((winners == -1).sum()),
        "draws": int((winners == 0).sum()), "avg_moves": float(out["n_plies"][ok].sum() / n),   # env.move_count
        "model1_winrate": float((winners == 1).sum() / n * 100),
        "model2_winrate": float((winners == -1).sum() / n * 100),
        "draw_rate": float((winners == 0).sum() / n * 100),
    }
    if verbose:
        for g in range(num_games):
            r = {1: "模型1胜", -1: "模型2胜"}.get(int(out["winner"][g]), "和局")
            print(f"  对局 {g + 1}/{num_games}... {r} ({int(out['n_plies'][g])}步)")
    return res
