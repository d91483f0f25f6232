"""On-disk formats of the reference that touch the self-play path, so that
the reference's tools keep reading what this engine writes and vice versa:

  * checkpoint dict of Trainer.save_model / load_model (trainer.py:434-458):
      {'model_state_dict', 'optimizer_state_dict', 'total_games', 'training_steps'} via torch.save
  * data/best_games.pkl of Trainer._save_best_games (trainer.py:468-502), read by
    view_best_games.py:193-213: a pickled list (last 500) of
      {'timestamp', 'total_games', 'game_data', 'winner', 'moves', 'type'}
    where game_data is self_play_game's [(board, {move: prob}, z)] list; the replay tool re-derives
    each move as argmax of the sample's probabilities.
"""
import os
import pickle
from datetime import datetime


def save_checkpoint(path, network, optimizer=None, total_games=0, training_steps=0):
    """trainer.py:434-444"""
    import torch
    os.makedirs(os.path.dirname(os.path.abspath(path)), exist_ok=True)
    torch.save({
        'model_state_dict': network.state_dict(),
        'optimizer_state_dict': optimizer.state_dict() if optimizer is not None else {},
        'total_games': total_games,
        'training_steps': training_steps,
    }, path)


def load_checkpoint(path, network=None, map_location="cpu"):
    """trainer.py:452-458.  Returns (network, meta).  A ChessNet is created with as many residual
    blocks as the state_dict holds (the reference hard-codes 4)."""
    import torch
    from .neural_network import ChessNet
    ckpt = torch.load(path, map_location=map_location)
    sd = ckpt['model_state_dict']
    if network is None:
        blocks = 1 + max(int(k.split('.')[1]) for k in sd if k.startswith('res_blocks.'))
        network = ChessNet(num_channels=sd['conv1.weight'].shape[0], num_blocks=blocks)
    network.load_state_dict(sd)
    meta = {'total_games': ckpt.get('total_games', 0), 'training_steps': ckpt.get('training_steps', 0),
            'optimizer_state_dict': ckpt.get('optimizer_state_dict')}
    return network, meta


def append_best_games(path, best_games, total_games, keep=500):
    """trainer.py:468-502.  best_games: [(game_data, winner, moves, game_type)]."""
    if not best_games:
        return 0
    os.makedirs(os.path.dirname(os.path.abspath(path)), exist_ok=True)
    existing = []
    if os.path.exists(path):
        try:
            with open(path, 'rb') as f:
                existing = pickle.load(f)
        except Exception:
            existing = []
    for game_data, winner, moves, game_type in best_games:
        existing.append({'timestamp': datetime.now(), 'total_games': total_games, 'game_data': game_data,
                         'winner': winner, 'moves': moves, 'type': game_type})
    existing = existing[-keep:]
    with open(path, 'wb') as f:
        pickle.dump(existing, f)
    return len(existing)


def best_games_from_results(results, game_type="训练"):
    """(game_data, winner, end_reason) triples of parallel_self_play -> the tuples
    _save_best_games stores (moves = len(game_data))."""
    return [(gd, w, len(gd), game_type) for gd, w, _ in results]
