"""On-disk formats: checkpoint dict and best_games.pkl round trips."""
import pickle

import numpy as np
import torch

from chinesechessai_amd import formats
from chinesechessai_amd.neural_network import ChessNet


def test_checkpoint_roundtrip_and_keys(tmp_path):
    torch.manual_seed(0)
    net = ChessNet(num_channels=16, num_blocks=3)
    opt = torch.optim.Adam(net.parameters(), lr=1e-3)
    p = tmp_path / "models" / "latest.pt"
    formats.save_checkpoint(str(p), net, opt, total_games=1234, training_steps=56)
    raw = torch.load(str(p), map_location="cpu")
    assert set(raw) == {"model_state_dict", "optimizer_state_dict", "total_games", "training_steps"}   # trainer.py:438-443
    net2, meta = formats.load_checkpoint(str(p))
    assert meta["total_games"] == 1234 and meta["training_steps"] == 56
    assert net2.num_blocks == 3 and net2.num_channels == 16
    for k, v in net.state_dict().items():
        assert torch.equal(v, net2.state_dict()[k]), k


def test_best_games_pickle_structure(tmp_path):
    board = np.zeros((10, 9), np.int8)
    gd = [(board, {(6, 0, 5, 0): np.float64(0.25), (6, 2, 5, 2): np.float64(0.75)}, -0.15)]
    results = [(gd, 0, "超过70步判和"), (gd * 3, 1, "将死黑方")]
    best = formats.best_games_from_results(results)
    assert best[1][1] == 1 and best[1][2] == 3 and best[0][3] == "训练"
    p = tmp_path / "data" / "best_games.pkl"
    n = formats.append_best_games(str(p), best, total_games=200)
    assert n == 2
    for _ in range(300):
        formats.append_best_games(str(p), best, total_games=201)
    games = pickle.load(open(p, "rb"))
    assert len(games) == 500                                       # trainer.py:491
    g = games[-1]
    assert set(g) == {"timestamp", "total_games", "game_data", "winner", "moves", "type"}   # trainer.py:481-488
    # view_best_games.py:205-208 re-derives the move as the argmax of the sample's probabilities
    b, probs, z = g["game_data"][0]
    assert max(probs.items(), key=lambda x: x[1])[0] == (6, 2, 5, 2)
