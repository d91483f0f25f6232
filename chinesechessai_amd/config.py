"""Constants of the reference's config.py that the self-play hot path reads
(config.py:9-10,48,62-74).  Trainer policy, display and path settings are out of scope."""
SELF_PLAY_GAMES = 100
MAX_MOVES = 70                  # config.py:9 (the rules engine hard-codes the same 70, chess_env.py:400)
MCTS_SIMULATIONS = 50           # config.py:10
NUM_WORKERS = 4                 # config.py:48 (accepted and ignored: one process per GPU, G games in lock-step)
USE_MULTIPROCESSING = True

BOARD_SIZE = 10                 # config.py:62-63
BOARD_WIDTH = 9

PIECES = {                      # config.py:66-74
    'EMPTY': 0,
    'R_KING': 1, 'R_ADVISOR': 2, 'R_BISHOP': 3, 'R_KNIGHT': 4, 'R_ROOK': 5, 'R_CANNON': 6, 'R_PAWN': 7,
    'B_KING': -1, 'B_ADVISOR': -2, 'B_BISHOP': -3, 'B_KNIGHT': -4, 'B_ROOK': -5, 'B_CANNON': -6, 'B_PAWN': -7,
}

LEAF_BATCH = 8                  # self_play.py:101
C_PUCT = 1.5                    # self_play.py:40


def get_dynamic_mcts_simulations(total_games):
    """config.py:13-28"""
    if total_games < 1000:
        return 30
    elif total_games < 3000:
        return 35
    elif total_games < 8000:
        return 60
    elif total_games < 15000:
        return 100
    return 150
