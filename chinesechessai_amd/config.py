"""The handful of reference constants the self-play hot path reads (config.py:9-10,48,62-74 of the
reference).  Trainer policy, display and path settings are out of scope."""

# search / game limits
MCTS_SIMULATIONS = 50            # simulations per move (reference config.py:10)
MAX_MOVES = 70                   # plies per game (config.py:9); make_move hard-codes the same cap (chess_env.py:400)
LEAF_BATCH = 8                   # leaves evaluated together (self_play.py:101)
C_PUCT = 1.5                     # exploration constant (self_play.py:40)
SELF_PLAY_GAMES = 100            # games per collection round (config.py:8)
NUM_WORKERS = 4                  # reference pool size (config.py:48); accepted and ignored here
USE_MULTIPROCESSING = True

# board geometry and piece codes: +code red, -code black (config.py:62-74)
BOARD_SIZE, BOARD_WIDTH = 10, 9
_KINDS = ("KING", "ADVISOR", "BISHOP", "KNIGHT", "ROOK", "CANNON", "PAWN")
PIECES = {"EMPTY": 0}
for _code, _kind in enumerate(_KINDS, start=1):
    PIECES["R_" + _kind] = _code
    PIECES["B_" + _kind] = -_code

_SIM_SCHEDULE = ((1000, 30), (3000, 35), (8000, 60), (15000, 100))


def get_dynamic_mcts_simulations(total_games):
    """Simulation count by games trained so far (config.py:13-28): 30 / 35 / 60 / 100 / 150."""
    for limit, sims in _SIM_SCHEDULE:
        if total_games < limit:
            return sims
    return 150
