"""Shape constants of the 2048 environment (same names/values as the reference src/env_definitions.py:1-8)."""
OBS_DIM = 31  # one-hot classes per cell: log2(tile) in 0..30
BOARD_DIM = (4, 4)
BOARD_FLAT_DIM = 16
ACTION_DIM = 4  # 0 left, 1 up, 2 right, 3 down
