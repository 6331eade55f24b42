"""Max-tile statistics over many episodes (reference src/runs/run_actions_max_tile.py:10-71)."""
import warnings
from typing import Callable

import numpy as np
import torch

from ..stats.running_stats_vec import RunningStatsVec
from .batch_runner import BatchRunner


def run_actions_max_tile(init_seed: int, batch_size: int, num_envs: int, act_fn: Callable,
                         rng_mode=None) -> RunningStatsVec:
    """``num_envs // batch_size`` batches of complete episodes; statistics of ``2 ** max(log2 board)``.

    As in the reference the board examined is the observation of the LAST lock-step (the board before
    the final step of the slowest env), and ``num_envs`` is rounded down to a multiple of ``batch_size``.
    Only the 16-byte boards leave the device, not the one-hot observations.
    """
    if num_envs % batch_size != 0:
        warnings.warn(f"The number of environments ({num_envs}) is not divisible by the batch size "
                      f"({batch_size}); it is rounded down to a multiple of it.")
    runner = BatchRunner(init_seed=init_seed, act_fn=act_fn, rng_mode=rng_mode)
    stats = RunningStatsVec()
    for _ in range(num_envs // batch_size):
        tr = runner.collect(batch_size, fill_frozen=True)
        last_obs_boards = tr.boards[tr.T - 1]  # observation recorded at the last lock-step
        max_tiles = np.left_shift(1, last_obs_boards.max(dim=1).values.cpu().numpy().astype(np.int64))  # exact
        stats.push(max_tiles.astype(np.float64).reshape(1, -1))
    return stats
