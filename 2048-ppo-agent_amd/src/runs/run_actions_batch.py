"""Free-function rollout (reference src/runs/run_actions_batch.py:10-57)."""
from typing import Callable

from .batch_runner import BatchRunner, State, states_from_trajectory


def run_actions_batch(init_seed: int, batch_size: int, act_fn: Callable, rng_mode=None) -> list:
    """Run ``batch_size`` boards to termination; returns the list of states AFTER each step
    (the initial state is not included, as in the reference)."""
    runner = BatchRunner(init_seed=init_seed, act_fn=act_fn, rng_mode=rng_mode)
    tr = runner.collect(batch_size, fill_frozen=True)
    return states_from_trajectory(tr, include_init=False)
