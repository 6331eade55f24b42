"""Max-tile evaluation protocol of the reference's visualisation scripts, without the plotting.

run/viz_naive_strategies.py:123-200 and run/viz_ppo_agent.py:238-300 both do: ``num_episodes`` episodes in batches
of 100, batch *i* from ``BatchRunner(init_seed=seed + 100 * i)``, metric = the largest tile on the FINAL board of
each episode; the PPO agent is evaluated greedily with the legal-action mask
(``TorchActionFunction(agent, use_mask=True, sample_actions=False)``).  The README histograms are this protocol
with seed 42 (random policy: mean max tile 109.17, DRUL: 189.44, the reference's trained PPO agent: about 383).
Only the 16-byte final boards leave the device.
"""
from __future__ import annotations

from collections import Counter
from typing import Callable, Dict

import numpy as np
import torch

from .batch_runner import BatchRunner


def evaluate_max_tile(act_fn: Callable, num_episodes: int = 1000, seed: int = 42, batch_size: int = 100,
                      rng_mode=None, device=None) -> Dict:
    """-> {"mean_max_tile", "percent": {tile: % of episodes}, "counts": {tile: n}, "mean_episode_length", "episodes"}"""
    batch_size = min(batch_size, num_episodes)
    tiles, lengths = [], []
    done, i = 0, 0
    while done < num_episodes:
        b = min(batch_size, num_episodes - done)
        runner = BatchRunner(init_seed=seed + i * batch_size, act_fn=act_fn, rng_mode=rng_mode, device=device)
        tr = runner.collect(b)
        tiles.append((1 << tr.final_boards.max(dim=1).values.cpu().numpy().astype(np.int64)))  # exact powers of two
        lengths.append(tr.ep_len.cpu().numpy())
        done += b
        i += 1
    tiles = np.concatenate(tiles).astype(np.int64)
    counts = Counter(tiles.tolist())
    return {
        "episodes": int(len(tiles)), "mean_max_tile": float(tiles.mean()),
        "counts": {int(k): int(v) for k, v in sorted(counts.items())},
        "percent": {int(k): round(100.0 * v / len(tiles), 1) for k, v in sorted(counts.items())},
        "mean_episode_length": float(np.concatenate(lengths).mean()),
    }


def evaluate_agent(agent, device, num_episodes: int = 1000, seed: int = 42, rng_mode=None) -> Dict:
    """Greedy, masked evaluation of a PPO agent (run/viz_ppo_agent.py:267-300)."""
    from ..ppo.torch_action_wrapper import TorchActionFunction

    was_training = agent.training
    fn = TorchActionFunction(agent, use_mask=True, sample_actions=False, device=device)
    try:
        return evaluate_max_tile(fn, num_episodes, seed, rng_mode=rng_mode, device=device)
    finally:
        agent.train(was_training)
