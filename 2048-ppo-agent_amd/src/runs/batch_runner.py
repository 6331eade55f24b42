"""BatchRunner: the reference's rollout driver (src/runs/batch_runner.py:10-195) on the MI355X engine.

Same constructor, ``act_fn`` property, ``run_actions_batch`` 7-tuple and ``run_rollout_batch`` state list.
The env, the RNG and -- for the three known plug-ins (act_drul, act_randomly, TorchActionFunction) -- the
policy sampling run in HIP kernels (src/g2048); an arbitrary Python ``act_fn`` is called per env on the
host between device steps, exactly as user code, with JAX-free key words instead of jax key objects.
"""
from __future__ import annotations

import os
from dataclasses import dataclass
from typing import Callable, Optional

import numpy as np
import torch

from ..g2048 import native as nv
from ..g2048.engine import RolloutEngine, Trajectory, mask_bits_to_bool, one_hot_observations

ENV_ID = "2048"


def _resolve_rng_mode(rng_mode) -> int:
    if rng_mode is None:
        rng_mode = os.environ.get("G2048_RNG_MODE", "partitionable")
    if isinstance(rng_mode, str):
        table = {"legacy": nv.RNG_LEGACY, "partitionable": nv.RNG_PARTITIONABLE, "0": 0, "1": 1}
        if rng_mode.lower() not in table:
            raise ValueError(f"unknown rng_mode {rng_mode!r}")
        return table[rng_mode.lower()]
    return int(rng_mode)


@dataclass
class State:
    """Host snapshot of one lock-step, with the field names of ``pgx.State`` that the reference reads."""

    observation: np.ndarray  # bool [B, 4, 4, 31]
    legal_action_mask: np.ndarray  # bool [B, 4]
    rewards: np.ndarray  # f32 [B, 1]
    terminated: np.ndarray  # bool [B]
    truncated: np.ndarray  # bool [B] (always False for 2048)
    board: np.ndarray  # u8 [B, 16] log2 tiles


class BatchRunner:
    """Run B 2048 boards in lock-step until all have terminated.

    Parameters
    ----------
    init_seed : int
        Seed of the JAX-compatible threefry key chain; the chain persists across calls.
    act_fn : Callable, optional
        ``(rng_key, obs, mask) -> (action, log_prob, value)`` plug-in (un-batched protocol).
    rng_mode : {"partitionable", "legacy"}, optional
        Which ``jax.random`` stream to reproduce (default: partitionable, the default of the reference's
        pinned jax 0.5.3; env var G2048_RNG_MODE overrides).
    env0, total_envs : shard description for multi-GPU runs: this runner owns envs [env0, env0 + B) of a
        global batch of ``total_envs`` and reproduces exactly that slice of the single-device run.
    """

    def __init__(self, init_seed: int, act_fn: Callable = None, rng_mode=None, device=None, env0: int = 0,
                 total_envs: Optional[int] = None):
        self.rng_mode = _resolve_rng_mode(rng_mode)
        self._engine = RolloutEngine(init_seed, self.rng_mode, device)
        self.device = self._engine.device
        self.env0 = int(env0)
        self.total_envs = total_envs
        self._act_fn = act_fn

    # the reference stores jit(vmap(act_fn)); here the callable itself is kept and dispatched on
    @property
    def act_fn(self):
        return self._act_fn

    @act_fn.setter
    def act_fn(self, act_fn: Callable):
        self._act_fn = act_fn

    @property
    def key(self) -> np.ndarray:
        """Current head of the key chain (two threefry words), as ``self.key`` in the reference."""
        return self._engine.key

    # ------------------------------------------------------------------ device path
    def collect(self, batch_size: int, fill_frozen: bool = False) -> Trajectory:
        """B complete episodes, left in HBM (what PPOTrainer consumes)."""
        if self._act_fn is None:
            raise ValueError("The action function is not set.")
        if batch_size is None or int(batch_size) <= 0:
            raise ValueError("batch_size must be a positive integer")
        B = int(batch_size)
        kw = dict(B_total=self.total_envs, env0=self.env0, fill_frozen=fill_frozen)
        fn = self._act_fn
        fused = getattr(fn, "fused_policy", None)
        if fused is not None:
            return self._engine.rollout_fused(B, fused, **kw)
        if hasattr(fn, "policy_fn"):  # TorchActionFunction
            return self._engine.rollout_policy(B, fn.policy_fn, use_mask=fn.use_mask, sample=fn.sample_actions,
                                               sync_every=getattr(fn, "sync_every", 8), compact=getattr(fn, "compact", True),
                                               **kw)
        return self._collect_host_callable(B, fn, fill_frozen)

    def collect_fixed(self, batch_size: int, horizon: int, restart: bool = False):
        """Throughput mode (the reference has only the lock-step loop of src/runs/batch_runner.py:117):
        ``horizon`` lock-steps of ``batch_size`` always-live lanes with per-lane auto-reset, policy in the loop.
        -> (FixedTrajectory, last_values f32 [B]) where last_values = V(state after the last step), the bootstrap of the
        GAE scan.  Env state persists across calls.  Needs a TorchActionFunction-style ``act_fn`` (``policy_fn``)."""
        if self._act_fn is None:
            raise ValueError("The action function is not set.")
        fn = self._act_fn
        if not hasattr(fn, "policy_fn"):
            raise ValueError("the fixed-horizon mode needs a policy act_fn (TorchActionFunction)")
        if batch_size is None or int(batch_size) <= 0 or int(horizon) <= 0:
            raise ValueError("batch_size and horizon must be positive integers")
        traj = self._engine.rollout_policy_fixed(int(batch_size), int(horizon), fn.policy_fn, use_mask=fn.use_mask,
                                                 sample=fn.sample_actions, B_total=self.total_envs, env0=self.env0,
                                                 restart=restart)
        _, last_values = fn.policy_fn(traj.final_boards, traj.final_masks)
        return traj, last_values.to(torch.float32).reshape(-1)

    def _collect_host_callable(self, B: int, fn: Callable, fill_frozen: bool) -> Trajectory:
        """Arbitrary Python act_fn: env + key splits on the device, the callable per env on the host."""
        eng, mode, dev = self._engine, self.rng_mode, self.device
        B_total = B if self.total_envs is None else self.total_envs
        key0 = eng.key.copy()
        key, sub = nv.chain_keys(key0, 1, mode)
        boards, masks, done, ep_len = eng._alloc_state(B)
        nv.reset_fused(sub[0], boards, masks, done, ep_len, B_total, self.env0, mode)
        init_boards = boards.clone()
        rewards = torch.empty(B, dtype=torch.float32, device=dev)
        frames = []
        while True:
            key, subs = nv.chain_keys(key, 2, mode)
            act_keys = nv.keys_to_numpy(nv.split(subs[0], B_total, mode, dev)[self.env0:self.env0 + B])
            obs = one_hot_observations(boards).cpu().numpy()
            mk = mask_bits_to_bool(masks).cpu().numpy()
            a = np.empty(B, np.int32)
            lp = np.zeros(B, np.float32)
            v = np.zeros(B, np.float32)
            has_lp = has_v = True
            for e in range(B):
                out = fn(act_keys[e], obs[e], mk[e])
                a[e] = int(np.asarray(out[0]))
                if out[1] is None:
                    has_lp = False
                else:
                    lp[e] = float(np.asarray(out[1]))
                if out[2] is None:
                    has_v = False
                else:
                    v[e] = float(np.asarray(out[2]))
            if (a < 0).any() or (a > 3).any():
                raise ValueError("act_fn returned an action outside [0, 3]")
            step_keys = nv.split(subs[1], B_total, mode, dev)[self.env0:self.env0 + B].contiguous()
            pre_b, pre_m, pre_d = boards.clone(), masks.clone(), done.clone()
            nv.step(boards, masks, done, torch.from_numpy(a).to(dev), step_keys, rewards, mode)
            ep_len += (pre_d == 0).to(torch.int32)
            frames.append((pre_b, (torch.from_numpy(a).to(dev).to(torch.uint8) | (pre_m << 2) | (done << 6)),
                           rewards.clone(), torch.from_numpy(lp).to(dev) if has_lp else None,
                           torch.from_numpy(v).to(dev) if has_v else None))
            if bool((done != 0).all().item()):
                break
        T = len(frames)
        eng.key, _ = nv.chain_keys(key0, 1 + 2 * T, mode)
        stack = lambda i: None if frames[0][i] is None else torch.stack([f[i] for f in frames])
        return Trajectory(boards=stack(0), meta=stack(1), rewards=stack(2), log_probs=stack(3), values=stack(4),
                          ep_len=ep_len, final_boards=boards, final_masks=masks, init_boards=init_boards, T=T, B=B,
                          frozen_filled=True)

    # ------------------------------------------------------------------ reference API (host arrays)
    def run_actions_batch(self, batch_size: int):
        """B complete episodes -> the reference's 7 numpy arrays, in its order and layout:

        observations bool [B,T,4,4,31] (board BEFORE step t), actions i32 [B,T], action_masks bool [B,T,4],
        log_probs f32 [B,T], values f32 [B,T], rewards f32 [B,T], terminations bool [B,T] (after step t).
        ``log_probs``/``values`` are ``None`` for plug-ins that return ``None`` there (act_drul: both,
        act_randomly: values) -- the reference raises at ``np.stack`` in that case.
        """
        tr = self.collect(batch_size, fill_frozen=True)
        T, B = tr.T, tr.B
        to_bt = lambda x: None if x is None else np.ascontiguousarray(np.swapaxes(x.cpu().numpy(), 0, 1))
        observations = to_bt(one_hot_observations(tr.boards))
        actions = to_bt(tr.actions.to(torch.int32))
        action_masks = to_bt(mask_bits_to_bool(tr.masks))
        log_probs = to_bt(tr.log_probs)
        values = to_bt(tr.values)
        rewards = to_bt(tr.rewards)
        terminations = to_bt(tr.terms.to(torch.bool))
        return observations, actions, action_masks, log_probs, values, rewards, terminations

    def run_rollout_batch(self, batch_size: int) -> list:
        """B complete episodes -> list of T+1 ``State`` snapshots, the initial state first."""
        tr = self.collect(batch_size, fill_frozen=True)
        return states_from_trajectory(tr, include_init=True)


def states_from_trajectory(tr: Trajectory, include_init: bool) -> list:
    T, B = tr.T, tr.B
    boards = torch.cat([tr.boards, tr.final_boards[None]], dim=0)  # [T+1, B, 16]
    masks = torch.cat([tr.masks, tr.final_masks[None]], dim=0)
    obs = one_hot_observations(boards).cpu().numpy()
    mk = mask_bits_to_bool(masks).cpu().numpy()
    bd = boards.cpu().numpy()
    rew = tr.rewards.cpu().numpy()
    term = tr.terms.cpu().numpy().astype(bool)
    states = []
    for k in range(0 if include_init else 1, T + 1):
        states.append(State(
            observation=obs[k], legal_action_mask=mk[k],
            rewards=(rew[k - 1] if k > 0 else np.zeros(B, np.float32)).reshape(B, 1),
            terminated=term[k - 1] if k > 0 else np.zeros(B, bool), truncated=np.zeros(B, bool), board=bd[k]))
    return states
