"""Device-resident rollout engine: the GPU replacement for BatchRunner's lock-step loop.

State (boards/masks/done/ep_len) and the whole trajectory stay in HBM; the host only advances the
JAX-compatible key chain (two threefry blocks per split) and polls a 4-byte live-env counter.
Reference being replaced: src/runs/batch_runner.py:67-154 (loop), :105-128 (key schedule).
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Callable, Optional

import numpy as np
import torch

from . import native as nv


def seed_key(seed: int) -> np.ndarray:
    """jax.random.key(seed) -> threefry key words (hi, lo)."""
    seed = int(seed)
    return np.array([(seed >> 32) & 0xFFFFFFFF, seed & 0xFFFFFFFF], dtype=np.uint32)


@dataclass
class Trajectory:
    """Step-major SoA trajectory in HBM.  x[t, e]; frames with t >= ep_len[e] are valid only if
    ``frozen_filled`` (they then hold what the reference's [B, T] arrays hold for finished envs)."""

    boards: torch.Tensor  # u8 [T, B, 16]  board before step t
    meta: torch.Tensor  # u8 [T, B]      action | mask_before << 2 | done_after << 6
    rewards: torch.Tensor  # f32 [T, B]
    log_probs: Optional[torch.Tensor]  # f32 [T, B] or None
    values: Optional[torch.Tensor]  # f32 [T, B] or None
    ep_len: torch.Tensor  # i32 [B] steps through first termination
    final_boards: torch.Tensor  # u8 [B, 16]
    final_masks: torch.Tensor  # u8 [B]  legal mask after the last step
    init_boards: torch.Tensor  # u8 [B, 16]
    T: int
    B: int
    frozen_filled: bool

    @property
    def actions(self) -> torch.Tensor:
        return self.meta & 3

    @property
    def masks(self) -> torch.Tensor:
        return (self.meta >> 2) & 15

    @property
    def terms(self) -> torch.Tensor:
        return (self.meta >> 6) & 1

    def valid(self) -> torch.Tensor:
        """bool [T, B]: steps the reference's RolloutBuffer keeps (0..first termination inclusive)."""
        t = torch.arange(self.T, device=self.ep_len.device, dtype=torch.int32)
        return t[:, None] < self.ep_len[None, :]

    def num_steps(self) -> int:
        return int(self.ep_len.sum().item())


@dataclass
class FixedTrajectory:
    """Fixed-horizon trajectory (throughput mode): every lane has exactly T rows; a lane whose step terminated started
    its next episode in the following row, ``terms`` marks the boundaries."""

    boards: torch.Tensor  # u8 [T, B, 16]  board before step t
    meta: torch.Tensor  # u8 [T, B]      action | mask_before << 2 | done_after << 6
    rewards: torch.Tensor  # f32 [T, B]
    log_probs: torch.Tensor  # f32 [T, B]
    values: torch.Tensor  # f32 [T, B]
    final_boards: torch.Tensor  # u8 [B, 16]  state after step T-1 (a fresh board where that step was terminal)
    final_masks: torch.Tensor  # u8 [B]
    ep_len: torch.Tensor  # i32 [B] steps of the episode that is still running after step T-1
    ep_len_before: torch.Tensor  # i32 [B] the same counter before step 0 (episodes continue across rollouts)
    T: int
    B: int

    actions = Trajectory.actions
    masks = Trajectory.masks
    terms = Trajectory.terms

    def num_steps(self) -> int:
        return self.T * self.B

    def finished_episode_lengths(self) -> torch.Tensor:
        """i32 [n]: lengths of the episodes that ended inside this rollout (counting their steps of earlier rollouts)."""
        done = self.terms.bool()  # [T, B]
        t = torch.arange(self.T, device=done.device, dtype=torch.int32)[:, None].expand_as(done)
        last = torch.cummax(torch.where(done, t, torch.full_like(t, -1)), dim=0).values  # last terminal row <= t
        prev = torch.cat([torch.full_like(last[:1], -1), last[:-1]], dim=0)  # last terminal row < t
        length = torch.where(prev >= 0, t - prev, t + 1 + self.ep_len_before[None, :])
        return length[done]

    def finished_episode_max_rewards(self, carry: Optional[torch.Tensor] = None):
        """-> (f32 [n], f32 [B]): the largest single-step reward of every episode that ended inside this rollout, in the order of
        ``finished_episode_lengths`` (the reference's per-episode "episode reward" statistic, src/ppo/ppo_trainer.py:204-215), and
        the running maximum of each lane's unfinished episode to hand to the next rollout.  ``carry``: that running maximum from
        the previous rollout of the same lanes (None: episodes start here)."""
        done = self.terms.bool()  # [T, B]
        T, B = done.shape
        seg = torch.cumsum(done.to(torch.int64), dim=0) - done.to(torch.int64)  # terminations before row t = its episode's index
        lane = torch.arange(B, device=done.device, dtype=torch.int64)[None, :]
        key = (seg * B + lane).reshape(-1)
        best = torch.full(((T + 1) * B,), float("-inf"), dtype=torch.float32, device=done.device)
        best = best.scatter_reduce(0, key, self.rewards.reshape(-1).float(), reduce="amax").view(T + 1, B)
        if carry is not None:
            best[0] = torch.maximum(best[0], carry.to(best.dtype))
        vals = best.reshape(-1)[key].view(T, B)[done]
        n_done = done.sum(dim=0)  # the unfinished episode of a lane is its segment number n_done (-inf when nothing of it ran yet)
        return vals, best.gather(0, n_done[None, :]).squeeze(0)


class RolloutEngine:
    """One shard of a lock-step batch: envs [env0, env0 + B) of B_total, on one GPU."""

    def __init__(self, seed: int, rng_mode: int = nv.RNG_PARTITIONABLE, device=None):
        nv.load()
        if device is None:
            device = torch.device("cuda", torch.cuda.current_device())
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise nv.NativeError("RolloutEngine needs a HIP device; there is no CPU path")
        self.rng_mode = int(rng_mode)
        self.key = seed_key(seed)
        self.initial_capacity = 256
        # trajectory workspace, reused by successive rollouts of the same batch size: a Trajectory is a view
        # into it and stays valid until the next rollout call on this engine (hipMalloc of GB-sized buffers
        # costs milliseconds; the rollout itself is ~1 ms at 65 536 boards)
        self._ws = None
        self._ws_key = None
        self._fixed = None  # persistent env state of the fixed-horizon mode: (boards, masks, ep_len, B, B_total, env0)
        # multi-GPU: a callable int -> int returning the maximum over all ranks.  The reference's key chain
        # advances by 1 + 2 T splits per rollout with T the longest episode of the WHOLE batch, so shards agree
        # on it before advancing their (identical) host-side chains.
        self.global_max = None

    # ------------------------------------------------------------------ buffers
    def _alloc_state(self, B):
        d = self.device
        return (torch.empty((B, 16), dtype=torch.uint8, device=d), torch.empty(B, dtype=torch.uint8, device=d),
                torch.empty(B, dtype=torch.uint8, device=d), torch.empty(B, dtype=torch.int32, device=d))

    def _alloc_traj(self, cap, B, with_logp, with_values):
        d = self.device
        ws_key = (B, bool(with_logp), bool(with_values))
        if self._ws is not None and self._ws_key == ws_key and self._ws["meta"].shape[0] >= cap:
            return self._ws
        self._ws = None  # release before allocating the replacement
        self._ws_key = ws_key
        bufs = {
            "boards": torch.empty((cap, B, 16), dtype=torch.uint8, device=d),
            "meta": torch.empty((cap, B), dtype=torch.uint8, device=d),
            "rewards": torch.empty((cap, B), dtype=torch.float32, device=d),
            "logp": torch.empty((cap, B), dtype=torch.float32, device=d) if with_logp else None,
            "values": torch.empty((cap, B), dtype=torch.float32, device=d) if with_values else None,
        }
        self._ws = bufs
        return bufs

    def _grow(self, bufs, used, new_cap):
        for k, v in bufs.items():
            if v is None:
                continue
            nv_ = torch.empty((new_cap,) + tuple(v.shape[1:]), dtype=v.dtype, device=v.device)
            nv_[:used].copy_(v[:used])
            bufs[k] = nv_
        self._ws = bufs
        return bufs

    def _finish(self, bufs, state, init_boards, B, fill_frozen, key0, n_splits_done_fn):
        boards, masks, done, ep_len = state
        T = int(ep_len.max().item())
        # the reference consumed exactly 1 + 2T splits (init + act/step per lock-step), T over the whole batch
        T_chain = T if self.global_max is None else int(self.global_max(T))
        self.key, _ = nv.chain_keys(key0, 1 + 2 * T_chain, self.rng_mode)
        return Trajectory(
            boards=bufs["boards"][:T], meta=bufs["meta"][:T], rewards=bufs["rewards"][:T],
            log_probs=None if bufs["logp"] is None else bufs["logp"][:T],
            values=None if bufs["values"] is None else bufs["values"][:T],
            ep_len=ep_len, final_boards=boards, final_masks=masks, init_boards=init_boards, T=T, B=B, frozen_filled=bool(fill_frozen))

    # ------------------------------------------------------------------ fused naive policies
    def rollout_fused(self, B: int, policy: int, fill_frozen: bool = False, chunk: int = 64,
                      B_total: Optional[int] = None, env0: int = 0, max_steps: int = 1 << 20) -> Trajectory:
        """B complete episodes with act_drul / act_randomly fused into the persistent step kernel."""
        if B <= 0:
            raise ValueError("batch_size must be positive")
        B_total = B if B_total is None else int(B_total)
        chunk = max(1, min(int(chunk), nv.MAX_FUSED_STEPS))
        key0 = self.key.copy()
        key, sub = nv.chain_keys(key0, 1, self.rng_mode)
        state = self._alloc_state(B)
        boards, masks, done, ep_len = state
        nv.reset_fused(sub[0], boards, masks, done, ep_len, B_total, env0, self.rng_mode)
        init_boards = boards.clone()
        bufs = self._alloc_traj(max(self.initial_capacity, chunk), B, with_logp=(policy == nv.POLICY_RANDOM),
                                with_values=False)
        cap = bufs["meta"].shape[0]
        live = torch.zeros(1, dtype=torch.int32, device=self.device)
        t = 0
        while True:
            if t + chunk > cap:
                cap *= 2
                bufs = self._grow(bufs, t, cap)
            key, subs = nv.chain_keys(key, 2 * chunk, self.rng_mode)
            live.zero_()
            nv.rollout_fused(subs.reshape(chunk, 4), t, boards, masks, done, ep_len, bufs["boards"], bufs["meta"],
                             bufs["rewards"], bufs["logp"], B_total, env0, policy, fill_frozen, self.rng_mode, live)
            t += chunk
            if int(live.item()) == 0:
                break
            if t >= max_steps:
                raise RuntimeError("rollout exceeded max_steps")
        return self._finish(bufs, state, init_boards, B, fill_frozen, key0, None)

    # ------------------------------------------------------------------ policy in the loop
    def rollout_policy(self, B: int, policy_fn: Callable, use_mask: bool, sample: bool = True,
                       fill_frozen: bool = False, sync_every: int = 8, B_total: Optional[int] = None,
                       env0: int = 0, max_steps: int = 1 << 20, compact: bool = True) -> Trajectory:
        """B complete episodes; ``policy_fn(boards u8[B,16], masks u8[B]) -> (logits f32[B,4], values f32[B])``
        runs on the device each lock-step, sampling + env step + trajectory write are one fused kernel."""
        if B <= 0:
            raise ValueError("batch_size must be positive")
        B_total = B if B_total is None else int(B_total)
        key0 = self.key.copy()
        key, sub = nv.chain_keys(key0, 1, self.rng_mode)
        state = self._alloc_state(B)
        boards, masks, done, ep_len = state
        nv.reset_fused(sub[0], boards, masks, done, ep_len, B_total, env0, self.rng_mode)
        init_boards = boards.clone()
        bufs = self._alloc_traj(max(self.initial_capacity, sync_every), B, with_logp=True, with_values=True)
        cap = bufs["meta"].shape[0]
        live = torch.zeros(1, dtype=torch.int32, device=self.device)
        # Policy inference only on envs that are still running (refreshed at every poll): the lock-step batch
        # keeps finished envs until the slowest one ends, but their logits are never used unless the caller
        # wants the reference's frozen frames (fill_frozen), so they are not computed.
        # (``compact`` False: a policy whose forward is cheaper than the two gathers and two scatters of the compaction.)
        compact = compact and not fill_frozen
        # (a policy that never reads the masks - TorchActionFunction.policy_fn: the kernel masks - is spared their gather per lock-step)
        needs_masks = bool(getattr(policy_fn, "needs_masks", True))
        live_idx = None
        logits_full = torch.zeros((B, 4), dtype=torch.float32, device=self.device)
        values_full = torch.zeros(B, dtype=torch.float32, device=self.device)
        t = 0
        while True:
            if t + sync_every > cap:
                cap *= 2
                bufs = self._grow(bufs, t, cap)
            key, subs = nv.chain_keys(key, 2 * sync_every, self.rng_mode)
            for s in range(sync_every):
                if live_idx is None:
                    logits, values = policy_fn(boards, masks)
                    logits = logits.to(torch.float32).contiguous()
                    values = values.to(torch.float32).reshape(-1).contiguous()
                else:
                    lg, vl = policy_fn(boards.index_select(0, live_idx), masks.index_select(0, live_idx) if needs_masks else None)
                    logits_full.index_copy_(0, live_idx, lg.to(torch.float32))
                    values_full.index_copy_(0, live_idx, vl.to(torch.float32).reshape(-1))
                    logits, values = logits_full, values_full
                poll = s == sync_every - 1  # only the launch the host reads back counts the live envs
                if poll:
                    live.zero_()
                nv.policy_step(subs[2 * s], subs[2 * s + 1], logits, values, use_mask, sample, t, boards, masks, done,
                               ep_len, bufs["boards"], bufs["meta"], bufs["rewards"], bufs["logp"], bufs["values"],
                               B_total, env0, fill_frozen, self.rng_mode, live if poll else None)
                t += 1
            n_live = int(live.item())
            if n_live == 0:
                break
            if t >= max_steps:
                raise RuntimeError("rollout exceeded max_steps")
            if compact and n_live < B:
                live_idx = torch.nonzero(done == 0).flatten()
        return self._finish(bufs, state, init_boards, B, fill_frozen, key0, None)


    # ------------------------------------------------------------------ fixed horizon, per-lane auto-reset
    def rollout_policy_fixed(self, B: int, T: int, policy_fn: Callable, use_mask: bool, sample: bool = True,
                             B_total: Optional[int] = None, env0: int = 0, restart: bool = False) -> FixedTrajectory:
        """T lock-steps of B always-live lanes (throughput mode; no reference counterpart: it replaces the
        ``while not all terminated`` of src/runs/batch_runner.py:117).  A lane whose step terminates is re-initialised in
        the same kernel from ``split(fold_in(step_sub, 0xFFFFFFFF), B_total)[env]``; the host key chain advances as in the
        reference (init split at the first call, then act/step splits per lock-step).  Env state persists across calls
        (``restart`` re-initialises it).  No host synchronisation inside the loop."""
        if B <= 0 or T <= 0:
            raise ValueError("batch_size and horizon must be positive")
        B_total = B if B_total is None else int(B_total)
        st = self._fixed
        if restart or st is None or st[3:] != (B, B_total, env0):
            self.key, sub = nv.chain_keys(self.key, 1, self.rng_mode)
            boards, masks, done, ep_len = self._alloc_state(B)
            nv.reset_fused(sub[0], boards, masks, done, ep_len, B_total, env0, self.rng_mode)
            self._fixed = st = (boards, masks, ep_len, B, B_total, env0)
        boards, masks, ep_len = st[:3]
        ep_before = ep_len.clone()
        bufs = self._alloc_traj(T, B, with_logp=True, with_values=True)
        self.key, subs = nv.chain_keys(self.key, 2 * T, self.rng_mode)
        for t in range(T):
            logits, values = policy_fn(boards, masks)
            nv.policy_step_autoreset(subs[2 * t], subs[2 * t + 1], logits.to(torch.float32).contiguous(),
                                     values.to(torch.float32).reshape(-1).contiguous(), use_mask, sample, t, boards, masks,
                                     ep_len, bufs["boards"], bufs["meta"], bufs["rewards"], bufs["logp"], bufs["values"],
                                     B_total, env0, self.rng_mode)
        return FixedTrajectory(boards=bufs["boards"][:T], meta=bufs["meta"][:T], rewards=bufs["rewards"][:T],
                               log_probs=bufs["logp"][:T], values=bufs["values"][:T], final_boards=boards, final_masks=masks,
                               ep_len=ep_len, ep_len_before=ep_before, T=T, B=B)


# ---------------------------------------------------------------------------------------------------
# reference-layout views of a trajectory (only for callers that want the reference's numpy arrays)
# ---------------------------------------------------------------------------------------------------
def one_hot_observations(boards: torch.Tensor) -> torch.Tensor:
    """u8 [..., 16] boards -> bool [..., 4, 4, 31] one-hot (State.observation)."""
    flat = boards.reshape(-1, 16).contiguous()
    obs = torch.empty((flat.shape[0], 4, 4, 31), dtype=torch.uint8, device=boards.device)
    nv.observe(flat, obs)
    return obs.view(*boards.shape[:-1], 4, 4, 31).to(torch.bool)


def mask_bits_to_bool(masks: torch.Tensor) -> torch.Tensor:
    """u8 [...] bitmask -> bool [..., 4]."""
    bits = torch.tensor([1, 2, 4, 8], dtype=torch.uint8, device=masks.device)
    return (masks.unsqueeze(-1) & bits) != 0
