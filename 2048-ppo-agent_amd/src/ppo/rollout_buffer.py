"""Trajectory store for PPO (API of the reference src/ppo/rollout_buffer.py:4-206) with a device-resident path.

Two ways in:
* ``store_batch(...)``       -- the reference's numpy interface ([B, T, ...] host arrays).
* ``store_trajectory(traj)`` -- an engine ``Trajectory`` in HBM; compaction (keep steps 0..first termination
  of every env, env-major) is the HIP kernel ``g2048_compact`` and nothing leaves the device.
Two ways out: ``get_buffer_data()`` (the reference's dict of numpy arrays) and ``device_data()`` (packed
device tensors for the trainer).
"""
from __future__ import annotations

import numpy as np
import torch

from ..g2048 import native as nv
from ..g2048.engine import Trajectory, mask_bits_to_bool


class RolloutBuffer:
    def __init__(self, observation_dim: int, observation_length, action_dim: int) -> None:
        self.observation_dim = observation_dim
        self.observation_length = observation_length
        self.action_dim = action_dim
        self.reset()

    def reset(self):
        # host segments (reference-layout numpy) and device segments (packed tensors), in arrival order
        self._segments = []
        self.buffer_size = 0

    # ------------------------------------------------------------------ reference numpy interface
    def _expected_obs_dims(self):
        if isinstance(self.observation_length, (tuple, list)):
            return (*self.observation_length, self.observation_dim)
        return (self.observation_length, self.observation_dim)

    def _validate_and_reshape_observations(self, observations: np.ndarray) -> np.ndarray:
        """Accept [B, T, *obs_dims] or anything with the same number of elements per step."""
        want = self._expected_obs_dims()
        if observations.ndim < 2:
            raise ValueError("Observations must have at least 2 dimensions (batch_size, time_steps, ...), "
                             f"but got shape {observations.shape}")
        if observations.shape[2:] == want:
            return observations
        B, T = observations.shape[:2]
        per_step = int(np.prod(observations.shape[2:])) if observations.ndim > 2 else 1
        if per_step != int(np.prod(want)):
            raise ValueError(f"Failed to reshape observations from shape {observations.shape} to expected shape "
                             f"(batch_size, time_steps, {want}). Error: Cannot reshape observations: "
                             f"{per_step} elements per timestep but expected {int(np.prod(want))} elements.")
        return observations.reshape(B, T, *want)

    def store_batch(self, observations, actions, action_masks, rewards, values, log_probs, terminations):
        """Keep, for every env, steps 0..first termination (inclusive); envs that never terminate are dropped.
        Kept steps are appended env-major (all of env 0, then env 1, ...)."""
        observations = self._validate_and_reshape_observations(np.asarray(observations))
        terminations = np.asarray(terminations).astype(bool)
        B, T = terminations.shape[:2]
        has = terminations.any(axis=1)
        lens = np.where(has, terminations.argmax(axis=1) + 1, 0)
        keep = np.arange(T)[None, :] < lens[:, None]  # [B, T]; boolean indexing walks it env-major
        seg = {
            "observations": np.asarray(observations)[keep], "actions": np.asarray(actions)[keep],
            "action_masks": np.asarray(action_masks)[keep], "rewards": np.asarray(rewards)[keep],
            "values": np.asarray(values)[keep], "log_probs": np.asarray(log_probs)[keep],
            "terminations": terminations[keep],
        }
        n = int(lens.sum())
        if n:
            self._segments.append(("host", seg))
            self.buffer_size += n

    # ------------------------------------------------------------------ device interface
    def store_trajectory(self, traj: Trajectory, gamma: float = None, lambda_gae: float = None) -> int:
        """Compact an engine trajectory into the buffer on the device; returns the number of kept steps.
        With ``gamma``/``lambda_gae`` the raw GAE advantages and returns are computed first, on the step-major [T][B] layout
        where the scan is fully coalesced (``g2048_gae_tb``, one lane per env, bit-identical to the reference's scan over
        the compacted buffer), and compacted along: PPODataset then skips its own scan."""
        if traj.log_probs is None or traj.values is None:
            raise ValueError("PPO needs log_probs and values; this trajectory has none (naive policy?)")
        dev = traj.ep_len.device
        lens = traj.ep_len.to(torch.int64)
        offsets = torch.cumsum(lens, 0) - lens
        N = int(lens.sum().item())
        if N == 0:
            return 0
        with_gae = gamma is not None and lambda_gae is not None
        tr_adv = tr_ret = None
        if with_gae:
            tr_adv, tr_ret = torch.empty_like(traj.rewards), torch.empty_like(traj.rewards)
            nv.gae_tb(traj.rewards, traj.values, traj.ep_len, tr_adv, tr_ret, traj.T, traj.B, gamma, lambda_gae)
        out = {
            "boards": torch.empty((N, 16), dtype=torch.uint8, device=dev),
            "actions": torch.empty(N, dtype=torch.uint8, device=dev),
            "masks": torch.empty(N, dtype=torch.uint8, device=dev),
            "rewards": torch.empty(N, dtype=torch.float32, device=dev),
            "log_probs": torch.empty(N, dtype=torch.float32, device=dev),
            "values": torch.empty(N, dtype=torch.float32, device=dev),
            "terms": torch.empty(N, dtype=torch.uint8, device=dev),
        }
        if with_gae:
            out["raw_advantages"] = torch.empty(N, dtype=torch.float32, device=dev)
            out["raw_returns"] = torch.empty(N, dtype=torch.float32, device=dev)
        nv.compact(traj.boards, traj.meta, traj.rewards, traj.log_probs, traj.values, traj.ep_len, offsets,
                   out["boards"], out["actions"], out["masks"], out["rewards"], out["log_probs"], out["values"],
                   out["terms"], traj.T, traj.B, N, tr_adv=tr_adv, tr_ret=tr_ret, out_adv=out.get("raw_advantages"),
                   out_ret=out.get("raw_returns"))
        self._segments.append(("device", out))
        self.buffer_size += N
        return N

    def store_fixed_trajectory(self, traj, last_values: torch.Tensor, gamma: float, lambda_gae: float) -> int:
        """A fixed-horizon trajectory (``FixedTrajectory``: all T x B rows are samples, ``terms`` marks the episode
        boundaries): GAE on the [T][B] layout bootstrapped from ``last_values`` = V(state after the last step)
        (``g2048_gae_tb_boot``), then the rows are taken over as they lie (step-major; copied, because the engine reuses
        its trajectory workspace)."""
        T, B = traj.T, traj.B
        adv, ret = torch.empty_like(traj.rewards), torch.empty_like(traj.rewards)
        nv.gae_tb_boot(traj.rewards, traj.values, traj.meta, last_values.to(torch.float32).reshape(-1).contiguous(), adv, ret,
                       T, B, gamma, lambda_gae)
        N = T * B
        out = {
            "boards": traj.boards.reshape(N, 16).clone(), "actions": (traj.meta & 3).reshape(N),
            "masks": ((traj.meta >> 2) & 15).reshape(N), "rewards": traj.rewards.reshape(N).clone(),
            "log_probs": traj.log_probs.reshape(N).clone(), "values": traj.values.reshape(N).clone(),
            "terms": ((traj.meta >> 6) & 1).reshape(N), "raw_advantages": adv.reshape(N), "raw_returns": ret.reshape(N),
        }
        self._segments.append(("device", out))
        self.buffer_size += N
        return N

    def device_data(self, device=None) -> dict:
        """All kept steps as packed device tensors: boards u8 [N,16], actions/masks/terms u8 [N], f32 [N] x3."""
        parts = []
        for kind, seg in self._segments:
            if kind == "device":
                parts.append(seg)
            else:
                parts.append(_host_segment_to_device(seg, device))
        if not parts:
            raise ValueError("rollout buffer is empty")
        keys = [k for k in parts[0] if all(k in p for p in parts)]  # raw GAE columns only if every segment carries them
        return {k: (parts[0][k] if len(parts) == 1 else torch.cat([p[k] for p in parts])) for k in keys}

    def get_buffer_data(self):
        """The reference's dict of numpy arrays: observations f32 [N, *obs_dims] (one-hot), actions f32 [N, 4]
        (one-hot when stored from the device), action_masks bool [N, 4], rewards/values/log_probs f32 [N],
        terminations bool [N]."""
        want = self._expected_obs_dims()
        parts = [seg if kind == "host" else _device_segment_to_host(seg, want) for kind, seg in self._segments]
        keys = ("observations", "actions", "action_masks", "rewards", "values", "log_probs", "terminations")
        dtypes = dict(observations=np.float32, actions=np.float32, action_masks=bool, rewards=np.float32,
                      values=np.float32, log_probs=np.float32, terminations=bool)
        if not parts:
            return {k: np.array([], dtype=dtypes[k]) for k in keys}
        return {k: np.concatenate([np.asarray(p[k], dtype=dtypes[k]) for p in parts], axis=0) for k in keys}


def _device_segment_to_host(seg: dict, obs_dims) -> dict:
    boards = seg["boards"].cpu().numpy()
    n_cls = obs_dims[-1]
    obs = (boards[:, :, None] == np.arange(n_cls, dtype=np.uint8)).astype(np.float32).reshape(len(boards), *obs_dims)
    return {
        "observations": obs,
        "actions": np.eye(4, dtype=np.float32)[seg["actions"].cpu().numpy()],
        "action_masks": mask_bits_to_bool(seg["masks"]).cpu().numpy(),
        "rewards": seg["rewards"].cpu().numpy(), "values": seg["values"].cpu().numpy(),
        "log_probs": seg["log_probs"].cpu().numpy(), "terminations": seg["terms"].cpu().numpy().astype(bool),
    }


def _host_segment_to_device(seg: dict, device) -> dict:
    if device is None:
        device = torch.device("cuda", torch.cuda.current_device())
    obs = np.asarray(seg["observations"])
    boards = obs.reshape(len(obs), 16, -1).argmax(-1).astype(np.uint8)
    acts = np.asarray(seg["actions"])
    acts = acts.argmax(-1) if acts.ndim > 1 else acts
    masks = (np.asarray(seg["action_masks"]).astype(np.uint8) * np.array([1, 2, 4, 8], np.uint8)).sum(-1)
    t = lambda a, dt: torch.from_numpy(np.ascontiguousarray(a)).to(device=device, dtype=dt)
    return {"boards": t(boards, torch.uint8), "actions": t(acts, torch.uint8), "masks": t(masks, torch.uint8),
            "rewards": t(seg["rewards"], torch.float32), "log_probs": t(seg["log_probs"], torch.float32),
            "values": t(seg["values"], torch.float32), "terms": t(seg["terminations"], torch.uint8)}
