"""GAE + minibatching for PPO (API of the reference src/ppo/data_loader.py:8-223) on the device.

The reverse GAE scan is the HIP kernel ``g2048_gae_flat`` (bit-identical to the reference's float32 scan);
advantages AND returns are then z-scored with the unbiased std, as the reference does.  All tensors stay in
HBM.  ``PPODataset``/``create_ppo_dataloader`` keep the reference's per-sample Dataset protocol;
``DeviceBatches`` is the path the trainer uses (device randperm + index_select, no per-sample Python).
"""
from __future__ import annotations

from typing import Dict, Optional

import numpy as np
import torch
from torch.utils.data import DataLoader, Dataset

from ..g2048 import native as nv


def _device():
    if not torch.cuda.is_available():
        raise nv.NativeError("GAE runs on the MI355X (g2048_gae_flat); no HIP device is visible, no CPU path exists")
    return torch.device("cuda", torch.cuda.current_device())


def compute_gae(rewards: torch.Tensor, values: torch.Tensor, terms: torch.Tensor, gamma: float, lam: float):
    """Raw (un-normalised) advantages and returns of a flat buffer; device tensors in, device tensors out."""
    rewards = rewards.to(torch.float32).contiguous()
    values = values.to(torch.float32).contiguous()
    terms = terms.to(torch.uint8).contiguous()
    adv = torch.empty_like(rewards)
    ret = torch.empty_like(rewards)
    if rewards.numel():
        nv.gae_flat(rewards, values, terms, adv, ret, gamma, lam)
    return adv, ret


def zscore(x: torch.Tensor, group=None) -> torch.Tensor:
    """(x - mean) / (unbiased std + 1e-8).  With a process group the statistics are global over all ranks
    (one 3-double all-reduce), which keeps a sharded run identical to the single-device normalisation."""
    if group is None:
        return (x - x.mean()) / (x.std() + 1e-8)
    import torch.distributed as dist

    x64 = x.double()
    s = torch.stack([x64.sum(), (x64 * x64).sum(), torch.tensor(float(x.numel()), dtype=torch.float64, device=x.device)])
    dist.all_reduce(s, group=group)
    mean = s[0] / s[2]
    var = (s[1] - s[2] * mean * mean) / (s[2] - 1)
    return ((x64 - mean) / (var.clamp_min(0).sqrt() + 1e-8)).float()


class PPODataset(Dataset):
    """Reference-compatible dataset over ``RolloutBuffer.get_buffer_data()`` (numpy dict) or
    ``RolloutBuffer.device_data()`` (packed device dict)."""

    def __init__(self, buffer_data: Dict, gamma: float = 0.99, lambda_gae: float = 0.95,
                 max_samples_per_epoch: int = None, shuffle_on_reset: bool = False, group=None):
        self.gamma, self.lambda_gae = gamma, lambda_gae
        self.max_samples_per_epoch, self.shuffle_on_reset = max_samples_per_epoch, shuffle_on_reset
        dev = _device()
        if "boards" in buffer_data:  # packed device dict
            self.packed = True
            self.observations = buffer_data["boards"]
            self.actions = buffer_data["actions"]
            self.action_masks = buffer_data["masks"]
            self.terminations = buffer_data["terms"].to(torch.bool)
        else:
            self.packed = False
            as_t = lambda k, dt: torch.from_numpy(np.ascontiguousarray(buffer_data[k])).to(device=dev, dtype=dt)
            self.observations = as_t("observations", torch.float32)
            self.actions = as_t("actions", torch.float32)
            self.action_masks = as_t("action_masks", torch.bool)
            self.terminations = as_t("terminations", torch.bool)
        get = lambda k: (buffer_data[k] if isinstance(buffer_data[k], torch.Tensor)
                         else torch.from_numpy(np.ascontiguousarray(buffer_data[k]))).to(device=dev, dtype=torch.float32)
        self.rewards, self.values, self.log_probs = get("rewards"), get("values"), get("log_probs")
        if "raw_advantages" in buffer_data and "raw_returns" in buffer_data:
            # the buffer already ran the scan on the coalesced [T][B] trajectory (RolloutBuffer.store_trajectory with
            # gamma / lambda, or the bootstrapped scan of the fixed-horizon mode)
            self.raw_advantages, self.raw_returns = get("raw_advantages"), get("raw_returns")
        else:
            self.raw_advantages, self.raw_returns = compute_gae(self.rewards, self.values, self.terminations,
                                                               gamma, lambda_gae)
        self.advantages = zscore(self.raw_advantages, group)
        self.returns = zscore(self.raw_returns, group)
        self.total_length = int(self.rewards.shape[0])
        if self.max_samples_per_epoch is None or self.max_samples_per_epoch >= self.total_length:
            self.length = self.total_length
            self.active_indices = None
        else:
            self.length = int(self.max_samples_per_epoch)
            self.active_indices = self._sample_indices()

    def _sample_indices(self) -> torch.Tensor:
        return torch.randperm(self.total_length, device=self.rewards.device)[: self.length]

    def reset_epoch(self):
        if self.shuffle_on_reset and self.active_indices is not None:
            self.active_indices = self._sample_indices()

    def __len__(self) -> int:
        return self.length

    def __getitem__(self, idx: int) -> Dict[str, torch.Tensor]:
        i = self.active_indices[idx] if self.active_indices is not None else idx
        return {
            "observations": self.observations[i], "actions": self.actions[i], "action_masks": self.action_masks[i],
            "rewards": self.rewards[i], "values": self.values[i], "log_probs": self.log_probs[i],
            "terminations": self.terminations[i], "advantages": self.advantages[i], "returns": self.returns[i],
        }


def create_ppo_dataloader(buffer_data: Dict, gamma: float = 0.99, lambda_gae: float = 0.95, batch_size: int = 32,
                          shuffle: bool = True, drop_last: bool = True, num_workers: int = 0,
                          max_samples_per_epoch: int = None, shuffle_on_reset: bool = False) -> DataLoader:
    """Reference-compatible factory.  The dataset's tensors live on the device, so ``num_workers`` must be 0."""
    dataset = PPODataset(buffer_data, gamma=gamma, lambda_gae=lambda_gae,
                         max_samples_per_epoch=max_samples_per_epoch, shuffle_on_reset=shuffle_on_reset)
    if num_workers:
        raise ValueError("device-resident dataset: num_workers must be 0")
    return DataLoader(dataset, batch_size=batch_size, shuffle=shuffle, drop_last=drop_last, num_workers=0)


class DeviceBatches:
    """Minibatch iterator over a PPODataset without leaving the device: per epoch one randperm (subset when
    ``max_samples_per_epoch`` applies, re-drawn if ``shuffle_on_reset``), then index_select per minibatch."""

    def __init__(self, dataset: PPODataset, batch_size: int, drop_last: bool = True):
        self.ds, self.batch_size, self.drop_last = dataset, int(batch_size), drop_last

    def __len__(self):
        n = len(self.ds)
        return n // self.batch_size if self.drop_last else -(-n // self.batch_size)

    def indices(self):
        """One epoch of minibatch index tensors (int64 on the data's device)."""
        ds = self.ds
        ds.reset_epoch()
        n = len(ds)
        order = torch.randperm(n, device=ds.rewards.device)
        if ds.active_indices is not None:
            order = ds.active_indices[order]
        stop = n - (n % self.batch_size) if self.drop_last else n
        for s in range(0, stop, self.batch_size):
            yield order[s:s + self.batch_size]

    def packed(self) -> bool:
        """Device-resident packed layout (boards u8 [N, 16], actions / mask bits u8 [N]): ``gather_packed`` applies."""
        ds = self.ds
        return (ds.observations.is_cuda and ds.observations.dtype == torch.uint8 and ds.observations.dim() == 2
                and ds.observations.shape[1] == 16 and ds.actions.dtype == torch.uint8 and ds.actions.dim() == 1
                and ds.action_masks.dtype == torch.uint8 and ds.action_masks.dim() == 1)

    def gather_packed(self, idx: torch.Tensor, out=None) -> dict:
        """-> dict(obs, actions, masks, old_lp, adv, ret) for the samples ``idx`` in ONE launch (``g2048_gather_minibatch``),
        optionally straight into pre-allocated tensors (the static inputs of the captured update)."""
        ds = self.ds
        return nv.gather_minibatch(idx.contiguous(), ds.observations, ds.actions, ds.action_masks, ds.log_probs, ds.advantages,
                                   ds.returns, out)

    def gather(self, idx: torch.Tensor) -> dict:
        ds = self.ds
        return {
            "observations": ds.observations.index_select(0, idx), "actions": ds.actions.index_select(0, idx),
            "action_masks": ds.action_masks.index_select(0, idx), "log_probs": ds.log_probs.index_select(0, idx),
            "advantages": ds.advantages.index_select(0, idx), "returns": ds.returns.index_select(0, idx),
        }

    def epoch(self):
        for idx in self.indices():
            yield self.gather(idx)
