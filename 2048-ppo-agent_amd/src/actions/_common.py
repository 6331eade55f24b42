"""Shared plumbing for the act_fn plug-ins: run the batched device kernel on un-batched host inputs."""
import numpy as np
import torch

from ..g2048 import native as nv


def default_rng_mode() -> int:
    import os

    m = os.environ.get("G2048_RNG_MODE", "partitionable").lower()
    return nv.RNG_LEGACY if m in ("legacy", "0") else nv.RNG_PARTITIONABLE


def device():
    if not torch.cuda.is_available():
        raise nv.NativeError("act_fn plug-ins run on the MI355X; no HIP device is visible and no CPU path exists")
    return torch.device("cuda", torch.cuda.current_device())


def mask_to_bits(mask) -> torch.Tensor:
    """bool [..., 4] (numpy or torch) -> u8 bitmask tensor [...] on the device."""
    m = torch.as_tensor(np.asarray(mask.cpu() if isinstance(mask, torch.Tensor) else mask)).to(torch.bool)
    if m.shape[-1] != 4:
        raise AssertionError(f"mask must have 4 entries, got shape {tuple(m.shape)}")
    w = torch.tensor([1, 2, 4, 8], dtype=torch.uint8)
    return (m.to(torch.uint8) * w).sum(-1).to(torch.uint8).reshape(-1).to(device())


def keys_tensor(rng_key) -> torch.Tensor:
    """key words u32 [2] or [B, 2] -> device key tensor [B, 2]."""
    k = np.asarray(rng_key.cpu() if isinstance(rng_key, torch.Tensor) else rng_key)
    return nv.keys_from_numpy(k.reshape(-1, 2), device())
