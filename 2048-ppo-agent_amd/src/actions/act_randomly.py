"""Uniform-over-legal-actions baseline (act_fn plug-in; reference src/actions/act_randomly.py:5-56)."""
import numpy as np
import torch

from ..g2048 import native as nv
from . import _common as C


def act_randomly(rng_key, obs, mask, rng_mode=None):
    """Sample uniformly among legal actions with the JAX-compatible categorical draw.

    Same protocol as the reference: un-batched ``(rng_key, obs[4,4,31], mask[4]) -> (action, log_prob, None)``
    where ``rng_key`` holds the two threefry key words; ``log_prob = log(1 / n_legal)`` (all four actions
    when none is legal).  Leading batch dimensions on ``rng_key``/``mask`` are accepted.
    BatchRunner recognises this function and runs it fused inside the rollout kernel.
    """
    shape = np.shape(obs)
    assert tuple(shape[-3:]) == (4, 4, 31), f"obs must be (4, 4, 31), got {shape}"
    assert np.shape(mask)[-1] == 4, "mask must be (4,)"
    mode = C.default_rng_mode() if rng_mode is None else rng_mode
    bits = C.mask_to_bits(mask)
    keys = C.keys_tensor(rng_key)
    if keys.shape[0] != bits.numel():
        raise ValueError("one key per mask row is required")
    actions = torch.empty(bits.numel(), dtype=torch.int32, device=bits.device)
    logp = torch.empty(bits.numel(), dtype=torch.float32, device=bits.device)
    nv.act_random(keys, bits, actions, logp, mode)
    a, lp = actions.cpu().numpy(), logp.cpu().numpy()
    if np.ndim(mask) > 1:
        lead = np.shape(mask)[:-1]
        return a.reshape(lead), lp.reshape(lead), None
    return np.int32(a[0]), np.float32(lp[0]), None


act_randomly.fused_policy = nv.POLICY_RANDOM
