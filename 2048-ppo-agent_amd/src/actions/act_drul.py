"""Down-Right-Up-Left heuristic (act_fn plug-in; reference src/actions/act_drul.py:5-49)."""
import numpy as np
import torch

from ..g2048 import native as nv
from . import _common as C


def act_drul(rng_key, obs, mask):
    """First legal action in the priority order down (3), right (2), up (1), left (0).

    Same protocol as the reference: un-batched ``(rng_key, obs[4,4,31], mask[4]) -> (action, None, None)``;
    the key and the observation are ignored.  Action indices: 0 left, 1 up, 2 right, 3 down.
    A leading batch dimension on ``mask`` is accepted too (what ``jax.vmap`` gave the reference).
    BatchRunner recognises this function and runs it fused inside the rollout kernel.
    """
    shape = np.shape(obs)
    assert tuple(shape[-3:]) == (4, 4, 31), f"obs must be (4, 4, 31), got {shape}"
    assert np.shape(mask)[-1] == 4, "mask must be (4,)"
    bits = C.mask_to_bits(mask)
    actions = torch.empty(bits.numel(), dtype=torch.int32, device=bits.device)
    nv.act_drul(bits, actions)
    out = actions.cpu().numpy()
    action = out.reshape(np.shape(mask)[:-1]) if np.ndim(mask) > 1 else np.int32(out[0])
    return action, None, None


act_drul.fused_policy = nv.POLICY_DRUL
