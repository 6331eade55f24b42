"""Deterministic weights for a policy of any shape, from nothing but the state-dict's names and shapes.

The default-shape agent (d_model 256, 4 layers, ff 1024: 3.96 M parameters = 15.8 MB) is too large to commit as a
fixture.  Instead both sides build the SAME state-dict from this recipe: tests/golden/make_golden_torch.py loads it into
the reference's PPOAgent (build container only) and stores inputs + outputs; the tests load it into this repository's
PPOAgent and must reproduce those outputs.  Values are drawn per tensor from numpy's PCG64 stream seeded with
(seed, crc32(name)), so the result does not depend on iteration order; buffers (positional code) are left alone.
"""
import zlib

import numpy as np


def fill_state_dict(shapes: dict, seed: int = 1234) -> dict:
    """shapes: {parameter name: shape tuple}  ->  {name: float32 ndarray}."""
    out = {}
    for name, shape in shapes.items():
        if "positional_encoding" in name:
            continue
        rng = np.random.default_rng([seed, zlib.crc32(name.encode())])
        shape = tuple(int(s) for s in shape)
        if name.endswith("cls_token"):
            w = rng.normal(0.0, 1.0, size=shape)
        elif "norm" in name and name.endswith("weight"):
            w = 1.0 + rng.normal(0.0, 0.1, size=shape)
        elif name.endswith("bias"):
            w = rng.normal(0.0, 0.1, size=shape)
        else:  # Linear / in_proj weights [out, in]
            w = rng.normal(0.0, 1.0 / np.sqrt(shape[-1]), size=shape)
        out[name] = w.astype(np.float32)
    return out


def sample_boards(n: int, seed: int = 7) -> np.ndarray:
    """u8 [n, 16] log2 tiles with a game-like distribution: many empties, geometric tiles, a few large ones."""
    rng = np.random.default_rng(seed)
    e = rng.geometric(0.35, size=(n, 16)).clip(1, 13)
    e[rng.random((n, 16)) < 0.3] = 0
    e[0] = 0  # empty board
    e[1] = np.arange(1, 17).clip(1, 15)  # full board, large tiles
    return e.astype(np.uint8)
