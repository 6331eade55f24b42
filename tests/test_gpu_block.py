"""g2048_block_fwd (csrc/g2048_block.hip) through its autograd node against a plain PyTorch composition of the same operators
(reference: nn.TransformerEncoderLayer(norm_first=True) of src/ppo/transformer_encoder.py:138-148 after its attention, plus norm1 of the
following layer)."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from src.g2048 import native as nv
from src.ppo.hip_ops import BLOCK_PARAM_ORDER, BlockPlan, _BlockFFN
from tests.test_gpu_tail import _rel

pytestmark = pytest.mark.gpu
SHAPES = dict(wo=(256, 256), bo=(256,), ln2_g=(256,), ln2_b=(256,), w1=(1024, 256), b1=(1024,), w2=(256, 1024), b2=(256,),
              lnn_g=(256,), lnn_b=(256,))


def _params(dev, seed):
    g = torch.Generator().manual_seed(seed)
    P = {}
    for k, shp in SHAPES.items():
        if k.endswith("_g"):
            t = 1.0 + 0.1 * torch.randn(shp, generator=g)
        elif len(shp) == 1:
            t = 0.1 * torch.randn(shp, generator=g)
        else:
            t = torch.randn(shp, generator=g) / shp[1] ** 0.5
        P[k] = t.to(dev).requires_grad_(True)
    return P


def _hash(idx, s0, s1):
    M32 = np.uint64(0xFFFFFFFF)
    x = ((idx & M32) * np.uint64(0x9E3779B1) & M32) ^ np.uint64(s0)
    x ^= (((idx >> np.uint64(32)) * np.uint64(0x85EBCA77)) + np.uint64(s1)) & M32
    x ^= x >> np.uint64(16)
    x = x * np.uint64(0x7FEB352D) & M32
    x ^= x >> np.uint64(15)
    x = x * np.uint64(0x846CA68B) & M32
    x ^= x >> np.uint64(16)
    return x


def _keep_elem(seed, rows, cols, p):
    """csrc/g2048_layernorm.hip keep_elem: one 24-bit decision per element."""
    if p == 0:
        return np.ones((rows, cols), bool)
    thr = np.uint64(int(np.float32(p) * np.float32(16777216.0)))
    x = _hash(np.arange(rows * cols, dtype=np.uint64), int(seed) & 0xFFFFFFFF, int(seed) >> 32)
    return ((x >> np.uint64(8)) >= thr).reshape(rows, cols)


@pytest.mark.parametrize("B,p", [(13, 0.0), (13, 0.1), (2048, 0.1)])
def test_block_node_matches_torch(dev, B, p):
    """Forward tensors (x_mid, h2, u, x_out, h_next), input gradients and parameter gradients of the fused node vs the fp32
    composition with the kernel's dropout masks (sites 1 / 3: the layer-norm kernels' per-element hash; site 2: the non-zero pattern
    of u, which also fixes the ReLU's active set).  B = 13 boards = 221 tokens: a partial last workgroup and token block."""
    torch.manual_seed(B)
    P = _params(dev, seed=B)
    T = 17 * B
    a = torch.randn(B, 17, 256, device=dev).to(torch.bfloat16).requires_grad_(True)
    x = torch.randn(B, 17, 256, device=dev).requires_grad_(True)
    bf = {k: P[k].detach().to(torch.bfloat16).contiguous() for k in ("wo", "w1", "w2")}
    plan = BlockPlan(P, {k: nv.pack_fragments(bf[k]) for k in bf}, dict(wo=bf["wo"], w1=bf["w1"], w2=bf["w2"], w2T=bf["w2"].t().contiguous()),
                     1e-5, 1e-5, p)
    import src.ppo.hip_ops as ho

    seeds, orig = [], ho._seed_pair
    ho._seed_pair = lambda t, pd: seeds.append(orig(t, pd)) or seeds[-1]
    try:
        x_out, h_next = _BlockFFN.apply(a, x, plan, *[P[k] for k in BLOCK_PARAM_ORDER])
    finally:
        ho._seed_pair = orig
    g_x, g_h = torch.randn(B, 17, 256, device=dev), torch.randn(B, 17, 256, device=dev).to(torch.bfloat16)
    saved = x_out.grad_fn.saved_tensors  # a2, x_mid, mean2, rstd2, h2, u, x_out, mean_n, rstd_n
    x_mid_k, h2_k, u_k = saved[1].clone(), saved[4].clone(), saved[5].clone()
    torch.autograd.backward([x_out, h_next], [g_x, g_h])
    # ---- reference
    inv = float(np.float32(1.0) / (np.float32(1.0) - np.float32(p)))
    k1 = torch.from_numpy(_keep_elem(seeds[0][0], T, 256, p)).to(dev).float() * inv
    k3 = torch.from_numpy(_keep_elem(seeds[2][0], T, 256, p)).to(dev).float() * inv
    act = (u_k != 0).float()
    W = {k: (P[k].detach().to(torch.bfloat16).float() if P[k].dim() == 2 else P[k].detach().clone()).requires_grad_(True) for k in P}
    a_r = a.detach().float().reshape(T, 256).requires_grad_(True)
    x_r = x.detach().reshape(T, 256).clone().requires_grad_(True)
    x_mid = x_r + F.linear(a_r, W["wo"], W["bo"]) * k1
    h2 = F.layer_norm(x_mid, (256,), W["ln2_g"], W["ln2_b"], 1e-5)
    u = F.linear(h2, W["w1"], W["b1"]) * act * inv
    x_o = x_mid + F.linear(u, W["w2"], W["b2"]) * k3
    h_n = F.layer_norm(x_o, (256,), W["lnn_g"], W["lnn_b"], 1e-5)
    torch.autograd.backward([x_o, h_n], [g_x.reshape(T, 256), g_h.float().reshape(T, 256)])
    err = {"x_mid": _rel(x_mid_k, x_mid.detach()), "h2": _rel(h2_k, h2.detach()), "u": _rel(u_k, u.detach()),
           "x_out": _rel(x_out.detach().reshape(T, 256), x_o.detach()), "h_next": _rel(h_next.detach().reshape(T, 256), h_n.detach()),
           "d_a": _rel(a.grad.reshape(T, 256), a_r.grad), "d_x": _rel(x.grad.reshape(T, 256), x_r.grad)}
    for k in BLOCK_PARAM_ORDER:
        assert P[k].grad is not None and P[k].grad.shape == P[k].shape, k
        err["d_" + k] = _rel(P[k].grad, W[k].grad)
    assert all(v < 2e-2 for v in err.values()), {k: round(v, 4) for k, v in err.items()}
    # the hidden pattern the hash keeps is a superset of what is active
    if p > 0:
        assert 0.85 < float((u_k != 0).float().mean()) / max(float((u.detach() != 0).float().mean()), 1e-9) < 1.15
