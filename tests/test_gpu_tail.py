"""The fused CLS tail of the update (g2048_cls_tail_fwd / _bwd + g2048_dweight_t, csrc/g2048_tail.hip) against plain PyTorch
compositions of the same operators (reference: nn.TransformerEncoderLayer(norm_first=True) of src/ppo/transformer_encoder.py:138-148
after its attention, and the actor / critic heads of src/ppo/ppo_agent.py:62-92)."""
import copy

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from src.g2048 import native as nv
from src.ppo import PPOAgent
from src.ppo.hip_ops import TAIL_PARAM_ORDER, GradSink, TailBufferCache, TailPlan, _ClsTailHeads, grad_sink

pytestmark = pytest.mark.gpu
SHAPES = dict(wo=(256, 256), bo=(256,), ln_g=(256,), ln_b=(256,), w1=(1024, 256), b1=(1024,), w2=(256, 1024), b2=(256,),
              a1=(512, 256), ab1=(512,), a2=(512, 512), ab2=(512,), a3=(4, 512), c1=(512, 256), cb1=(512,), c2=(512, 512),
              cb2=(512,), c3=(1, 512))


def _params(dev, seed=0):
    g = torch.Generator().manual_seed(seed)
    P = {}
    for k, shp in SHAPES.items():
        if k == "ln_g":
            t = 1.0 + 0.1 * torch.randn(shp, generator=g)
        elif len(shp) == 1:
            t = 0.1 * torch.randn(shp, generator=g)
        else:
            t = torch.randn(shp, generator=g) / shp[1] ** 0.5
        P[k] = t.to(dev).requires_grad_(True)
    return P


def _plan(P, p_drop, cache=None):
    w = [k for k in SHAPES if len(SHAPES[k]) == 2]
    bf = {k: P[k].detach().to(torch.bfloat16).contiguous() for k in w}
    # fragment-packed copies of the weights / of their transposes (what the optimiser kernel maintains); a3 / c3 stay row-major
    dense = {k: (bf[k] if k in ("a3", "c3") else nv.pack_fragments(bf[k])) for k in w}
    transposed = {k: nv.pack_fragments(bf[k].t()) for k in w if k not in ("a3", "c3")}
    return TailPlan(P, dense, transposed, TailBufferCache() if cache is None else cache, 1e-5, p_drop)


def _keep(seed, site, rows, cols, p):
    """numpy replica of the kernels' dropout decision (Drop::site / Drop::apply4 in csrc/g2048_tail.hip: one hash per PAIR of
    consecutive elements, its 16-bit halves against the threshold at 16-bit resolution) -> bool [rows, cols]."""
    if p == 0:
        return np.ones((rows, cols), bool)
    M32 = np.uint64(0xFFFFFFFF)
    s0 = np.uint64((int(seed) & 0xFFFFFFFF) + site * 0x632BE5AB) & M32
    s1 = np.uint64((int(seed) >> 32) ^ ((site * 0x7F4A7C15) & 0xFFFFFFFF)) & M32
    thr16 = np.uint64(int(np.float32(p) * np.float32(16777216.0)) >> 8)
    idx = np.arange(rows * cols, dtype=np.uint64)
    pair = idx >> np.uint64(1)
    x = ((pair & M32) * np.uint64(0x9E3779B1) & M32) ^ s0
    x ^= (((pair >> np.uint64(32)) * np.uint64(0x85EBCA77)) + s1) & M32
    x ^= x >> np.uint64(16)
    x = x * np.uint64(0x7FEB352D) & M32
    x ^= x >> np.uint64(15)
    x = x * np.uint64(0x846CA68B) & M32
    x ^= x >> np.uint64(16)
    half = np.where((idx & np.uint64(1)) == 0, x & np.uint64(0xFFFF), x >> np.uint64(16))
    return (half >= thr16).reshape(rows, cols)


def _reference(P, o, x, p, seed, buf):
    """fp32 PyTorch composition with the kernel's bf16 weights, its dropout masks and its ReLU patterns (a bf16 pipeline and an
    fp32 one disagree on the sign of a few pre-activations near zero; with ~0.3 % of the units flipped the gradients differ by
    sqrt(0.003) ~ 5 % in norm, which says nothing about the kernel: the comparison therefore fixes the active sets to the ones
    the forward kernel saved -- non-zero entries of its transposed activation copies)."""
    dev = o.device
    W = {k: (P[k].detach().to(torch.bfloat16).float() if P[k].dim() == 2 else P[k].detach()).requires_grad_(True) for k in P}
    M = o.shape[0]
    inv = float(np.float32(1.0) / (np.float32(1.0) - np.float32(p)))
    k1, k2, k3 = (torch.from_numpy(_keep(seed, s, M, c, p)).to(dev).float() * inv for s, c in ((1, 256), (2, 1024), (3, 256)))
    act = {k: (buf.unpacked(k)[:, :M].t() != 0).float() for k in ("uT", "a1T", "a2T", "c1T", "c2T")}
    assert bool(((act["uT"] == 0) | (k2 != 0)).all())  # every active hidden unit is one the hash keeps
    a = F.linear(o, W["wo"], W["bo"])
    x_mid = x + a * k1
    h2 = F.layer_norm(x_mid, (256,), W["ln_g"], W["ln_b"], 1e-5)
    u = F.linear(h2, W["w1"], W["b1"]) * act["uT"] * inv
    feats = x_mid + F.linear(u, W["w2"], W["b2"]) * k3
    a2 = F.linear(F.linear(feats, W["a1"], W["ab1"]) * act["a1T"], W["a2"], W["ab2"]) * act["a2T"]
    c2 = F.linear(F.linear(feats, W["c1"], W["cb1"]) * act["c1T"], W["c2"], W["cb2"]) * act["c2T"]
    return F.linear(a2, W["a3"]), F.linear(c2, W["c3"]), W, x_mid


def _rel(a, b):
    return float((a.float() - b.float()).norm() / b.float().norm().clamp_min(1e-12))


@pytest.mark.parametrize("M,p", [(80, 0.0), (2048, 0.0), (200, 0.1), (2048, 0.1)])
def test_cls_tail_node_matches_torch(dev, M, p):
    """Forward values, input gradients and every parameter gradient (autograd path without a GradSink) of the fused node vs
    the fp32 composition with identical dropout masks; M = 80 / 200 exercise the partial last workgroup."""
    torch.manual_seed(M)
    P = _params(dev, seed=M)
    o = torch.randn(M, 1, 256, device=dev).to(torch.bfloat16).requires_grad_(True)
    xs = torch.randn(M, 17, 256, device=dev)  # the CLS rows are read in place with row stride 17 * 256
    x = xs[:, :1].detach().requires_grad_(True)
    plan = _plan(P, p)
    import src.ppo.hip_ops as ho

    seeds = []
    orig = ho._seed_pair
    ho._seed_pair = lambda t, pd: seeds.append(orig(t, pd)) or seeds[-1]
    try:
        logits, values = _ClsTailHeads.apply(o, x, plan, *[P[k] for k in TAIL_PARAM_ORDER])
    finally:
        ho._seed_pair = orig
    seed = seeds[0][0]
    gl, gv = torch.randn(M, 4, device=dev), torch.randn(M, 1, device=dev)
    torch.autograd.backward([logits, values], [gl, gv])
    o_r = o.detach().float().reshape(M, 256).requires_grad_(True)
    x_r = x.detach().reshape(M, 256).clone().requires_grad_(True)
    buf = plan.cache[(M, str(o.device))]
    lr, vr, W, x_mid_r = _reference(P, o_r, x_r, p, seed, buf)
    torch.autograd.backward([lr, vr], [gl, gv])
    err = {"logits": _rel(logits.detach(), lr.detach()), "values": _rel(values.detach(), vr.detach()),
           "x_mid": _rel(buf.saved["x_mid"], x_mid_r.detach()), "d_o": _rel(o.grad.reshape(M, 256), o_r.grad),
           "d_x": _rel(x.grad.reshape(M, 256), x_r.grad)}
    for k in TAIL_PARAM_ORDER:
        assert P[k].grad is not None and P[k].grad.shape == P[k].shape, k
        err["d_" + k] = _rel(P[k].grad, W[k].grad)
    assert not buf.busy
    assert all(v < 2e-2 for v in err.values()), {k: round(v, 4) for k, v in err.items()}
    # columns past M of the transposed operands are zero (the weight-gradient kernel sums over all ld columns)
    for name in buf.rows:
        assert not bool(buf.unpacked(name)[:, M:].any()), name


def test_dweight_t_matches_matmul(dev):
    """g2048_dweight_t: dW = dY^T X and the bias row sums from fragment-packed transposed operands, all slices summed, vs torch in
    f32; a 32-row dY^T (one row tile per block) and the pack / unpack helpers round-trip."""
    torch.manual_seed(3)
    ld, m, slices = 256, 256, 8
    jobs, want = [], []
    for N, K, bias in ((64, 64, True), (32, 128, False), (128, 192, True)):
        dyT = torch.randn(N, ld, device=dev).to(torch.bfloat16)
        xT = torch.randn(K, ld, device=dev).to(torch.bfloat16)
        assert torch.equal(nv.unpack_fragments(nv.pack_fragments(dyT), N, ld), dyT)
        dw = torch.full((slices, N, K), float("nan"), device=dev)
        db = torch.full((slices, N), float("nan"), device=dev) if bias else None
        jobs.append((nv.pack_fragments(dyT), nv.pack_fragments(xT), dw, db))
        want.append((dyT.float() @ xT.float().t(), dyT.float().sum(1)))
    nv.dweight_t(jobs, ld, m, slices)
    for (dyT, xT, dw, db), (w_ref, b_ref) in zip(jobs, want):
        torch.testing.assert_close(dw.sum(0), w_ref, rtol=1e-4, atol=1e-3)
        if db is not None:
            torch.testing.assert_close(db.sum(0), b_ref, rtol=1e-4, atol=1e-3)
    dw2 = torch.full((2, 64, 64), float("nan"), device=dev)
    nv.dweight_t([(jobs[0][0], jobs[0][1], dw2, None)], ld, 96, 2)  # a shorter row range, two slices of 3 k-steps: the remainder loop
    torch.testing.assert_close(dw2.sum(0), nv.unpack_fragments(jobs[0][0], 64, ld)[:, :96].float()
                               @ nv.unpack_fragments(jobs[0][1], 64, ld)[:, :96].float().t(), rtol=1e-4, atol=1e-3)
    with pytest.raises(nv.NativeError):
        nv.dweight_t(jobs, ld, 250, slices)  # rows not a multiple of 16 * slices


def test_agent_fused_tail_equals_unfused_path(dev, monkeypatch):
    """PPOAgent on the update path (bf16 autocast, gradients, packed boards) with and without the fused tail: same logits / values
    and the same parameter gradients up to bf16 rounding (dropout off: eval mode), through the GradSink as the trainer runs it."""
    torch.manual_seed(21)
    agent = PPOAgent(hidden_dim=512, d_model=256, nhead=8, num_layers=2, dim_feedforward=1024, dropout=0.1, reduction="cls").to(dev)
    agent.eval()
    boards = torch.randint(0, 12, (96, 16), dtype=torch.uint8, device=dev)
    gl, gv = torch.randn(96, 4, device=dev), torch.randn(96, 1, device=dev)

    def run(fused: bool):
        a = copy.deepcopy(agent)
        if not fused:
            monkeypatch.setattr(type(a), "_tail_heads_ok", lambda self: False)
        params = list(a.parameters())
        flat = [torch.zeros_like(p) for p in params]
        sink = GradSink({id(p): f for p, f in zip(params, flat)})
        with torch.autocast("cuda", dtype=torch.bfloat16):
            logits, values = a(boards, None)
        with grad_sink(sink):
            torch.autograd.backward([logits.float(), values.float()], [gl, gv])
        grads = {}
        for (n, p), f in zip(a.named_parameters(), flat):
            grads[n] = f if id(p) in sink.written else p.grad
        monkeypatch.undo()
        return logits.float(), values.float(), grads, a

    l1, v1, g1, a1 = run(True)
    l0, v0, g0, _ = run(False)
    assert hasattr(a1, "_tail_buffers") and len(a1._tail_buffers) == 1  # the fused node really ran
    # (the critic's outputs of a freshly initialised agent are ~1e-3: absolute, not relative, bf16 noise)
    assert _rel(l1, l0) < 2e-2 and float((v1 - v0).abs().max()) < 2e-3
    for n in g0:
        assert g1[n] is not None and g0[n] is not None, n
        # two bf16 pipelines with different rounding points flip a few ReLU / dropout-free units near zero: per-tensor agreement to
        # a few per cent is what identical arithmetic in different association gives (the node-level test above is the exact one)
        assert _rel(g1[n], g0[n]) < 8e-2, (n, _rel(g1[n], g0[n]))
    assert len(copy.deepcopy(a1)._tail_buffers) == 0  # buffers never travel with a copy of the agent
