"""g2048_linear_add_ln_fwd / _bwd (csrc/g2048_rowgemm.hip): a 256-output Linear fused with the add + dropout + LayerNorm kernel behind
it, against the unfused pair it replaces (same dropout hash on the same element index: the masks must be IDENTICAL) and against an
f64 composition of the reference's operators (nn.Linear -> dropout -> residual add -> LayerNorm of
nn.TransformerEncoderLayer(norm_first=True), reference src/ppo/transformer_encoder.py:138-148, and their autograd)."""
import pytest
import torch

from src.g2048 import native as nv

pytestmark = pytest.mark.gpu


def _operands(dev, T, K, seed):
    g = torch.Generator(device="cpu").manual_seed(seed)
    u = (torch.randn(T, K, generator=g) * 0.5).to(torch.bfloat16).to(dev)
    W = (torch.randn(256, K, generator=g) / K ** 0.5).to(torch.bfloat16).to(dev)
    bias = (torch.randn(256, generator=g) * 0.1).to(dev)
    x = torch.randn(T, 256, generator=g).to(dev)
    gamma = (1 + 0.1 * torch.randn(256, generator=g)).to(dev)
    beta = (0.1 * torch.randn(256, generator=g)).to(dev)
    return u, W, bias, x, gamma, beta


@pytest.mark.parametrize("T,K", [(136 * 3 + 5, 256), (1000, 1024), (34816, 256), (34816, 768), (34816, 1024), (41000, 1024)])
def test_linear_add_ln_fwd_equals_the_unfused_pair(dev, T, K):
    u, W, bias, x, gamma, beta = _operands(dev, T, K, 7 + K)
    Wp = nv.pack_fragments(W)
    for p, seed in ((0.0, 0), (0.1, 0x1234567812345)):
        x_new, h = torch.empty_like(x), torch.empty(T, 256, dtype=torch.bfloat16, device=dev)
        mean, rstd = torch.empty(T, device=dev), torch.empty(T, device=dev)
        nv.linear_add_ln_fwd(u, Wp, bias, x.data_ptr(), 256, gamma, beta, x_new, h, mean, rstd, 1e-5, p, seed)
        # the unfused pair: bf16 Linear (f32 accumulation, bias, rounded to bf16), then g2048_add_ln_fwd with the same seed
        a = torch.nn.functional.linear(u.float(), W.float(), bias).to(torch.bfloat16)
        x_new_u, h_u = torch.empty_like(x), torch.empty_like(h)
        mean_u, rstd_u = torch.empty(T, device=dev), torch.empty(T, device=dev)
        nv.add_ln_fwd(x.data_ptr(), 256, a, gamma, beta, x_new_u, h_u, mean_u, rstd_u, T, 1e-5, p, seed)
        d_fused, d_unf = x_new - x, x_new_u - x
        assert torch.equal(d_fused == 0, d_unf == 0) or ((d_fused == 0) != (d_unf == 0)).float().mean() < 1e-4, "dropout masks differ"
        if p > 0:
            kept = (d_fused != 0).float().mean().item()
            assert abs(kept - 0.9) < 0.01, kept
        # values: the two GEMMs sum in different orders, so `a` may differ by one bf16 rounding step (2^-8 relative) on some elements
        tol = 2.0 ** -7 * a.float().abs() / (1 - p) + 1e-6
        assert bool(((x_new - x_new_u).abs() <= tol + 1e-6 * x.abs()).all()), (T, K, p, (x_new - x_new_u).abs().max().item())
        # the LayerNorm on the fused kernel's own x_new, in f64
        xn = x_new.double()
        mu, var = xn.mean(1, keepdim=True), xn.var(1, unbiased=False, keepdim=True)
        h_ref = ((xn - mu) / torch.sqrt(var + 1e-5) * gamma.double() + beta.double())
        assert (h.double() - h_ref).abs().max().item() < 0.03 and torch.allclose(mean.double(), mu.flatten(), atol=1e-5)
        assert torch.allclose(rstd.double(), 1 / torch.sqrt(var.flatten() + 1e-5), rtol=1e-4)
        # and the whole thing against the f64 Linear (p = 0: nothing random in between)
        if p == 0:
            a64 = u.double() @ W.double().t() + bias.double()
            assert ((x_new.double() - (x.double() + a64)).abs() <= 2.0 ** -8 * a64.abs() + 1e-5).all()


def test_linear_add_ln_fwd_strided_residual_and_no_bias(dev):
    """The residual rows as a strided view (the CLS rows [B, 1, 256] of a [B, 17, 256] stream), a leading dimension wider than K, no bias."""
    T, K = 221, 256
    u_wide, W, _, _, gamma, beta = _operands(dev, T, 512, 3)
    u = u_wide[:, 128:128 + K]
    W = W[:, :K].contiguous()
    stream = torch.randn(T, 17, 256, device=dev)
    x = stream[:, 0]
    x_new, h = torch.empty(T, 256, device=dev), torch.empty(T, 256, dtype=torch.bfloat16, device=dev)
    mean, rstd = torch.empty(T, device=dev), torch.empty(T, device=dev)
    nv.linear_add_ln_fwd(u, nv.pack_fragments(W), None, x.data_ptr(), x.stride(0), gamma, beta, x_new, h, mean, rstd, 1e-5, 0.0, 0)
    a64 = u.double() @ W.double().t()
    assert ((x_new.double() - (x.double() + a64)).abs() <= 2.0 ** -8 * a64.abs() + 1e-5).all()
    assert torch.allclose(mean, x_new.mean(1), atol=1e-5)


@pytest.mark.parametrize("T,K,period,with_da", [(136 * 3 + 5, 256, 1, True), (1000, 768, 1, True), (34816, 1024, 1, True),
                                                  (34816, 768, 1, True), (34816, 768, 17, True), (34816, 768, 1, False),
                                                  (41000, 1024, 1, True)])
def test_linear_add_ln_bwd_equals_the_unfused_pair(dev, T, K, period, with_da):
    g = torch.Generator(device="cpu").manual_seed(99 + K + period)
    dy = (torch.randn(T, K, generator=g) * 0.1).to(torch.bfloat16).to(dev)
    Wt = (torch.randn(256, K, generator=g) / K ** 0.5).to(torch.bfloat16).to(dev)  # [256][K] = the consuming Linear's weight, transposed
    xn = torch.randn(T, 256, generator=g).to(dev)
    gamma = (1 + 0.1 * torch.randn(256, generator=g)).to(dev)
    g_x = torch.randn(T // period if period > 1 else T, 256, generator=g).to(dev)
    mean = xn.mean(1).contiguous()
    rstd = (1 / torch.sqrt(xn.var(1, unbiased=False) + 1e-5)).contiguous()
    Wtp = nv.pack_fragments(Wt)
    for p, seed in ((0.0, 0), (0.1, 0x9876543210)):
        dx = torch.empty_like(xn)
        da = torch.empty(T, 256, dtype=torch.bfloat16, device=dev) if with_da else None
        ws = nv.linear_add_ln_bwd(dy, Wtp, xn.data_ptr(), 256, g_x, mean, rstd, gamma, dx, da, p, seed, g_x_period=period)
        g_h = (dy.float() @ Wt.float().t()).to(torch.bfloat16)  # what the unfused input-gradient GEMM hands over
        dx_u = torch.empty_like(xn)
        da_u = torch.empty_like(da) if with_da else None
        ws_u = nv.add_ln_bwd(xn.data_ptr(), 256, g_x, g_h, mean, rstd, gamma, dx_u, da_u, None, T, p, seed, g_x_period=period)
        scale = dx_u.abs().max().item()
        assert (dx - dx_u).abs().max().item() < 0.02 * scale, (T, K, p, (dx - dx_u).abs().max().item(), scale)
        assert ((dx - dx_u).norm() / dx_u.norm()).item() < 4e-3
        if with_da:
            assert ((da.float() == 0) != (da_u.float() == 0)).float().mean().item() < 1e-3, "dropout masks differ"
            assert ((da.float() - da_u.float()).norm() / da_u.float().norm()).item() < 6e-3
        sums, sums_u = ws.sum(0), ws_u.sum(0)
        for k, name in enumerate(("dgamma", "dbeta", "dbias")):
            a, b = sums[256 * k:256 * (k + 1)], sums_u[256 * k:256 * (k + 1)]
            if name == "dbias" and not with_da:
                assert not bool(a.any())
                continue
            assert ((a - b).norm() / b.norm().clamp_min(1e-6)).item() < 1e-2, (name, T, K, p)
        # against f64: dLayerNorm of g_h (as the bf16 tensor the unfused path sees)
        if p == 0 and period == 1:
            gh, x64 = g_h.double(), xn.double()
            xh = (x64 - mean.double()[:, None]) * rstd.double()[:, None]
            dxh = gh * gamma.double()
            ref = g_x.double() + rstd.double()[:, None] * (dxh - dxh.mean(1, keepdim=True) - xh * (dxh * xh).mean(1, keepdim=True))
            assert ((dx.double() - ref).norm() / ref.norm()).item() < 4e-3
    assert 0 < ws.shape[0] <= 256 and ws.shape[1] == 768


def test_linear_add_ln_bwd_strided_packed_weight_and_extra_rows(dev):
    """The CLS-only last layer's case: the weight is K = 512 columns (the K/V rows of in_proj) of a packed [256][768] transpose, and
    every 17th row receives an extra bf16 term (the query projection's share) - against the unfused sequence GEMM, addmm into the
    CLS rows, g2048_add_ln_bwd with the CLS-row residual gradient (period 17)."""
    B, S, D = 2048, 17, 256
    T = B * S
    g = torch.Generator(device="cpu").manual_seed(321)
    dkv = (torch.randn(T, 2 * D, generator=g) * 0.1).to(torch.bfloat16).to(dev)
    w_in = (torch.randn(3 * D, D, generator=g) / D ** 0.5).to(torch.bfloat16).to(dev)  # in_proj weight [768][256]
    extra = (torch.randn(B, D, generator=g) * 0.1).to(torch.bfloat16).to(dev)
    xn = torch.randn(T, D, generator=g).to(dev)
    gamma = (1 + 0.1 * torch.randn(D, generator=g)).to(dev)
    g_x = torch.randn(B, D, generator=g).to(dev)  # CLS rows only
    mean, rstd = xn.mean(1).contiguous(), (1 / torch.sqrt(xn.var(1, unbiased=False) + 1e-5)).contiguous()
    wtp = nv.pack_fragments(w_in.t().contiguous())  # [256][768] packed
    dx, da = torch.empty_like(xn), torch.empty(T, D, dtype=torch.bfloat16, device=dev)
    ws = nv.linear_add_ln_bwd(dkv, wtp[(D // 16) * 512:], xn.data_ptr(), D, g_x, mean, rstd, gamma, dx, da, 0.1, 77, g_x_period=S,
                              tile_stride=(3 * D // 16) * 512, g_h_extra=extra, extra_period=S)
    g_h = (dkv.float() @ w_in[D:].float()).to(torch.bfloat16)
    g_h[::S] = (g_h[::S].float() + extra.float()).to(torch.bfloat16)
    dx_u, da_u = torch.empty_like(xn), torch.empty_like(da)
    ws_u = nv.add_ln_bwd(xn.data_ptr(), D, g_x, g_h, mean, rstd, gamma, dx_u, da_u, None, T, 0.1, 77, g_x_period=S)
    assert ((dx - dx_u).norm() / dx_u.norm()).item() < 4e-3
    assert ((dx[::S] - dx_u[::S]).norm() / dx_u[::S].norm()).item() < 4e-3, "the CLS rows (extra term + residual gradient)"
    assert ((da.float() - da_u.float()).norm() / da_u.float().norm()).item() < 6e-3
    assert ((ws.sum(0) - ws_u.sum(0)).norm() / ws_u.sum(0).norm()).item() < 1e-2


def test_rowgemm_refuses_what_it_cannot_take(dev):
    u = torch.zeros(64, 384, dtype=torch.bfloat16, device=dev)
    assert not nv.rowgemm_ok(u, torch.zeros(256 * 384, dtype=torch.bfloat16, device=dev))       # K not a multiple of 256
    assert not nv.rowgemm_ok(torch.zeros(64, 256, device=dev), torch.zeros(256 * 256, dtype=torch.bfloat16, device=dev))  # f32 input
    assert not nv.rowgemm_ok(torch.zeros(64, 256, dtype=torch.bfloat16, device=dev), torch.zeros(100, dtype=torch.bfloat16, device=dev))
    with pytest.raises(nv.NativeError):
        nv.linear_add_ln_fwd(u, torch.zeros(256 * 384, dtype=torch.bfloat16, device=dev), None, 0, 256, None, None, None, None, None, None,
                             1e-5, 0.0, 0)


def test_agent_update_path_with_and_without_the_fused_launches(dev, monkeypatch):
    """The default-shape agent's forward + backward at minibatch 2048 (34 816 tokens) through the update path with the fused Linear +
    add + LayerNorm launches (default) and without them (G2048_ROWGEMM=0): same outputs and per-tensor gradients up to the summation
    order of the K > 256 GEMMs (dropout 0: nothing random); with dropout the two paths draw the same masks at the fused sites in the
    forward (same seeds, same element indices).  Also checks that the fused path really ran: no [T, 256] input gradient is produced by
    the linked Linears."""
    import copy

    from src.ppo import hip_ops
    from test_host_logic import default_shape_agent

    agent = default_shape_agent(dropout=0.0).to(dev).train()
    g = torch.Generator(device="cpu").manual_seed(5)
    boards = torch.randint(0, 12, (2048, 16), generator=g, dtype=torch.uint8).to(dev)
    gl, gv = torch.randn(2048, 4, generator=g).to(dev), torch.randn(2048, generator=g).to(dev)

    def run(on):
        monkeypatch.setenv("G2048_ROWGEMM", "1" if on else "0")
        m = copy.deepcopy(agent)
        calls = {"fwd": 0, "bwd": 0}
        real_f, real_b = nv.linear_add_ln_fwd, nv.linear_add_ln_bwd

        def cf(*a, **k):
            calls["fwd"] += 1
            return real_f(*a, **k)

        def cb(*a, **k):
            calls["bwd"] += 1
            return real_b(*a, **k)

        monkeypatch.setattr(nv, "linear_add_ln_fwd", cf)
        monkeypatch.setattr(nv, "linear_add_ln_bwd", cb)
        with torch.autocast("cuda", dtype=torch.bfloat16):
            lo, va = m(boards, None)
        ((lo.float() * gl).sum() + (va.float().flatten() * gv).sum()).backward()
        monkeypatch.setattr(nv, "linear_add_ln_fwd", real_f)
        monkeypatch.setattr(nv, "linear_add_ln_bwd", real_b)
        return lo.detach().float(), va.detach().float(), {n: p.grad.detach().float().clone() for n, p in m.named_parameters()}, calls

    lo1, va1, g1, c1 = run(True)
    lo0, va0, g0, c0 = run(False)
    # 3 full layers: out_proj and linear2 forward; in the backward linear1's and in_proj's input gradients
    # (+ the CLS-only last layer's K/V input gradient, fused into the LayerNorm backward of the layer in front of it)
    assert c1 == {"fwd": 6, "bwd": 7} and c0 == {"fwd": 0, "bwd": 0}, (c1, c0)
    assert (lo1 - lo0).abs().max().item() < 0.03 and (va1 - va0).abs().max().item() < 0.03
    # The two paths differ in the summation order of the K > 256 GEMMs only, i.e. by one bf16 rounding step on some elements of a
    # Linear's output - but three ReLU layers downstream a flipped pre-activation sign moves a gradient tensor by a few per cent
    # (measured here: 3.7-6.5 % on the heads' first Linears), exactly as between ANY two bf16 implementations.  What pins the
    # accuracy of the (now default) fused path against fp32 is test_update_path_gradients_per_tensor_at_minibatch_size with PyTorch's
    # autocast backward as the yardstick; this test pins that the two routes compute the same thing.
    rel = lambda a, b: ((a - b).norm() / b.norm().clamp_min(1e-12)).item()
    bad = {n: round(rel(g1[n], g0[n]), 4) for n in g1 if not rel(g1[n], g0[n]) < 0.12}
    assert not bad, bad
    f1, f0 = torch.cat([v.flatten() for v in g1.values()]), torch.cat([g0[n].flatten() for n in g1])
    assert torch.nn.functional.cosine_similarity(f1, f0, dim=0).item() > 0.998
