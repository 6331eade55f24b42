"""The MLP policy's update path (BASELINE.json configs[1]) through csrc/g2048_mlp.hip: every kernel against its PyTorch composition,
and the whole policy - logits, values and ALL parameter gradients at minibatch 2048 - against an fp32 PyTorch forward / backward of the
same weights (the reference has no MLP policy; its heads are src/ppo/ppo_agent.py:72-87 of the reference)."""
import copy

import pytest
import torch

from src.g2048 import native as nv
from src.ppo.ppo_agent import MLPAgent

pytestmark = pytest.mark.gpu
bf = torch.bfloat16
rel = lambda a, b: ((a.float() - b.float()).norm() / b.float().norm().clamp_min(1e-12)).item()


def test_gemm_jobs_matches_torch(dev):
    """g2048_gemm_jobs: bias + ReLU, the ReLU mask of a saved activation, two K-segments, several jobs per launch writing column halves
    of one buffer, ragged row counts, and the refusals."""
    torch.manual_seed(3)
    for M in (2048, 200, 64):
        x = (torch.randn(M, 512, device=dev) * 0.5).to(bf)
        x2 = (torch.randn(M, 1024, device=dev) * 0.5).to(bf)
        w1 = (torch.randn(512, 512, device=dev) / 512 ** 0.5).to(bf)
        w2 = (torch.randn(512, 512, device=dev) / 512 ** 0.5).to(bf)
        b1, b2 = torch.randn(512, device=dev) * 0.1, torch.randn(512, device=dev) * 0.1
        # forward form: two jobs, one input, outputs = the halves of a [M, 1024] buffer
        y = torch.full((M, 1024), float("nan"), dtype=bf, device=dev)
        nv.gemm_jobs([dict(segs=[(x, w1)], bias=b1, relu=True, y=y[:, :512]), dict(segs=[(x, w2)], bias=b2, relu=True, y=y[:, 512:])], M)
        want = torch.cat([torch.relu(x.float() @ w1.float().t() + b1), torch.relu(x.float() @ w2.float().t() + b2)], 1)
        assert torch.isfinite(y.float()).all() and rel(y, want) < 4e-3, (M, rel(y, want))
        # backward form: two K-segments (column halves of one buffer) into one output, masked by a saved activation
        act = torch.relu(torch.randn(M, 512, device=dev)).to(bf)
        dz = torch.empty(M, 512, dtype=bf, device=dev)
        nv.gemm_jobs([dict(segs=[(x2[:, :512], w1), (x2[:, 512:], w2)], act=act, y=dz)], M)
        want = (x2[:, :512].float() @ w1.float().t() + x2[:, 512:].float() @ w2.float().t()) * (act.float() > 0)
        assert rel(dz, want) < 4e-3 and bool((dz.float()[act.float() == 0] == 0).all())
        # no epilogue at all, K = 64, N = 64
        xs, ws = x[:, :64].contiguous(), w1[:64, :64].contiguous()
        ys = torch.empty(M, 64, dtype=bf, device=dev)
        nv.gemm_jobs([dict(segs=[(xs, ws)], y=ys)], M)
        assert rel(ys, xs.float() @ ws.float().t()) < 4e-3
    with pytest.raises(nv.NativeError):
        nv.gemm_jobs([dict(segs=[(x[:, :48], w1[:, :48])], y=y[:, :512])], M)  # K not a multiple of 64
    with pytest.raises(nv.NativeError):
        nv.gemm_jobs([dict(segs=[(x.float(), w1)], y=y[:, :512])], M)          # f32 input


def test_mlp_embed_and_output_kernels(dev):
    torch.manual_seed(4)
    M = 300
    boards = torch.randint(0, 18, (M, 16), dtype=torch.uint8, device=dev)
    w = (torch.randn(512, 496, device=dev) / 4).to(bf)  # trunk_in.weight
    bias = torch.randn(512, device=dev) * 0.1
    y, oh = torch.empty(M, 512, dtype=bf, device=dev), torch.full((M, 512), 7.0, dtype=bf, device=dev)
    nv.mlp_embed_fwd(boards, w.t().contiguous(), bias, y, oh)
    cols = boards.long() + torch.arange(16, device=dev) * 31
    onehot = torch.zeros(M, 512, device=dev).scatter_(1, cols, 1.0)
    onehot[:, 496] = 1.0  # the column of ones whose row of onehot^T dY is trunk_in's bias gradient
    assert torch.equal(oh.float(), onehot)
    want = torch.relu(onehot[:, :496] @ w.float().t() + bias)
    assert rel(y, want) < 3e-3
    # output layers
    h2 = torch.relu(torch.randn(M, 1024, device=dev)).to(bf)
    w3 = (torch.randn(5, 512, device=dev) / 512 ** 0.5).to(bf)
    logits, values = torch.empty(M, 4, device=dev), torch.empty(M, device=dev)
    nv.mlp_out_fwd(h2, w3, logits, values)
    assert torch.allclose(logits, h2[:, :512].float() @ w3[:4].float().t(), atol=2e-3, rtol=1e-3)
    assert torch.allclose(values, h2[:, 512:].float() @ w3[4].float(), atol=2e-3, rtol=1e-3)
    dl, dv = torch.randn(M, 4, device=dev), torch.randn(M, device=dev)
    dh2 = torch.empty(M, 1024, dtype=bf, device=dev)
    ws = nv.mlp_out_bwd(dl, dv, h2, w3, dh2)
    dlb, dvb = dl.to(bf).float(), dv.to(bf).float()
    want = torch.cat([dlb @ w3[:4].float(), dvb[:, None] * w3[4].float()[None]], 1) * (h2.float() > 0)
    assert rel(dh2, want) < 4e-3
    s = ws.sum(0)
    assert rel(s[:4], dlb.t() @ h2[:, :512].float()) < 1e-4 and rel(s[4], dvb @ h2[:, 512:].float()) < 1e-4


def _fp32_reference(agent, boards, gl, gv):
    ref = copy.deepcopy(agent).float()
    ref._trunk_shadow = ref._head_shadow = ref._mlp_plan = None
    oh = torch.nn.functional.one_hot(boards.long(), 31).float().flatten(1)
    h = torch.relu(ref.trunk_hidden(torch.relu(ref.trunk_in(oh))))
    lo, va = ref.actor(h), ref.critic(h)
    ((lo * gl).sum() + (va.flatten() * gv).sum()).backward()
    return lo.detach(), va.detach().flatten(), {n: p.grad.clone() for n, p in ref.named_parameters()}


@pytest.mark.parametrize("sink", [False, True], ids=["autograd", "sink"])
def test_mlp_policy_update_node_against_fp32(dev, sink, monkeypatch):
    """MLPAgent at minibatch 2048, bf16 autocast: logits, values and every parameter gradient of the ten-launch node against fp32
    PyTorch, with the per-layer nodes of rounds 1-3 (G2048_MLP_FUSED=0, i.e. hipBLASLt + ReLU kernels under the same autocast) as the
    yardstick: per tensor the node's error may be 1.3 x theirs + 1 %.  "sink": the trainer's route - every 512-wide weight gradient
    (trunk_in's through the one-hot matrix, stored transposed) in the grouped g2048_dweight_jobs launch, all second stages in
    g2048_reduce_jobs."""
    from src.ppo.hip_ops import GradSink, grad_sink

    torch.manual_seed(11)
    agent = MLPAgent(hidden_dim=512, trunk_dim=512).to(dev).train()
    g = torch.Generator(device="cpu").manual_seed(5)
    boards = torch.randint(0, 12, (2048, 16), generator=g, dtype=torch.uint8).to(dev)
    gl, gv = torch.randn(2048, 4, generator=g).to(dev), torch.randn(2048, generator=g).to(dev)
    l32, v32, g32 = _fp32_reference(agent, boards, gl, gv)

    def run(fused, use_sink):
        monkeypatch.setenv("G2048_MLP_FUSED", "1" if fused else "0")
        m = copy.deepcopy(agent)
        m._trunk_shadow = m._head_shadow = m._mlp_plan = None
        calls = []
        real = nv.gemm_jobs
        monkeypatch.setattr(nv, "gemm_jobs", lambda *a, **k: (calls.append(1), real(*a, **k))[1])
        with torch.autocast("cuda", dtype=bf):
            lo, va = m(boards, None)
        ps = list(m.parameters())
        bucket = [torch.zeros_like(p) for p in ps]
        s = GradSink({id(p): b for p, b in zip(ps, bucket)}) if use_sink else None
        with grad_sink(s):
            ((lo.float() * gl).sum() + (va.float().flatten() * gv).sum()).backward()
        monkeypatch.setattr(nv, "gemm_jobs", real)
        if use_sink:
            assert len(s.written) == len(ps), "the sink must take every gradient of the MLP policy"
        grads = {n: (b if use_sink else p.grad.detach().float()) for (n, p), b in zip(m.named_parameters(), bucket)}
        return lo.detach().float(), va.detach().float().flatten(), grads, len(calls)

    lo1, va1, g1, n1 = run(True, sink)
    lo0, va0, g0, n0 = run(False, sink)
    assert n1 == 6 and n0 == 0, (n1, n0)  # 3 forward + 3 backward launches of the job-table GEMM
    assert (lo1 - l32).abs().max().item() < 0.05 and (va1 - v32).abs().max().item() < 0.05
    assert rel(lo1, l32) < 1.3 * rel(lo0, l32) + 0.01 and rel(va1, v32) < 1.3 * rel(va0, v32) + 0.01
    bad = {}
    for n in g32:
        assert torch.isfinite(g1[n]).all(), n
        e1, e0 = rel(g1[n], g32[n]), rel(g0[n], g32[n])
        if not e1 < 1.3 * e0 + 0.01:
            bad[n] = (round(e1, 4), round(e0, 4))
    assert not bad, bad
