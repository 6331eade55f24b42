"""g2048_opt_step (clip + AdamW + GradScaler in two launches) against the PyTorch calls of the reference's update loop
(src/ppo/ppo_trainer.py:413-434): scaler.unscale_, clip_grad_norm_, scaler.step(AdamW), scaler.update."""
import copy

import numpy as np
import pytest
import torch

from src.g2048 import native as nv
import torch.nn as nn
from torch.amp import GradScaler

from src.optim import configure_bert_optimizers
from src.optim.flat_step import FlatAdamWStep
from src.ppo import PPOAgent, PPOTrainer, RolloutBuffer
from src.runs import BatchRunner

pytestmark = pytest.mark.gpu
OPTIM = dict(opt_name="adamw", max_lr=4e-4, betas=(0.9, 0.999), eps=1e-6, weight_decay=0.01, warmup_steps_ratio=0.025,
             scheduler_names=["linear", "cosine"], blacklist_weight_modules=["norm", "embedding"])


class _Net(nn.Module):
    """Shapes that exercise the chunk table: > 1 chunk, ragged tails (numel % 4 != 0), a scalar, both decay groups."""

    def __init__(self):
        super().__init__()
        self.a = nn.Linear(300, 41)       # 12 300 weights: 7 chunks with a ragged tail, bias 41
        self.norm = nn.LayerNorm(41)      # no-decay group
        self.b = nn.Linear(41, 1)         # bias of one element
        self.embedding = nn.Embedding(7, 3)
        self.big = nn.Parameter(torch.randn(5000, 3))


def _pair(dev, with_scaler):
    torch.manual_seed(3)
    net_t = _Net().to(dev)
    net_f = copy.deepcopy(net_t)
    mk = lambda net: configure_bert_optimizers(net, steps=50, **OPTIM)
    ot, of = mk(net_t), mk(net_f)
    st = GradScaler(init_scale=1024.0, growth_interval=3) if with_scaler else None
    sf = GradScaler(init_scale=1024.0, growth_interval=3) if with_scaler else None
    flat = FlatAdamWStep(of["optimizer"], dev)
    return net_t, net_f, ot, of, st, sf, flat


@pytest.mark.parametrize("with_scaler", [True, False])
def test_flat_step_matches_torch_sequence(dev, with_scaler):
    net_t, net_f, ot, of, st, sf, flat = _pair(dev, with_scaler)
    opt_t, opt_f = ot["optimizer"], of["optimizer"]
    sch_t, sch_f = ot["lr_scheduler"]["scheduler"], of["lr_scheduler"]["scheduler"]
    max_norm = 0.5
    g = torch.Generator(device="cpu").manual_seed(11)
    for it in range(10):
        scale = float(st.get_scale()) if with_scaler else 1.0
        big = 30.0 if it % 2 else 0.01  # norms on both sides of the clip threshold
        grads = [torch.randn(p.shape, generator=g).to(dev) * big for p in net_t.parameters()]
        if with_scaler and it in (4, 5):  # two consecutive overflow steps: skipped, scale halves twice
            grads[2][0] = float("inf") if it == 4 else float("nan")
        by_param = {id(q): gr for q, gr in zip(net_f.parameters(), grads)}
        for p, gr in zip(net_t.parameters(), grads):
            p.grad = gr * scale
        for v, q in zip(flat.grad_views, flat.params):  # the flat layout is ordered by parameter group
            v.copy_(by_param[id(q)] * scale)
        if with_scaler:
            st.scale(torch.zeros(1, device=dev))  # lazy init of the scale, as scaler.scale(loss) does in the loop
            st.unscale_(opt_t)
            torch.nn.utils.clip_grad_norm_(net_t.parameters(), max_norm)
            st.step(opt_t)
            st.update()
        else:
            torch.nn.utils.clip_grad_norm_(net_t.parameters(), max_norm)
            opt_t.step()
        flat.step(max_norm, sf)
        sch_t.step()
        sch_f.step()
        if with_scaler:
            assert float(st.get_scale()) == float(sf.get_scale()), it
            assert int(st._growth_tracker.item()) == int(sf._growth_tracker.item()), it
            assert bool(flat.info[1].item()) == (it in (4, 5))
        for (n, p), q in zip(net_t.named_parameters(), net_f.parameters()):
            assert torch.allclose(p, q, rtol=2e-6, atol=1e-8), (it, n, (p - q).abs().max().item())
    # moments and step counts in the optimiser's own state (what a checkpoint stores)
    for p, q in zip(net_t.parameters(), net_f.parameters()):
        a, b = opt_t.state[p], opt_f.state[q]
        assert float(a["step"]) == float(b["step"]) == (8 if with_scaler else 10)
        # (sums with cancellation: the absolute error scales with the largest term, not with the result)
        assert torch.allclose(a["exp_avg"], b["exp_avg"], rtol=2e-6, atol=1e-6 * float(a["exp_avg"].abs().max()))
        assert torch.allclose(a["exp_avg_sq"], b["exp_avg_sq"], rtol=2e-6, atol=1e-6 * float(a["exp_avg_sq"].abs().max()))
    if with_scaler:
        assert float(sf.get_scale()) == 1024.0 * 2 / 4 * 2  # grew after 3 clean steps, halved twice, grew again


def test_flat_step_bumps_versions_so_cached_weights_refresh(dev, tmp_path, monkeypatch):
    """The kernel writes parameters through raw pointers; everything that caches derived weights (bf16 shadows of the update
    path, the packed weights of the fused rollout encoder) keys on autograd's version counters, so the step must advance
    them: a TorchActionFunction kept across an optimiser step has to act with the NEW weights."""
    from src.ppo import TorchActionFunction

    monkeypatch.chdir(tmp_path)
    torch.manual_seed(9)
    agent = PPOAgent(hidden_dim=512, d_model=256, nhead=8, num_layers=2, dim_feedforward=1024, dropout=0.0, reduction="cls")
    tr = _trainer(dev, agent, tmp_path / "v", use_hip_graph=False, rollout_amp=True)
    assert tr._flat_step is not None
    act = TorchActionFunction(agent, use_mask=True, device=dev, amp_dtype=torch.bfloat16)
    assert act._fused is not None
    boards = torch.randint(0, 6, (64, 16), dtype=torch.uint8, device=dev)
    masks = torch.full((64,), 15, dtype=torch.uint8, device=dev)
    before, _ = act.policy_fn(boards, masks)
    v0 = [p._version for p in agent.parameters()]
    tr.collect_rollouts(batch_size=64, num_batches=1)
    tr.update_policy(batch_size=256, n_epochs=1)
    assert all(p._version > a for p, a in zip(agent.parameters(), v0))
    after, _ = act.policy_fn(boards, masks)      # the SAME act_fn object: must have re-packed
    fresh, _ = TorchActionFunction(agent, use_mask=True, device=dev, amp_dtype=torch.bfloat16).policy_fn(boards, masks)
    assert torch.equal(after, fresh) and not torch.equal(after, before)


def test_optimiser_keeps_bf16_shadows_current(dev, tmp_path, monkeypatch):
    """With the flat step the bf16 shadows of the update path (dense and transposed) are rewritten by the optimiser kernel and
    no longer by cast kernels inside the captured graph: after several graph-replayed minibatches every shadow equals the
    bf16 rounding of its parameter bit for bit, and so after an eager step."""
    from src.ppo.hip_ops import Bf16Shadow

    monkeypatch.chdir(tmp_path)
    torch.manual_seed(11)
    agent = PPOAgent(hidden_dim=512, d_model=256, nhead=8, num_layers=2, dim_feedforward=1024, dropout=0.1, reduction="cls")
    tr = _trainer(dev, agent, tmp_path / "s", rollout_amp=True, max_samples_per_epoch=6 * 1024)
    assert tr._flat_step is not None and tr.use_hip_graph
    tr.collect_rollouts(batch_size=128, num_batches=1)
    m = tr.update_policy(batch_size=1024, n_epochs=1)
    assert m["hip_graph"] and m["n_updates"] >= 3
    mine = {id(p) for p in agent.parameters()}
    shadows = [s for s in Bf16Shadow._live if s.views is not None and all(id(p) in mine for p in s.params)]
    assert len(shadows) >= 2 and all(s.maintainer is tr._flat_step for s in shadows)  # encoder layers + heads
    assert sum(len(s.packed) for s in shadows) == 7  # the last layer's out_proj / linear1 / linear2 + four head matrices
    def check():
        for s in shadows:
            assert s.key == s.current_key()
            for i, (p, v) in enumerate(zip(s.params, s.views)):
                assert torch.equal(v, p.detach().to(torch.bfloat16)), i
            for i, tv in s.tviews.items():
                assert torch.equal(tv, s.params[i].detach().to(torch.bfloat16).t()), i
            for i in s.packed:  # fragment-packed copies of the tensor and of its transpose (the fused CLS tail's operands)
                ref = s.params[i].detach().to(torch.bfloat16)
                assert torch.equal(s.pviews[i], nv.pack_fragments(ref)) and torch.equal(s.ptviews[i], nv.pack_fragments(ref.t())), i
    check()
    before = [p.detach().clone() for p in agent.parameters()]
    tr.use_hip_graph = False  # an eager minibatch takes the same optimiser path
    tr.update_policy(batch_size=1024, n_epochs=1)
    assert any(not torch.equal(a, b) for a, b in zip(before, agent.parameters()))
    check()
    # a parameter changed behind the optimiser's back (load_state_dict) makes the shadow copy again on its next use
    with torch.no_grad():
        agent.transformer.encoder.layers[0].linear1.weight.mul_(1.5)
    enc = agent.transformer._shadow
    assert enc.key != enc.current_key()
    enc()
    check()


def test_flat_step_state_dict_round_trip(dev):
    """optimizer.state_dict() of the flat step loads into a plain AdamW and back (checkpoint interchange)."""
    net_t, net_f, ot, of, st, sf, flat = _pair(dev, False)
    for v in flat.grad_views:
        v.normal_()
    flat.step(0.5, None)
    flat.step(0.5, None)
    sd = copy.deepcopy(of["optimizer"].state_dict())
    ot["optimizer"].load_state_dict(sd)  # plain torch optimiser takes it
    for p, q in zip(net_t.parameters(), net_f.parameters()):
        assert torch.equal(ot["optimizer"].state[p]["exp_avg"], of["optimizer"].state[q]["exp_avg"])
        assert float(ot["optimizer"].state[p]["step"]) == 2.0
    # and back: a freshly built flat step adopts a loaded state
    net_n = copy.deepcopy(net_f)
    on = configure_bert_optimizers(net_n, steps=50, **OPTIM)
    flat_n = FlatAdamWStep(on["optimizer"], dev)
    on["optimizer"].load_state_dict(sd)
    flat_n.adopt_state()
    assert torch.equal(flat_n.exp_avg, flat.exp_avg) and torch.equal(flat_n.exp_avg_sq, flat.exp_avg_sq)
    assert torch.equal(flat_n.steps, flat.steps) and float(flat_n.steps[0]) == 2.0
    for v, w in zip(flat.grad_views, flat_n.grad_views):
        w.copy_(v)
    flat.step(0.5, None)
    flat_n.step(0.5, None)
    for p, q in zip(net_f.parameters(), net_n.parameters()):
        assert torch.equal(p, q)


def _trainer(dev, agent, log_dir, **kw):
    optim = dict(OPTIM, scheduler_names=["constant", "constant"])
    args = dict(gamma=0.99, lambda_gae=0.95, clip_epsilon=0.2, value_loss_coef=0.5, entropy_coef=0.01, max_grad_norm=0.5,
                target_kl=10.0, use_action_mask=True, device=dev, mixed_precision="bfloat16", max_samples_per_epoch=1024,
                shuffle_on_reset=False, log_dir=str(log_dir))
    args.update(kw)
    return PPOTrainer(agent, BatchRunner(init_seed=0), RolloutBuffer(31, 16, 4), optim, max_steps=1000, **args)


def test_trainer_update_with_flat_step_matches_torch_step(dev, tmp_path, monkeypatch):
    """The same rollouts and minibatches through update_policy with g2048_opt_step and with PyTorch's calls
    (G2048_FLAT_OPT=0): parameters agree to f32 rounding after several optimiser steps (dropout off, eager mode)."""
    monkeypatch.chdir(tmp_path)

    def run(flat, batch, epochs):
        monkeypatch.setenv("G2048_FLAT_OPT", "1" if flat else "0")
        torch.manual_seed(5)
        agent = PPOAgent(hidden_dim=64, d_model=64, nhead=4, num_layers=2, dim_feedforward=128, dropout=0.0, reduction="cls")
        tr = _trainer(dev, agent, tmp_path / ("f" if flat else "t"), use_hip_graph=False)
        assert (tr._flat_step is not None) == flat
        tr.collect_rollouts(batch_size=64, num_batches=1)
        torch.manual_seed(6)
        m = tr.update_policy(batch_size=batch, n_epochs=epochs)
        return tr, m

    torch.manual_seed(5)
    init = PPOAgent(hidden_dim=64, d_model=64, nhead=4, num_layers=2, dim_feedforward=128, dropout=0.0, reduction="cls")
    p0 = torch.cat([p.detach().flatten() for p in init.parameters()]).to(dev)
    disp = lambda tr: torch.cat([p.detach().flatten() for p in tr.agent.parameters()]) - p0
    # ONE optimiser step (the first AdamW step moves every element by ~lr * g / (|g| + eps): elements with |g| ~ eps feel the
    # different summation orders of the two gradient paths -- GradSink vs at::sum -- so "equal" means 2 %, not 1e-6; the
    # arithmetic itself is pinned by test_flat_step_matches_torch_sequence on identical gradients)
    (tr_f, m_f), (tr_t, m_t) = run(True, 1024, 1), run(False, 1024, 1)
    assert m_f["n_updates"] == m_t["n_updates"] == 1
    assert (disp(tr_f) - disp(tr_t)).norm() / disp(tr_t).norm() < 2e-2
    # eight dependent steps: AdamW's m / sqrt(v) amplifies last-bit differences of the bf16 forward (an element whose
    # gradient is near zero moves by +-lr either way), so the displacements agree as vectors, not element by element.
    # (Both paths must run every forward on the CURRENT weights: before round 2 the eager path kept using the bf16 shadows
    # of the first step, because torch's fused AdamW does not advance the parameters' version counters.)
    tr_f, m_f = run(True, 256, 2)
    tr_t, m_t = run(False, 256, 2)
    assert m_f["n_updates"] == m_t["n_updates"] >= 4
    df, dt = disp(tr_f), disp(tr_t)
    assert df.norm() > 0 and (df - dt).norm() / dt.norm() < 0.25, ((df - dt).norm() / dt.norm()).item()
    np.testing.assert_allclose(m_f["total_loss"], m_t["total_loss"], rtol=2e-2, atol=1e-3)
    # checkpoint written with the flat step loads into the PyTorch-step trainer and the other way round
    tr_f.save_checkpoint(str(tmp_path / "f.pt"))
    tr_t.load_checkpoint(str(tmp_path / "f.pt"), load_optimizer=True)
    p0 = next(iter(tr_t.agent.parameters()))
    assert float(tr_t.optimizer.state[p0]["step"]) == m_f["n_updates"]
    tr_t.save_checkpoint(str(tmp_path / "t.pt"))
    tr_f.load_checkpoint(str(tmp_path / "t.pt"), load_optimizer=True)
    assert float(tr_f._flat_step.steps[0]) == m_f["n_updates"]
    m2 = tr_f.update_policy(batch_size=256, n_epochs=1)  # and keeps training
    assert m2["n_updates"] >= 2 and np.isfinite(m2["total_loss"])


def test_reduce_jobs_kernel(dev):
    """g2048_reduce_jobs: many jobs of mixed dtype / parts / widths in one call, vs f64 sums; ragged widths, strided
    sources (column blocks of a wider partial matrix), > 64 jobs (two launches), bit-reproducible."""
    from src.g2048 import native as nv

    torch.manual_seed(31)
    jobs, want, outs = [], [], []
    shapes = [(16, 768 * 256, torch.bfloat16), (16, 256 * 256, torch.bfloat16), (1, 4 * 512, torch.bfloat16), (1, 513, torch.bfloat16),
              (544, 256, torch.float32), (64, 1024, torch.float32), (3, 1, torch.float32), (7, 6, torch.bfloat16), (1, 1, torch.float32)]
    for rep in range(8):  # 72 + 24 jobs
        for parts, n, dt in shapes:
            src = torch.randn(parts, n, device=dev).to(dt)
            dst = torch.full((n,), float("nan"), device=dev)
            jobs.append((src, dst, n, n, parts))
            want.append(src.double().sum(0))
            outs.append(dst)
        wide = torch.randn(100, 768, device=dev)  # three column blocks of one partial matrix (the add+LN backward's layout)
        for c in range(3):
            dst = torch.empty(256, device=dev)
            jobs.append((wide[:, 256 * c:], dst, 768, 256, 100))
            want.append(wide[:, 256 * c:256 * (c + 1)].double().sum(0))
            outs.append(dst)
    # transposed stores (transpose_rows): few parts and many parts, f32 and bf16 sources, a column block of a wider matrix
    for parts, R, cols, dt, stride in ((256, 31, 256, torch.float32, 32 * 256), (3, 4, 8, torch.bfloat16, 32), (40, 5, 7, torch.float32, 35)):
        src = torch.randn(parts, stride, device=dev).to(dt)
        dst = torch.full((R * cols,), float("nan"), device=dev)
        jobs.append((src, dst, stride, R * cols, parts, R))
        want.append(src[:, :R * cols].double().sum(0).view(R, cols).t().reshape(-1))
        outs.append(dst)
    nv.reduce_jobs(jobs)
    for (src, dst, stride, n, parts, *_), w, o in zip(jobs, want, outs):
        tol = 1e-6 * float(src.float().abs().sum(0).max()) + 1e-7
        assert torch.allclose(o.double(), w, rtol=1e-6, atol=tol), (parts, n, src.dtype, (o.double() - w).abs().max().item())
    first = [o.clone() for o in outs]
    nv.reduce_jobs(jobs)
    assert all(torch.equal(a, b) for a, b in zip(first, outs))
    nv.reduce_jobs([])


def test_grad_sink_equals_per_parameter_reductions(dev, tmp_path, monkeypatch):
    """One backward at the default model shape with the GradSink (one reduction launch) and without (at::sum /
    k_colsum_final per parameter): every gradient agrees to f32 summation-order accuracy, and every parameter gets one."""
    monkeypatch.chdir(tmp_path)

    def grads(sink_on):
        monkeypatch.setenv("G2048_GRAD_SINK", "1" if sink_on else "0")
        torch.manual_seed(7)
        agent = PPOAgent(hidden_dim=512, d_model=256, nhead=8, num_layers=2, dim_feedforward=1024, dropout=0.0, reduction="cls")
        tr = _trainer(dev, agent, tmp_path / ("s" if sink_on else "n"), use_hip_graph=False, rollout_amp=True)
        assert tr.grad_sink == sink_on and tr._flat_grad is not None
        tr.collect_rollouts(batch_size=64, num_batches=1)
        data = tr.rollout_buffer.device_data(dev)
        M = 2048
        assert data["boards"].shape[0] >= M
        sample = dict(obs=data["boards"][:M].contiguous(), actions=data["actions"][:M].contiguous(),
                      masks=data["masks"][:M].contiguous(), old_lp=data["log_probs"][:M].contiguous(),
                      adv=data["raw_advantages"][:M].clamp(-3, 3).contiguous(), ret=data["raw_returns"][:M].clamp(-3, 3).contiguous())
        agent.train()
        tr._loss_backward(**sample)
        assert all(p.grad is not None for p in agent.parameters())
        tr._allreduce_grads()  # gathers the gradients autograd produced into the bucket (world 1: no collective)
        return {n: p.grad.detach().clone() for n, p in agent.named_parameters()}, tr

    g_s, tr_s = grads(True)
    g_n, _ = grads(False)
    assert set(g_s) == set(g_n)
    for n in g_s:
        a, b = g_s[n], g_n[n]
        assert torch.isfinite(a).all(), n
        err = (a - b).norm() / b.norm().clamp_min(1e-20)
        # bf16 rounding either way: 16 split-K slices rounded separately and summed in f32 (sink) vs. one bf16 GEMM output
        # (the 2048-row GEMMs of the reference path) -- both within bf16 accuracy of the exact gradient
        assert err < 6e-3, (n, err.item())
    # and the bucket holds exactly these values (what the optimiser kernel reads)
    for p, v in zip(tr_s._params, tr_s._flat_views):
        assert p.grad.data_ptr() == v.data_ptr()


def test_flat_step_leaves_parameters_without_gradient_alone(dev):
    """A parameter that received no gradient (``skip``) is not decayed, its moments stay untouched and it does not enter the
    gradient norm -- what torch.optim.AdamW does with ``p.grad is None`` -- while the others step exactly as without it."""
    torch.manual_seed(2)
    ps = [nn.Parameter(torch.randn(300, 256, device=dev)), nn.Parameter(torch.randn(77, device=dev)),
          nn.Parameter(torch.randn(64, 32, device=dev))]
    ref = [nn.Parameter(p.detach().clone()) for p in ps]
    kw = dict(lr=1e-2, betas=(0.9, 0.999), eps=1e-6, weight_decay=0.1)
    opt, opt_ref = torch.optim.AdamW(ps, **kw), torch.optim.AdamW(ref, **kw)
    fs = FlatAdamWStep(opt, dev)
    for step in range(3):
        g0, g2 = torch.randn_like(ps[0]), torch.randn_like(ps[2])
        fs.grad_views[0].copy_(g0)
        fs.grad_views[1].fill_(float("nan"))  # never read: the parameter is skipped
        fs.grad_views[2].copy_(g2)
        ref[0].grad, ref[1].grad, ref[2].grad = g0.clone(), None, g2.clone()
        torch.nn.utils.clip_grad_norm_([ref[0], ref[2]], 0.5)
        opt_ref.step()
        fs.step(0.5, None, skip=(1,))
        assert float(fs.info[1]) == 0.0  # the NaN slice did not reach the norm
    torch.testing.assert_close(ps[0].detach(), ref[0].detach(), rtol=2e-6, atol=2e-6)
    torch.testing.assert_close(ps[2].detach(), ref[2].detach(), rtol=2e-6, atol=2e-6)
    assert torch.equal(ps[1].detach(), ref[1].detach()) and not bool(fs.m_views[1].any()) and not bool(fs.v_views[1].any())


def test_graph_replayed_update_leaves_an_unused_parameter_alone(dev, tmp_path, monkeypatch):
    """An agent with a parameter that no forward uses, through ``update_policy`` with the hipGraph on and off: the parameter and
    its moments must stay untouched either way (torch.optim.AdamW skips ``p.grad is None``).  After a replay every ``p.grad``
    points at a bucket slice, so the list of grad-less parameters has to come from the capture (``_GraphedFwdBwd.no_grad``);
    recomputed from ``p.grad`` it was empty and the parameter was decayed by the graph path only."""
    monkeypatch.chdir(tmp_path)

    def run(graph):
        torch.manual_seed(5)
        agent = PPOAgent(hidden_dim=64, d_model=64, nhead=4, num_layers=2, dim_feedforward=128, dropout=0.0, reduction="cls")
        agent.unused = nn.Parameter(torch.full((37, 5), 3.0))  # falls into the decay group: weight decay would shrink it
        tr = _trainer(dev, agent, tmp_path / ("g" if graph else "e"), use_hip_graph=graph, rollout_amp=True)
        assert tr._flat_step is not None
        tr.collect_rollouts(batch_size=64, num_batches=1)
        torch.manual_seed(6)
        m = tr.update_policy(batch_size=256, n_epochs=2)
        assert m["n_updates"] >= 4 and m["hip_graph"] == graph, m
        return tr

    for graph in (False, True):
        tr = run(graph)
        i = [k for k, p in enumerate(tr._params) if p is tr.agent.unused][0]
        assert tr._no_grad == [i]
        if graph:
            assert all(g.no_grad == [i] for g in tr._graphs.values())
        assert torch.equal(tr.agent.unused.detach(), torch.full((37, 5), 3.0, device=dev)), graph
        fs = tr._flat_step
        assert not bool(fs.m_views[i].any()) and not bool(fs.v_views[i].any())
        # and the used parameters did move
        assert float((tr.agent.actor[0].weight.detach() - 0).abs().sum()) > 0 and float(fs.m_views[0].abs().sum()) > 0
