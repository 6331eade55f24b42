"""Reference-produced vectors through the HIP kernels of the update, at the reference's DEFAULT model shape.

tests/golden/make_golden_torch.py ran the reference's own PPOAgent / PPOTrainer._compute_ppo_loss (d_model 256, 8 heads,
4 layers, ff 1024, "cls") on the weights of tests/golden/weights_recipe.py and stored inputs + outputs (``default/*``);
here the same weights go into this repository's agent on the MI355X and every path that can serve the update or the
rollout must reproduce them: the fp32 path to 1e-5 (north_star), the bf16 HIP paths to the distance bf16 autocast itself
has from fp32.  Also: the hipGraph-replayed update at the shape bench.py runs against a pure-PyTorch fp32 backward.
"""
import copy
import os

import numpy as np
import pytest
import torch

from src.g2048 import native as nv
from src.ppo import PPOAgent, PPOTrainer, RolloutBuffer
from src.ppo.data_loader import DeviceBatches, PPODataset
from src.runs import BatchRunner
from test_host_logic import default_shape_agent

pytestmark = pytest.mark.gpu
REF = np.load(os.path.join(os.path.dirname(__file__), "golden", "torch_reference.npz"))
OPTIM = dict(opt_name="adamw", max_lr=4e-4, betas=(0.9, 0.999), eps=1e-6, weight_decay=0.01, warmup_steps_ratio=0.025,
             scheduler_names=["constant", "constant"], blacklist_weight_modules=["norm", "embedding"])
BITS = [1, 2, 4, 8]


def _t(key, dev):
    return torch.from_numpy(REF[key]).to(dev)


@pytest.mark.parametrize("which", ["default", "small"])
def test_ppo_loss_kernel_on_reference_vectors(dev, which):
    """g2048_ppo_loss (f32 inputs) on the logits / values / actions / masks / old log-probs / advantages / returns of the
    reference's own _compute_ppo_loss run: new log-probs and the five means within 1e-5 of what the reference returned."""
    if which == "default":
        logits, values = _t("default/logits", dev), _t("default/values", dev).reshape(-1)
        actions, bits = _t("default/actions", dev).to(torch.uint8), _t("default/mask_bits", dev)
        old, adv, ret = _t("default/old_logp", dev), _t("default/adv", dev), _t("default/ret", dev)
        want = {k: REF[f"default/{n}"] for k, n in (("pl", "loss_policy"), ("vl", "loss_value"), ("el", "loss_entropy"),
                                                    ("tot", "loss_total"), ("nlp", "new_logp"))}
    else:
        logits, values = _t("agent_mean/logits", dev), _t("agent_mean/values", dev).reshape(-1)
        actions = _t("agent_mean/actions", dev).to(torch.uint8)
        bits = (_t("agent_mean/masks", dev).to(torch.uint8) * torch.tensor(BITS, dtype=torch.uint8, device=dev)).sum(-1).to(torch.uint8)
        old, adv, ret = _t("loss/old_logp", dev), _t("loss/adv", dev), _t("loss/ret", dev)
        want = {k: REF[f"loss/{n}"] for k, n in (("pl", "policy"), ("vl", "value"), ("el", "entropy"), ("tot", "total"),
                                                 ("nlp", "new_logp"))}
    new_lp, sums, dl, dv = nv.ppo_loss(logits.contiguous(), values.contiguous(), actions, bits, old, adv, ret, 0.2, 0.5, 0.01)
    np.testing.assert_allclose(new_lp.cpu().numpy(), want["nlp"], atol=1e-5, rtol=1e-5)
    means = [want["pl"].mean(), want["vl"].mean(), want["el"].mean(), float(want["tot"]),
             float((old.cpu().numpy().astype(np.float64) - want["nlp"]).mean())]
    np.testing.assert_allclose(sums.cpu().numpy(), np.array(means, np.float32), atol=1e-5, rtol=1e-5)
    assert torch.isfinite(dl).all() and torch.isfinite(dv).all()


def test_default_shape_agent_on_every_device_path(dev):
    """One state-dict, four ways to evaluate it on the MI355X, against the reference's fp32 outputs:
    eager fp32 (1e-5), PyTorch bf16 autocast (the yardstick), the HIP bf16 update path (embedding, attention, add+LN,
    Linear blocks: grad mode + autocast) and the fused rollout encoder k_encoder (both forms)."""
    from src.ppo.fused_policy import FusedPolicy

    agent = default_shape_agent(dropout=0.0).to(dev)
    boards, actions = _t("default/boards", dev), _t("default/actions", dev)
    bits = _t("default/mask_bits", dev)
    masks = (bits.unsqueeze(-1) & torch.tensor(BITS, dtype=torch.uint8, device=dev)) != 0
    want = {k: _t(f"default/{k}", dev) for k in ("features", "logits", "values", "eval_logp", "eval_entropy")}
    agent.eval()
    with torch.no_grad():
        feats = agent.features(boards)
        logits, values = agent(boards, None)
        lp, _, ent = agent.evaluate_actions(boards, actions, masks)
    for got, key, atol in ((feats, "features", 3e-5), (logits, "logits", 1e-5), (values, "values", 1e-5),
                           (lp, "eval_logp", 1e-5), (ent, "eval_entropy", 1e-5)):
        np.testing.assert_allclose(got.cpu().numpy(), want[key].cpu().numpy(), atol=atol, rtol=1e-5, err_msg=key)
    # yardstick: what bf16 autocast of the plain PyTorch operators costs (no gradients -> no HIP kernel engages)
    with torch.no_grad(), torch.autocast("cuda", dtype=torch.bfloat16):
        f16 = agent.features(boards).float()
        l16, v16 = agent(boards, None)
    base_f = (f16 - want["features"]).abs().mean().item()
    base_l = (l16.float() - want["logits"]).abs().max().item()
    assert 0 < base_f < 0.05
    # the update path: train mode (dropout p = 0), gradients on, bf16 autocast -> every HIP kernel of the update runs
    agent.train()
    calls = {"attn": 0, "ln": 0, "embed": 0}
    orig = (nv.attn_fwd, nv.add_ln_fwd, nv.embed_fwd)

    def counted(name, fn):
        def wrapper(*a, **k):
            calls[name] += 1
            return fn(*a, **k)
        return wrapper

    nv.attn_fwd, nv.add_ln_fwd, nv.embed_fwd = (counted("attn", orig[0]), counted("ln", orig[1]), counted("embed", orig[2]))
    try:
        with torch.autocast("cuda", dtype=torch.bfloat16):
            fh = agent.features(boards)
            lh, vh = agent(boards, None)
    finally:
        nv.attn_fwd, nv.add_ln_fwd, nv.embed_fwd = orig
    # 2 forwards x 4 layers (the first LayerNorm of each forward runs inside the embedding kernel: g2048_embed_ln_fwd)
    assert calls["attn"] == 8 and calls["ln"] >= 14 and calls["embed"] == 2, calls
    err_f = (fh.float() - want["features"]).abs().mean().item()
    assert err_f < 1.5 * base_f + 1e-4, (err_f, base_f)
    assert (lh.float() - want["logits"]).abs().max().item() < 2.0 * base_l + 0.02
    assert (vh.float() - want["values"]).abs().max().item() < 0.05
    # the rollout encoder
    agent.eval()
    fp = FusedPolicy(agent)
    for split in (False, True):
        fk = fp.features(boards, split=split)
        err_k = (fk - want["features"]).abs().mean().item()
        assert err_k < 1.5 * base_f + 1e-4, (split, err_k, base_f)
    lk, vk = fp(boards)
    assert (lk - want["logits"]).abs().max().item() < 2.0 * base_l + 0.03
    assert (vk - want["values"].reshape(-1)).abs().max().item() < 0.05
    # a rollout decision is the argmax of logits + Gumbel noise: the greedy action must agree wherever the reference's
    # top-2 margin exceeds the bf16 noise
    top2 = want["logits"].topk(2, dim=1).values
    clear = (top2[:, 0] - top2[:, 1]) > 0.1
    assert (lk.argmax(1)[clear] == want["logits"].argmax(1)[clear]).all()


def _fp32_reference_grads(agent, tr, sample):
    """Pure-PyTorch fp32 backward of the reference loss (_compute_ppo_loss, no autocast, no HIP kernel) on a copy."""
    ref = copy.deepcopy(agent).float().train()
    ref.transformer._shadow = None
    ref._head_shadow = None
    tr_ref = PPOTrainer.__new__(PPOTrainer)
    tr_ref.agent, tr_ref.clip_epsilon, tr_ref.value_loss_coef, tr_ref.entropy_coef = ref, tr.clip_epsilon, tr.value_loss_coef, tr.entropy_coef
    tr_ref.use_action_mask = tr.use_action_mask
    masks = (sample["masks"].unsqueeze(-1) & torch.tensor(BITS, dtype=torch.uint8, device=sample["masks"].device)) != 0
    loss = tr_ref._compute_ppo_loss(sample["obs"], sample["actions"].long(), masks, sample["old_lp"], sample["adv"], sample["ret"])[0]
    loss.backward()
    return [p.grad.clone() for p in ref.parameters()], loss.item()


def _autocast_reference_grads(agent, tr, sample):
    """The yardstick: the reference loss back-propagated through PLAIN PyTorch modules under bf16 autocast (one-hot Linear,
    nn.TransformerEncoder, the heads; no kernel of this repository) on a copy of the weights -- what the reference's own update
    computes for this minibatch.  Its distance to the fp32 backward is what bf16 costs on THIS batch."""
    from torch.distributions import Categorical

    ref = copy.deepcopy(agent).float().train()
    ref.transformer._shadow = None
    ref._head_shadow = None
    masks = (sample["masks"].unsqueeze(-1) & torch.tensor(BITS, dtype=torch.uint8, device=sample["masks"].device)) != 0
    with torch.autocast("cuda", dtype=torch.bfloat16):
        logits, values = _torch_forward(ref, sample["obs"], "cls")
        if tr.use_action_mask:
            logits = logits - 1e8 * (1 - masks.float())
        dist = Categorical(logits=logits, validate_args=False)
        new_lp, entropy = dist.log_prob(sample["actions"].long()), dist.entropy()
        ratio = torch.exp(new_lp - sample["old_lp"])
        adv = sample["adv"]
        policy_loss = -torch.min(ratio * adv, torch.clamp(ratio, 1 - tr.clip_epsilon, 1 + tr.clip_epsilon) * adv)
        value_loss = torch.nn.functional.mse_loss(values.flatten(), sample["ret"], reduction="none")
        loss = (policy_loss + tr.value_loss_coef * value_loss - tr.entropy_coef * entropy).mean()
    loss.backward()
    return [p.grad.float().clone() for p in ref.parameters()]


def test_hip_graph_update_at_bench_shape_matches_fp32_backward(dev, tmp_path):
    """The shape bench.py runs (4 layers, ff 1024, minibatch 2048; dropout forced to 0 so every path is deterministic):
    eager HIP-path gradients AND hipGraph-replayed gradients, on batches other than the captured one, against a
    pure-PyTorch fp32 backward of the reference loss.  Also the regression test of the 02:59 fault's path (CLS-only last
    layer with packed bf16 shadows at minibatch 2048)."""
    from src.ppo.ppo_trainer import _GraphedFwdBwd

    torch.manual_seed(0)
    agent = PPOAgent(d_model=256, nhead=8, num_layers=4, dim_feedforward=1024, hidden_dim=512, dropout=0.0, reduction="cls")
    tr = PPOTrainer(agent, BatchRunner(0, device=dev), RolloutBuffer(31, 16, 4), dict(OPTIM), max_steps=1000, device=dev,
                    rollout_amp=True, log_dir=str(tmp_path), max_samples_per_epoch=100000, use_action_mask=True)
    assert tr.use_hip_graph
    tr.collect_rollouts(256, 1)
    M = 2048
    data = tr.rollout_buffer.device_data(dev)
    ds = PPODataset(data, gamma=tr.gamma, lambda_gae=tr.lambda_gae, max_samples_per_epoch=4 * M, shuffle_on_reset=False)
    db = DeviceBatches(ds, M, drop_last=True)
    samples = [db.gather_packed(idx) for idx in list(db.indices())[:3]]
    assert len(samples) == 3
    agent.train()
    scale = tr.scaler.get_scale()
    want = [_fp32_reference_grads(agent, tr, s) for s in samples]
    names = [n for n, _ in agent.named_parameters()]

    # PER BATCH (round 4; round 3 had bounded the error against the mean gradient norm of the three batches, behind which a bad
    # batch could hide): the bf16 noise of a minibatch gradient is a sum of 2048 per-board rounding errors and does not shrink when
    # the boards' contributions cancel, while the gradient itself does (z-scored returns: its size follows the batch mean of R;
    # tools/debug_benchshape_rows.py: |g_hip - g32| = 0.0138 / 0.0174 / 0.0203 for |g32| = 0.307 / 0.419 / 0.181).  So the yardstick
    # for batch i is what PyTorch's OWN bf16 autocast backward loses on batch i (`_autocast_reference_grads`): the HIP path may be
    # 1.3 x that + 1 % of the batch's gradient norm, as a whole and per parameter tensor, and its direction no worse than
    # autocast's by more than 0.003 in cosine.
    yard = [_autocast_reference_grads(agent, tr, smp) for smp in samples]
    flat = lambda gs: torch.cat([g.flatten() for g in gs])
    report = []

    def check(tag, i):
        flat_g = torch.cat([(p.grad / scale).flatten() for p in agent.parameters()])
        flat_w, flat_y = flat(want[i][0]), flat(yard[i])
        assert torch.isfinite(flat_g).all(), tag
        cos = torch.nn.functional.cosine_similarity(flat_g, flat_w, dim=0).item()
        cos_y = torch.nn.functional.cosine_similarity(flat_y, flat_w, dim=0).item()
        err, err_y, gn = (flat_g - flat_w).norm().item(), (flat_y - flat_w).norm().item(), flat_w.norm().item()
        report.append((tag, i, round(err / gn, 4), round(err_y / gn, 4), round(cos, 5), round(cos_y, 5)))
        assert err < 1.3 * err_y + 0.01 * gn and cos > cos_y - 0.003, (tag, i, cos, cos_y, err, err_y, gn)
        bad = {}
        for n, p, gw, gy in zip(names, agent.parameters(), want[i][0], yard[i]):
            e_hip, e_y = (p.grad / scale - gw).norm().item(), (gy - gw).norm().item()
            if not e_hip < 1.3 * e_y + 0.01 * gw.norm().item() + 1e-3 * gn:  # (floor for tensors with a near-zero gradient)
                bad[n] = (round(e_hip / gn, 5), round(e_y / gn, 5))
        assert not bad, (tag, i, bad)

    for i, s in enumerate(samples):  # eager HIP path
        stats, _ = tr._loss_backward(**s)
        assert abs(stats[3].item() - want[i][1]) < 0.02 * max(1.0, abs(want[i][1]))
        check("eager", i)
    gr = _GraphedFwdBwd(tr, M, samples[0])
    for i in (1, 2, 0, 2):  # replays on other batches than the captured one
        stats, _ = gr.run(samples[i])
        assert abs(stats[3].item() - want[i][1]) < 0.02 * max(1.0, abs(want[i][1]))
        check("graph", i)
    print("bench-shape gradient check (tag, batch, |g_hip - g32| / |g32|, |g_autocast - g32| / |g32|, cos hip, cos autocast):", report)


def test_update_policy_replays_a_graph_at_bench_shape(dev, tmp_path):
    """update_policy at bench.py's model shape and minibatch with the default dropout: the captured graph must still be in
    use afterwards (a silent eager fallback would halve the update's throughput) and the parameters stay finite."""
    torch.manual_seed(1)
    agent = PPOAgent(d_model=256, nhead=8, num_layers=4, dim_feedforward=1024, hidden_dim=512, dropout=0.1, reduction="cls")
    tr = PPOTrainer(agent, BatchRunner(0, device=dev), RolloutBuffer(31, 16, 4), dict(OPTIM), max_steps=1000, device=dev,
                    rollout_amp=True, log_dir=str(tmp_path), max_samples_per_epoch=3 * 2048 + 5, use_action_mask=True,
                    target_kl=0.25)
    tr.collect_rollouts(256, 1)
    before = [p.detach().clone() for p in agent.parameters()]
    m = tr.update_policy(batch_size=2048, n_epochs=2)
    assert m["n_updates"] >= 3 and m["hip_graph"] is True and m["hip_graphs_captured"] == 1
    assert tr.use_hip_graph and tr._graphs and tr.hip_graph_fallback is None
    assert all(torch.isfinite(p).all() for p in agent.parameters())
    assert any(not torch.equal(a, b) for a, b in zip(before, agent.parameters()))


def test_colsum_wide_and_capture_guard(dev):
    """g2048_colsum tiles matrices wider than 1024 columns; a shape outside it must refuse to be captured in a hipGraph
    (at::sum does not survive a replay on this stack) instead of silently producing wrong bias gradients."""
    from src.ppo.hip_ops import _colsum

    torch.manual_seed(4)
    for T, N in ((300, 1028), (4097, 1536), (50, 4096)):
        for dt in (torch.bfloat16, torch.float32):
            x = torch.randn(T, N, device=dev).to(dt)
            got = _colsum(x)
            assert torch.allclose(got.double(), x.double().sum(0), rtol=2e-5, atol=2e-4 * T ** 0.5)
    odd = torch.randn(64, 6, device=dev)
    assert torch.allclose(_colsum(odd), odd.sum(0), atol=1e-4)  # eager: at::sum fallback is fine
    g = torch.cuda.CUDAGraph()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        _colsum(odd)
    torch.cuda.current_stream().wait_stream(side)
    with pytest.raises(RuntimeError, match="must not be captured"):
        with torch.cuda.graph(g):
            _colsum(odd)


def _torch_forward(agent, boards, reduction):
    """The agent's forward out of plain PyTorch modules only (one-hot Linear, nn.TransformerEncoder, the heads): what autocast makes
    of the reference's architecture, no kernel of this repository."""
    import torch.nn.functional as F

    t = agent.transformer
    x = agent.input_embedding(F.one_hot(boards.long(), 31).float())
    x = x + t.positional_encoding.flat_table().unsqueeze(0).to(x.dtype)
    x = torch.cat([t.cls_token.expand(x.shape[0], -1, -1).to(x.dtype), x], dim=1)
    x = t.encoder(x)
    feats = x[:, 0] if reduction == "cls" else x[:, 1:].mean(dim=1)  # "mean": over the 16 board tokens
    return agent.actor(feats), agent.critic(feats)


@pytest.mark.parametrize("sink", [False, True], ids=["autograd", "sink"])
@pytest.mark.parametrize("reduction", ["cls", "mean"])
def test_update_path_gradients_per_tensor_at_minibatch_size(dev, reduction, sink):
    """Every parameter gradient of the default-shape agent through the HIP update path (bf16 autocast, dropout 0, minibatch 2048 =
    34 816 tokens: `k_linear_ws`, `k_dweight` with column sums, attention, add+LayerNorm, the fused CLS tail for "cls" / the full last
    layer for "mean") - per TENSOR, relative to the tensor's own gradient norm, against an fp32 PyTorch backward of the same weights,
    with PyTorch's own bf16 autocast backward as the yardstick (ReLU units whose pre-activation changes sign under bf16 rounding put
    ~5 % on every tensor whichever bf16 implementation runs).  Random downstream gradients on logits and values, so nothing cancels
    (unlike the PPO loss of the bench-shape test).  "sink": the trainer's route - a `GradSink` over a flat f32 bucket, every
    minibatch-sized weight gradient deferred into the one grouped `g2048_dweight_jobs` launch (8 token slices) and all second-stage
    sums in one `g2048_reduce_jobs` launch."""
    from src.ppo.hip_ops import GradSink, grad_sink

    agent = default_shape_agent(dropout=0.0).to(dev)
    agent.reduction = reduction
    agent.train()
    g = torch.Generator(device="cpu").manual_seed(5)
    boards = torch.randint(0, 12, (2048, 16), generator=g, dtype=torch.uint8).to(dev)
    gl, gv = torch.randn(2048, 4, generator=g).to(dev), torch.randn(2048, generator=g).to(dev)

    def grads(model, fwd, autocast, use_sink=False):
        model.zero_grad(set_to_none=True)
        with torch.autocast("cuda", dtype=torch.bfloat16, enabled=autocast):
            lo, va = fwd(model)
        ps = list(model.parameters())
        bucket = [torch.zeros_like(p, dtype=torch.float32) for p in ps]
        s = GradSink({id(p): b for p, b in zip(ps, bucket)}) if use_sink else None
        with grad_sink(s):
            ((lo.float() * gl).sum() + (va.float().flatten() * gv).sum()).backward()
        if use_sink:
            assert len(s.written) > len(ps) // 2, "the sink took few gradients: not the trainer's route"
        out = [b if (use_sink and id(p) in s.written) else p.grad.detach().float().clone() for p, b in zip(ps, bucket)]
        return out, lo.detach().float(), va.detach().float().flatten()

    ref = copy.deepcopy(agent).float()
    ref.transformer._shadow, ref._head_shadow = None, None
    g32, l32, v32 = grads(ref, lambda m: _torch_forward(m, boards, reduction), False)
    g16, _, _ = grads(ref, lambda m: _torch_forward(m, boards, reduction), True)
    gh, lh, vh = grads(agent, lambda m: m(boards, None), True, use_sink=sink)
    assert (lh - l32).abs().max().item() < 0.05 and (vh - v32).abs().max().item() < 0.05
    rel = lambda a, b: ((a - b).norm() / b.norm().clamp_min(1e-12)).item()
    bad = {}
    for (n, _), a32, a16, ah in zip(agent.named_parameters(), g32, g16, gh):
        assert torch.isfinite(ah).all(), n
        e_hip, e_torch = rel(ah, a32), rel(a16, a32)
        if not e_hip < 1.3 * e_torch + 0.01:
            bad[n] = (round(e_hip, 4), round(e_torch, 4))
    assert not bad, (reduction, bad)
