"""Device-resident buffer -> GAE -> PPO update on the GPU, against reference-produced fixtures and invariants."""
import os

import numpy as np
import pytest
import torch

from oracle import c_oracle as orc
from oracle import g2048_oracle as npo
from src.ppo import MLPAgent, PPOAgent, PPOTrainer, RolloutBuffer, TorchActionFunction, create_ppo_dataloader
from src.ppo.data_loader import DeviceBatches, PPODataset
from src.runs import BatchRunner

pytestmark = pytest.mark.gpu
REF = np.load(os.path.join(os.path.dirname(__file__), "golden", "torch_reference.npz"))
OPTIM = dict(opt_name="adamw", max_lr=4e-4, betas=(0.9, 0.999), eps=1e-6, weight_decay=0.01, warmup_steps_ratio=0.025,
             scheduler_names=["constant", "constant"], blacklist_weight_modules=["norm", "embedding"])


def _buffer_dict(i):
    N = len(REF[f"gae{i}/rewards"])
    return dict(observations=np.zeros((N, 16, 31), np.float32), actions=np.zeros((N, 4), np.float32),
                action_masks=np.ones((N, 4), bool), rewards=REF[f"gae{i}/rewards"], values=REF[f"gae{i}/values"],
                log_probs=np.zeros(N, np.float32), terminations=REF[f"gae{i}/terms"])


@pytest.mark.parametrize("i", [0, 1, 2])
def test_dataset_gae_matches_reference(dev, i):
    gamma, lam = (float(x) for x in REF[f"gae{i}/params"])
    ds = PPODataset(_buffer_dict(i), gamma=gamma, lambda_gae=lam)
    assert (ds.raw_advantages.cpu().numpy() == REF[f"gae{i}/raw_adv"]).all()  # bit-exact scan
    assert (ds.raw_returns.cpu().numpy() == REF[f"gae{i}/raw_ret"]).all()
    np.testing.assert_allclose(ds.advantages.cpu().numpy(), REF[f"gae{i}/adv"], atol=1e-5, rtol=1e-5)
    np.testing.assert_allclose(ds.returns.cpu().numpy(), REF[f"gae{i}/ret"], atol=1e-5, rtol=1e-5)


def test_dataloader_protocol(dev):
    data = _buffer_dict(0)
    dl = create_ppo_dataloader(data, batch_size=32, shuffle=True, drop_last=True)
    assert len(dl.dataset) == 300 and len(dl) == 9
    batch = next(iter(dl))
    assert set(batch) == {"observations", "actions", "action_masks", "rewards", "values", "log_probs",
                          "terminations", "advantages", "returns"}
    assert batch["observations"].shape == (32, 16, 31) and batch["action_masks"].dtype == torch.bool
    ds = PPODataset(data, max_samples_per_epoch=100, shuffle_on_reset=True)
    assert len(ds) == 100 and ds.total_length == 300
    first = ds.active_indices.clone()
    ds.reset_epoch()
    assert not torch.equal(first, ds.active_indices)
    ds2 = PPODataset(data, max_samples_per_epoch=100, shuffle_on_reset=False)
    first = ds2.active_indices.clone()
    ds2.reset_epoch()
    assert torch.equal(first, ds2.active_indices)
    assert len(PPODataset(data, max_samples_per_epoch=1000)) == 300
    seen = torch.cat([b["log_probs"] for b in DeviceBatches(ds, 32).epoch()])
    assert seen.numel() == 96  # drop_last


def test_device_buffer_equals_reference_buffer_semantics(dev):
    """store_trajectory (HIP compaction) == the reference's store_batch on the same [B, T] arrays."""
    torch.manual_seed(0)
    agent = PPOAgent(hidden_dim=32, d_model=32, nhead=4, num_layers=1, dim_feedforward=64, dropout=0.0)
    runner = BatchRunner(init_seed=2, act_fn=TorchActionFunction(agent, use_mask=True, device=dev))
    traj = runner.collect(24, fill_frozen=True)
    buf = RolloutBuffer(31, 16, 4)
    n = buf.store_trajectory(traj)
    got = buf.get_buffer_data()
    # the same data through the numpy interface, as the reference trainer would feed it
    bt = lambda x: np.swapaxes(x.cpu().numpy(), 0, 1)
    obs = npo.observation(bt(traj.boards).reshape(-1, 16)).reshape(24, traj.T, 4, 4, 31)
    act = np.eye(4, dtype=np.float32)[bt(traj.actions)]
    msk = ((bt(traj.masks)[..., None] >> np.arange(4)) & 1).astype(bool)
    host = RolloutBuffer(31, 16, 4)
    host.store_batch(obs, act, msk, bt(traj.rewards), bt(traj.values), bt(traj.log_probs), bt(traj.terms).astype(bool))
    want = host.get_buffer_data()
    assert n == host.buffer_size == buf.buffer_size == int(traj.ep_len.sum())
    for k in want:
        assert got[k].dtype == want[k].dtype and got[k].shape == want[k].shape, k
        assert (got[k] == want[k]).all(), k
    assert got["terminations"].sum() == 24  # exactly one terminal step per env, each segment ends with it
    # GAE over the device buffer == oracle scan over the reference layout
    ds = PPODataset(buf.device_data(), gamma=0.99, lambda_gae=0.95)
    oa, orr = orc.gae(want["rewards"], want["values"], want["terminations"], 0.99, 0.95)
    assert (ds.raw_advantages.cpu().numpy() == oa).all() and (ds.raw_returns.cpu().numpy() == orr).all()


def _trainer(dev, agent, **kw):
    args = dict(gamma=0.99, lambda_gae=0.95, clip_epsilon=0.2, value_loss_coef=0.5, entropy_coef=0.01,
                max_grad_norm=0.5, target_kl=0.25, use_action_mask=True, device=dev, mixed_precision="bfloat16",
                max_samples_per_epoch=2000, shuffle_on_reset=True)
    args.update(kw)
    return PPOTrainer(agent, BatchRunner(init_seed=0), RolloutBuffer(31, 16, 4), OPTIM, max_steps=1000, **args)


@pytest.mark.parametrize("kind", ["transformer", "mlp"])
def test_collect_update_smoke(dev, kind, tmp_path, monkeypatch):
    monkeypatch.chdir(tmp_path)
    torch.manual_seed(0)
    agent = (PPOAgent(hidden_dim=64, d_model=64, nhead=4, num_layers=2, dim_feedforward=128, reduction="cls")
             if kind == "transformer" else MLPAgent(hidden_dim=64, trunk_dim=64))
    tr = _trainer(dev, agent)
    before = [p.detach().clone() for p in agent.parameters()]
    tr.collect_rollouts(batch_size=32, num_batches=2)
    n = tr.rollout_buffer.buffer_size
    assert n > 0 and tr.total_timesteps == n and len(tr.episode_rewards) == 64
    assert tr.episode_rewards[-10:] == list(tr.episode_rewards)[-10:]  # sliceable history
    assert all(l >= 1 for l in tr.episode_lengths)
    m = tr.update_policy(batch_size=256, n_epochs=2)
    assert m["n_updates"] >= 2 and np.isfinite([m["policy_loss"], m["value_loss"], m["entropy_loss"], m["total_loss"],
                                                m["kl_divergence"]]).all()
    assert tr.total_update_steps == m["n_updates"] and tr.total_epochs >= 1
    assert any(not torch.equal(a, b) for a, b in zip(before, agent.parameters()))
    # checkpoint round trip, reference key set
    tr.save_checkpoint("ck.pt")
    ck = torch.load("ck.pt", weights_only=False)
    assert {"agent_state_dict", "optimizer_state_dict", "total_timesteps", "total_epochs", "total_update_steps",
            "episode_rewards", "episode_lengths", "last_save_timestep", "scaler_state_dict"} <= set(ck)
    agent2 = type(agent)(**({"hidden_dim": 64, "d_model": 64, "nhead": 4, "num_layers": 2, "dim_feedforward": 128,
                             "reduction": "cls"} if kind == "transformer" else {"hidden_dim": 64, "trunk_dim": 64}))
    tr2 = _trainer(dev, agent2)
    tr2.load_checkpoint("ck.pt", load_optimizer=True)
    assert tr2.total_timesteps == tr.total_timesteps and tr2.total_update_steps == tr.total_update_steps
    for a, b in zip(agent.state_dict().values(), agent2.state_dict().values()):
        assert torch.equal(a, b)
    with pytest.raises(ValueError, match="missing required keys"):
        torch.save({"agent_state_dict": {}}, "bad.pt")
        tr2.load_checkpoint("bad.pt")


def test_train_loop_and_resume_modes(dev, tmp_path, monkeypatch):
    monkeypatch.chdir(tmp_path)
    torch.manual_seed(0)
    tr = _trainer(dev, MLPAgent(hidden_dim=32, trunk_dim=32), mixed_precision=None)
    tr.train(total_timesteps=1500, rollout_batch_size=16, rollout_batches=1, update_epochs=1, train_batch_size=128,
             save_freq=10**9)
    assert tr.total_timesteps >= 1500 and os.path.exists("final_model.pt")
    t = tr.total_timesteps
    tr.train(total_timesteps=100, rollout_batch_size=16, rollout_batches=1, update_epochs=1, train_batch_size=128,
             resume_extend_steps=False)  # already past the absolute target: nothing happens
    assert tr.total_timesteps == t
    tr.train(total_timesteps=500, rollout_batch_size=16, rollout_batches=1, update_epochs=1, train_batch_size=128,
             resume_extend_steps=True)
    assert tr.total_timesteps >= t + 500


def test_kl_early_stop(dev, tmp_path, monkeypatch):
    monkeypatch.chdir(tmp_path)
    torch.manual_seed(0)
    tr = _trainer(dev, MLPAgent(hidden_dim=32, trunk_dim=32), target_kl=-1.0, mixed_precision=None)
    tr.collect_rollouts(16, 1)
    tr.update_policy(batch_size=128, n_epochs=5)
    assert tr.total_epochs == 1  # mean(old - new) > target after the first epoch -> stop


def test_fused_encoder_matches_autocast_forward(dev):
    """g2048_policy_encoder (bf16 MFMA megakernel) vs the PyTorch encoder under bf16 autocast: same numerics class
    (its distance to autocast is smaller than autocast's own distance to fp32), any batch size, any layer count."""
    from src.ppo.fused_policy import FusedPolicy, supports

    torch.manual_seed(0)
    for layers in (1, 4):
        agent = PPOAgent(hidden_dim=512, d_model=256, nhead=8, num_layers=layers, dim_feedforward=1024,
                         reduction="cls").to(dev).eval()
        with torch.no_grad():
            for p in agent.parameters():
                if p.dim() == 1:
                    p.add_(torch.randn_like(p) * 0.05)  # non-trivial biases / LayerNorm affine
        assert supports(agent)
        fp = FusedPolicy(agent)
        for B in (1, 7, 8, 100, 4097):
            boards = torch.randint(0, 14, (B, 16), dtype=torch.uint8, device=dev)
            with torch.no_grad():
                ref32 = agent.features(boards)
                with torch.autocast("cuda", dtype=torch.bfloat16):
                    ref16 = agent.features(boards).float()
                    l16, v16 = agent(boards)
            got = fp.features(boards)
            assert torch.isfinite(got).all()
            err, base = (got - ref16).abs().mean().item(), (ref16 - ref32).abs().mean().item()
            assert err < 1.5 * base + 1e-4, (layers, B, err, base)
            assert (got - ref32).abs().max().item() < 0.05 * max(1.0, ref32.abs().max().item())
            logits, values = fp(boards)
            assert (logits - l16.float()).abs().max().item() < 0.05 and (values - v16.float().flatten()).abs().max().item() < 0.05
    assert not supports(PPOAgent(d_model=128, nhead=8, num_layers=1, dim_feedforward=256).to(dev))
    assert not supports(PPOAgent(reduction="mean").to(dev))


def test_bf16_rollout_uses_fused_encoder_and_trains(dev, tmp_path, monkeypatch):
    monkeypatch.chdir(tmp_path)
    torch.manual_seed(0)
    agent = PPOAgent(hidden_dim=512, d_model=256, nhead=8, num_layers=2, dim_feedforward=1024, reduction="cls")
    tr = _trainer(dev, agent, rollout_amp=True)
    tr.collect_rollouts(64, 1)
    assert tr.batch_runner.act_fn._fused is not None
    m = tr.update_policy(batch_size=256, n_epochs=1)
    assert np.isfinite(m["total_loss"]) and abs(m["kl_divergence"]) < 0.05  # rollout and update policies agree


def test_small_attention_kernels_match_sdpa(dev):
    """g2048_attn_fwd/bwd (17 tokens, head_dim 32) vs torch SDPA: outputs and gradients as close to an fp32 reference as
    SDPA's own bf16 path, packed and CLS-row variants, odd batch sizes; dropout keeps the expectation."""
    import torch.nn.functional as F

    from src.ppo.transformer_encoder import _AttnCls, _AttnPacked

    H, hd, S = 8, 32, 17
    rel = lambda a, b: ((a.float() - b.float()).norm() / b.float().norm()).item()
    torch.manual_seed(0)
    for B in (1, 5, 683):
        qkv = (torch.randn(B, S, 3 * H * hd, device=dev) * 1.5).to(torch.bfloat16).requires_grad_(True)
        g = torch.randn(B, S, H * hd, device=dev).to(torch.bfloat16)
        o = _AttnPacked.apply(qkv, H, 0.0)
        o.backward(g)
        x32 = qkv.detach().float().requires_grad_(True)
        q, k, v = x32.view(B, S, 3, H, hd).unbind(2)
        o32 = F.scaled_dot_product_attention(q.transpose(1, 2), k.transpose(1, 2), v.transpose(1, 2)).transpose(1, 2).reshape(B, S, -1)
        o32.backward(g.float())
        assert rel(o, o32) < 4e-3 and rel(qkv.grad, x32.grad) < 4e-3
        qc = torch.randn(B, 1, H * hd, device=dev).to(torch.bfloat16).requires_grad_(True)
        kv = torch.randn(B, S, 2 * H * hd, device=dev).to(torch.bfloat16).requires_grad_(True)
        gc = torch.randn(B, 1, H * hd, device=dev).to(torch.bfloat16)
        oc = _AttnCls.apply(qc, kv, H, 0.0)
        oc.backward(gc)
        q32, kv32 = qc.detach().float().requires_grad_(True), kv.detach().float().requires_grad_(True)
        k32, v32 = kv32.view(B, S, 2, H, hd).unbind(2)
        oc32 = F.scaled_dot_product_attention(q32.view(B, 1, H, hd).transpose(1, 2), k32.transpose(1, 2),
                                              v32.transpose(1, 2)).transpose(1, 2).reshape(B, 1, -1)
        oc32.backward(gc.float())
        assert rel(oc, oc32) < 4e-3 and rel(qc.grad, q32.grad) < 4e-3 and rel(kv.grad, kv32.grad) < 4e-3
    qkv = torch.randn(2048, S, 3 * H * hd, device=dev).to(torch.bfloat16)
    o0 = _AttnPacked.apply(qkv.clone().requires_grad_(True), H, 0.0).float()
    acc = torch.zeros_like(o0)
    for _ in range(16):
        acc += _AttnPacked.apply(qkv.clone().requires_grad_(True), H, 0.1).float()
    assert rel(acc / 16, o0) < 0.15  # unbiased: the 16-sample mean approaches the no-dropout output
    a = _AttnPacked.apply(qkv.clone().requires_grad_(True), H, 0.1)
    b = _AttnPacked.apply(qkv.clone().requires_grad_(True), H, 0.1)
    assert not torch.equal(a, b)  # fresh mask per call
