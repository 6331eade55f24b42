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


def test_mlp_policy_full_loop_at_4096_envs(dev, tmp_path, monkeypatch):
    """BASELINE.json configs[1]: 4 096 parallel envs, MLP policy, full PPO loop with the reference's trainer config
    (5 epochs, minibatch 2048, 300 k samples per epoch, bf16): two iterations learn something measurable and every
    episode of the rollout is a complete one."""
    monkeypatch.chdir(tmp_path)
    torch.manual_seed(0)
    agent = MLPAgent(hidden_dim=512, trunk_dim=512)
    tr = PPOTrainer(agent, BatchRunner(init_seed=0, device=dev), RolloutBuffer(31, 16, 4), OPTIM, max_steps=500000, device=dev,
                    gamma=0.99, lambda_gae=0.95, clip_epsilon=0.2, value_loss_coef=0.5, entropy_coef=0.01, max_grad_norm=0.5,
                    target_kl=0.25, use_action_mask=True, mixed_precision="bfloat16", max_samples_per_epoch=300000,
                    shuffle_on_reset=True, rollout_amp=True)
    lengths = []
    for _ in range(2):
        tr.collect_rollouts(4096, 1)
        n = tr.rollout_buffer.buffer_size
        data = tr.rollout_buffer.device_data(dev)
        assert int(data["terms"].sum()) == 4096 and n == tr.last_rollout_stats["timesteps"]  # 4 096 complete episodes
        assert (data["rewards"] >= 0).all()  # masked policy: no illegal move
        m = tr.update_policy(batch_size=2048, n_epochs=5)
        assert m["n_updates"] == 5 * (min(n, 300000) // 2048) and np.isfinite(m["total_loss"])
        assert m["hip_graph"] is True
        lengths.append(tr.last_rollout_stats["mean_episode_length"])
    assert len(tr.episode_lengths) == 8192 and lengths[0] > 50
    assert all(torch.isfinite(p).all() for p in agent.parameters())


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
        for B in (1, 7, 8, 100, 129, 4097):
            boards = torch.randint(0, 14, (B, 16), dtype=torch.uint8, device=dev)
            with torch.no_grad():
                ref32 = agent.features(boards)
                with torch.autocast("cuda", dtype=torch.bfloat16):
                    ref16 = agent.features(boards).float()
                    l16, v16 = agent(boards)
            for split in (False, True):  # one kernel / CLS-only last layer in a second kernel
                got = fp.features(boards, split=split)
                assert torch.isfinite(got).all()
                err, base = (got - ref16).abs().mean().item(), (ref16 - ref32).abs().mean().item()
                assert err < 1.5 * base + 1e-4, (layers, B, split, err, base)
                assert (got - ref32).abs().max().item() < 0.05 * max(1.0, ref32.abs().max().item())
            one, two = fp.features(boards, split=False), fp.features(boards, split=True)
            assert (one - two).abs().max().item() < 0.02 * max(1.0, one.abs().max().item())
            logits, values = fp(boards)
            assert (logits - l16.float()).abs().max().item() < 0.05 and (values - v16.float().flatten()).abs().max().item() < 0.05
            # the heads on g2048_gemm_jobs / g2048_mlp_out_fwd (G2048_OWN_HEADS_MAX, opt-in) against the library GEMMs (default)
            assert fp.own is not None
            fp.OWN_HEADS_MAX_BOARDS = 16384
            try:
                l_own, v_own = fp(boards)
            finally:
                del fp.OWN_HEADS_MAX_BOARDS  # (back to the class attribute)
            assert l_own.dtype == torch.float32 and l_own.shape == logits.shape and v_own.shape == values.shape
            assert (logits - l_own).abs().max().item() < 0.03 and (values - v_own).abs().max().item() < 0.03
            assert (l_own - l16.float()).abs().max().item() < 0.05 and (v_own - v16.float().flatten()).abs().max().item() < 0.05
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

    from src.ppo.hip_ops import _AttnCls, _AttnPacked

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


def test_mfma_attention_equals_scalar_attention(dev, monkeypatch):
    """The 17-token attention on MFMA tiles (default) vs the scalar kernels (G2048_ATTN_SCALAR=1) on the same inputs and the
    SAME dropout seed: both use the element index (pair * 17 + query) * 32 + key for the mask, so outputs, log-sum-exps and
    gradients agree to bf16 rounding and either forward can be paired with either backward."""
    from src.g2048 import native as nv

    H, hd, S = 8, 32, 17
    rel = lambda a, b: ((a.float() - b.float()).norm() / b.float().norm().clamp_min(1e-20)).item()
    torch.manual_seed(3)
    for B in (1, 7, 300, 2048):
        for p_drop in (0.0, 0.1):
            qkv = (torch.randn(B, S, 3 * H * hd, device=dev) * 1.5).to(torch.bfloat16)
            do = torch.randn(B, S, H * hd, device=dev).to(torch.bfloat16)
            W, hw = 3 * H * hd, H * hd
            base = qkv.data_ptr()
            res = {}
            for impl in ("mfma", "scalar"):
                monkeypatch.setenv("G2048_ATTN_SCALAR", "1" if impl == "scalar" else "0")
                o = torch.empty((B, S, hw), dtype=torch.bfloat16, device=dev)
                lse = torch.empty((B, H, S), dtype=torch.float32, device=dev)
                nv.attn_fwd(base, base + 2 * hw, base + 4 * hw, o, lse, B, H, S, (S * W, W) * 3, hd ** -0.5, p_drop, 1234, 0)
                dqkv = torch.empty_like(qkv)
                db = dqkv.data_ptr()
                nv.attn_bwd(base, base + 2 * hw, base + 4 * hw, do, lse, db, db + 2 * hw, db + 4 * hw, B, H, S, (S * W, W) * 3,
                            hd ** -0.5, p_drop, 1234, 0)
                res[impl] = (o, lse, dqkv)
            (o1, l1, g1), (o2, l2, g2) = res["mfma"], res["scalar"]
            assert rel(o1, o2) < 6e-3, (B, p_drop, rel(o1, o2))
            assert torch.allclose(l1, l2, rtol=1e-5, atol=2e-2), (l1 - l2).abs().max()  # scores from bf16 MFMA vs f32 FMAs
            assert rel(g1, g2) < 8e-3, (B, p_drop, rel(g1, g2))
            if p_drop > 0:  # the same entries are dropped: zeros of the attention-weighted sums cannot be compared directly,
                # but an output row with every key dropped is exactly zero in both
                assert torch.equal((o1 == 0).all(-1), (o2 == 0).all(-1))


def test_mfma_attention_dropout_index_forms_agree_bitwise(dev, monkeypatch):
    """The MFMA attention kernels take the dropout decision of element base + c from (uint32)base * C + c * C when every element
    index of the launch is below 2^32 (host-checked) and from the plain 64-bit form otherwise (more than 7.9 M (sample, head) pairs:
    not reachable in a test).  G2048_ATTN_WIDE_INDEX=1 forces the 64-bit form: outputs, log-sum-exps and gradients must be equal
    bit for bit, with and without the device-resident seed word."""
    from src.g2048 import native as nv

    H, hd, S = 8, 32, 17
    W, hw = 3 * H * hd, H * hd
    torch.manual_seed(5)
    state = torch.tensor([0x1234_5678_9ABC_DEF0], dtype=torch.int64, device=dev)
    for B in (3, 777):
        qkv = (torch.randn(B, S, W, device=dev) * 1.5).to(torch.bfloat16)
        do = torch.randn(B, S, hw, device=dev).to(torch.bfloat16)
        base = qkv.data_ptr()
        for seed_state in (0, state.data_ptr()):
            res = {}
            for wide in ("0", "1"):
                monkeypatch.setenv("G2048_ATTN_WIDE_INDEX", wide)
                o = torch.empty((B, S, hw), dtype=torch.bfloat16, device=dev)
                lse = torch.empty((B, H, S), dtype=torch.float32, device=dev)
                nv.attn_fwd(base, base + 2 * hw, base + 4 * hw, o, lse, B, H, S, (S * W, W) * 3, hd ** -0.5, 0.3, 99, seed_state)
                dqkv = torch.empty_like(qkv)
                db = dqkv.data_ptr()
                nv.attn_bwd(base, base + 2 * hw, base + 4 * hw, do, lse, db, db + 2 * hw, db + 4 * hw, B, H, S, (S * W, W) * 3,
                            hd ** -0.5, 0.3, 99, seed_state)
                res[wide] = (o, lse, dqkv)
            monkeypatch.delenv("G2048_ATTN_WIDE_INDEX")
            for a, b in zip(res["0"], res["1"]):
                assert torch.equal(a, b), (B, seed_state != 0)
            o0, lse0 = torch.empty_like(res["0"][0]), torch.empty_like(res["0"][1])
            nv.attn_fwd(base, base + 2 * hw, base + 4 * hw, o0, lse0, B, H, S, (S * W, W) * 3, hd ** -0.5, 0.0, 99, seed_state)
            assert not torch.equal(o0, res["0"][0])  # dropout was on


def test_cls_attention_row_coalesced_equals_scalar(dev, monkeypatch):
    """The CLS-row attention (Sq = 1) with a 32-lane group per sample and 512-byte K/V row fetches (default for 8 heads) vs
    one lane per (sample, head) pair (G2048_ATTN_SCALAR=1), same inputs and dropout seed, K/V read from the packed
    [B, 17, 2 * 256] projection: f32 arithmetic in both, only the summation order differs; ragged B (the last workgroup's
    groups exit early)."""
    from src.g2048 import native as nv

    H, hd, S = 8, 32, 17
    rel = lambda a, b: ((a.float() - b.float()).norm() / b.float().norm().clamp_min(1e-20)).item()
    torch.manual_seed(4)
    for B in (1, 3, 301, 2048):
        for p_drop in (0.0, 0.1):
            hw = H * hd
            q = (torch.randn(B, 1, hw, device=dev) * 1.5).to(torch.bfloat16)
            kv = (torch.randn(B, S, 2 * hw, device=dev) * 1.5).to(torch.bfloat16)
            do = torch.randn(B, 1, hw, device=dev).to(torch.bfloat16)
            strides = (hw, 0, S * 2 * hw, 2 * hw, S * 2 * hw, 2 * hw)
            res = {}
            for impl in ("rows", "scalar"):
                monkeypatch.setenv("G2048_ATTN_SCALAR", "1" if impl == "scalar" else "0")
                o = torch.empty((B, 1, hw), dtype=torch.bfloat16, device=dev)
                lse = torch.empty((B, H, 1), dtype=torch.float32, device=dev)
                nv.attn_fwd(q.data_ptr(), kv.data_ptr(), kv.data_ptr() + 2 * hw, o, lse, B, H, 1, strides, hd ** -0.5, p_drop, 77, 0)
                dq, dkv = torch.empty_like(q), torch.full_like(kv, float("nan"))
                nv.attn_bwd(q.data_ptr(), kv.data_ptr(), kv.data_ptr() + 2 * hw, do, lse, dq.data_ptr(), dkv.data_ptr(),
                            dkv.data_ptr() + 2 * hw, B, H, 1, strides, hd ** -0.5, p_drop, 77, 0)
                res[impl] = (o, lse, dq, dkv)
            (o1, l1, q1, k1), (o2, l2, q2, k2) = res["rows"], res["scalar"]
            assert torch.isfinite(k1.float()).all()  # every K / V gradient row was written
            assert rel(o1, o2) < 4e-3 and torch.allclose(l1, l2, rtol=1e-5, atol=1e-4), (B, p_drop)
            assert rel(q1, q2) < 4e-3 and rel(k1, k2) < 4e-3, (B, p_drop, rel(q1, q2), rel(k1, k2))
            if p_drop > 0:
                assert torch.equal(k1[..., hw:] == 0, k2[..., hw:] == 0)  # dV rows of dropped keys are zero in both


def test_fused_add_layernorm_matches_torch(dev):
    """g2048_add_ln_fwd/bwd vs torch (`x + dropout(a)` then F.layer_norm in f32, cast to bf16): values and all five
    gradients, contiguous and strided ([B, 1, 256] slice) residual input, ragged row counts; with dropout the kept
    elements are scaled by 1/(1-p), the mask differs per call and the backward uses the forward's mask."""
    import torch.nn.functional as F

    from src.ppo.hip_ops import _AddLayerNorm

    rel = lambda a, b: ((a.float() - b.float()).norm() / b.float().norm().clamp_min(1e-12)).item()
    torch.manual_seed(1)
    for shape, sliced in (((3, 17, 256), False), ((1, 1, 256), False), ((2049, 17, 256), False), ((37, 17, 256), True)):
        base = (torch.randn(shape, device=dev) * 2 + 0.5).requires_grad_(True)
        x = base[:, :1] if sliced else base
        a = torch.randn(x.shape, device=dev).to(torch.bfloat16).requires_grad_(True)
        gamma = (1 + 0.1 * torch.randn(256, device=dev)).requires_grad_(True)
        beta = (0.1 * torch.randn(256, device=dev)).requires_grad_(True)
        gx, gh = torch.randn(x.shape, device=dev), torch.randn(x.shape, device=dev).to(torch.bfloat16)
        for with_a in (True, False):
            for t in (base, a, gamma, beta):
                t.grad = None
            x_new, h = _AddLayerNorm.apply(x, a if with_a else None, gamma, beta, 1e-5, 0.0)
            if with_a:
                torch.autograd.backward([x_new, h], [gx, gh])
            else:
                h.backward(gh)
            got = [t.grad.clone() if t.grad is not None else None for t in (base, a, gamma, beta)]
            for t in (base, a, gamma, beta):
                t.grad = None
            xr = x + a.float() if with_a else x
            hr = F.layer_norm(xr, (256,), gamma, beta, 1e-5)
            if with_a:
                torch.autograd.backward([xr, hr], [gx, gh.float()])
            else:
                hr.backward(gh.float())
            if with_a:
                assert torch.allclose(x_new, xr, atol=1e-6)
            assert h.dtype == torch.bfloat16 and rel(h, hr) < 3e-3
            assert rel(got[0], base.grad) < 1e-4, (shape, with_a)
            if with_a:
                assert rel(got[1], a.grad) < 4e-3  # bf16 output
            else:
                assert got[1] is None
            assert rel(got[2], gamma.grad) < 1e-4 and rel(got[3], beta.grad) < 1e-4
    # without a branch the node passes x through as its first output: the stream's gradient and the LayerNorm gradient are
    # added inside the backward kernel
    base = torch.randn(5, 17, 256, device=dev).requires_grad_(True)
    gamma = (1 + 0.1 * torch.randn(256, device=dev)).requires_grad_(True)
    beta = (0.1 * torch.randn(256, device=dev)).requires_grad_(True)
    gx, gh = torch.randn(base.shape, device=dev), torch.randn(base.shape, device=dev).to(torch.bfloat16)
    x_out, h = _AddLayerNorm.apply(base, None, gamma, beta, 1e-5, 0.0)
    assert torch.equal(x_out, base)
    torch.autograd.backward([x_out, h], [gx, gh])
    got = base.grad.clone()
    base.grad = None
    F.layer_norm(base, (256,), gamma, beta, 1e-5).backward(gh.float())
    assert rel(got, base.grad + gx) < 1e-5
    # g_x_period: a gradient for every 17th row only (the CLS rows) == the same gradient scattered into zeros
    from src.g2048 import native as nv

    B, S = 300, 17
    xn = torch.randn(B, S, 256, device=dev)
    g_cls = torch.randn(B, 1, 256, device=dev)
    g_full = torch.zeros(B, S, 256, device=dev)
    g_full[:, :1] = g_cls
    gh = torch.randn(B, S, 256, device=dev).to(torch.bfloat16)
    mean, var = xn.mean(-1).reshape(-1).contiguous(), xn.var(-1, unbiased=False).reshape(-1)
    rstd = (var + 1e-5).rsqrt().contiguous()
    outs = []
    for g, period in ((g_full, 1), (g_cls.contiguous(), S)):
        dx, da = torch.empty_like(xn), torch.empty(B, S, 256, device=dev, dtype=torch.bfloat16)
        dp = torch.empty(3, 256, device=dev)
        nv.add_ln_bwd(xn.data_ptr(), 256, g, gh, mean, rstd, gamma.detach(), dx, da, dp, B * S, 0.0, 0, 0, g_x_period=period)
        outs.append((dx, da, dp))
    assert all(torch.equal(a, b) for a, b in zip(outs[0], outs[1]))
    # dropout: x_new - x is either 0 or a/(1-p); the gradient for a is masked the same way
    x = torch.zeros(4096, 17, 256, device=dev)  # so that x_new - x is exact
    a = (torch.randn(4096, 17, 256, device=dev).abs() + 0.01).to(torch.bfloat16).requires_grad_(True)
    gamma, beta = torch.ones(256, device=dev, requires_grad=True), torch.zeros(256, device=dev, requires_grad=True)
    x_new, h = _AddLayerNorm.apply(x, a, gamma, beta, 1e-5, 0.1)
    d = x_new - x
    kept = d != 0
    assert abs(kept.float().mean().item() - 0.9) < 2e-3
    assert torch.allclose(d[kept], (a.detach().float() / 0.9)[kept], atol=1e-5, rtol=1e-5)
    x_new.sum().backward()
    assert torch.equal(a.grad != 0, kept) and torch.allclose(a.grad[kept].float(), torch.full((1,), 1 / 0.9, device=dev), atol=5e-3)
    x_new2, _ = _AddLayerNorm.apply(x, a, gamma, beta, 1e-5, 0.1)
    assert not torch.equal(x_new2, x_new)
    # every column of a row block keeps ~90 %: no structure along rows or columns
    assert (kept.float().mean((0, 1)) - 0.9).abs().max() < 0.01 and (kept.float().mean(2) - 0.9).abs().max() < 0.1


def test_colsum_matches_torch(dev):
    """g2048_colsum (bias gradients): f32 column sums of bf16/f32 [T, N], strided rows, ragged T; bit-reproducible."""
    from src.g2048 import native as nv

    torch.manual_seed(2)
    for T, N in ((1, 4), (7, 256), (2048, 512), (34816, 768), (34816, 1024), (33000, 256), (5, 1024)):
        for dt in (torch.bfloat16, torch.float32):
            x = torch.randn(T, N, device=dev).to(dt)
            ref = x.double().sum(0)
            got = nv.colsum(x)
            assert got.dtype == torch.float32 and got.shape == (N,)
            assert torch.allclose(got.double(), ref, rtol=2e-5, atol=2e-4 * max(1.0, T ** 0.5))
            assert torch.equal(got, nv.colsum(x))
    big = torch.randn(3000, 17, 256, device=dev)
    assert torch.allclose(nv.colsum(big[:, 0]), big[:, 0].sum(0), rtol=1e-4, atol=1e-3)  # row stride 17*256
    with pytest.raises(nv.NativeError):
        nv.colsum(torch.zeros(4, 6, device=dev))  # N not a multiple of 4
    with pytest.raises(nv.NativeError):
        nv.colsum(torch.zeros(4, 8))  # host tensor: no CPU path


def _ppo_minibatches(tr, dev, n, M):
    data = tr.rollout_buffer.device_data(dev)
    ds = PPODataset(data, gamma=tr.gamma, lambda_gae=tr.lambda_gae, max_samples_per_epoch=n * M + M, shuffle_on_reset=False)
    return list(DeviceBatches(ds, M, drop_last=True).epoch())[:n]


def test_hip_graph_update_matches_eager(dev, tmp_path):
    """The hipGraph-replayed forward+loss+backward gives the eager gradients on batches OTHER than the captured one
    (dropout off so both are deterministic).  Guards the at::sum-in-a-graph hazard: every bias gradient and the
    CLS-token gradient must come out of g2048_colsum."""
    from src.ppo.ppo_trainer import _GraphedFwdBwd

    torch.manual_seed(0)
    agent = PPOAgent(d_model=256, nhead=8, num_layers=2, dim_feedforward=512, hidden_dim=256, dropout=0.0, reduction="cls")
    tr = PPOTrainer(agent, BatchRunner(0, device=dev), RolloutBuffer(31, 16, 4), dict(OPTIM), max_steps=1000, device=dev,
                    rollout_amp=True, log_dir=str(tmp_path), max_samples_per_epoch=100000)
    assert tr.use_hip_graph  # default for a bf16 update of a graph-safe agent
    tr.collect_rollouts(512, 1)
    M = 2048
    batches = _ppo_minibatches(tr, dev, 3, M)
    assert len(batches) == 3
    agent.train()
    eager = []
    for b in batches:
        stats, _ = tr._loss_backward(*tr._unpack_batch(b, packed=True))  # the same code path, launched eagerly
        eager.append(([p.grad.clone() for p in agent.parameters()], stats[3].item()))
    obs, actions, masks, old_lp, adv, ret = tr._unpack_batch(batches[0], packed=True)
    gr = _GraphedFwdBwd(tr, M, dict(obs=obs, actions=actions, masks=masks, old_lp=old_lp, adv=adv, ret=ret))
    names = [n for n, _ in agent.named_parameters()]
    for i in (1, 2, 0, 1):
        obs, actions, masks, old_lp, adv, ret = tr._unpack_batch(batches[i], packed=True)
        stats, _ = gr.run(dict(obs=obs, actions=actions, masks=masks, old_lp=old_lp, adv=adv, ret=ret))
        assert abs(stats[3].item() - eager[i][1]) < 1e-6
        for n, p, ge in zip(names, agent.parameters(), eager[i][0]):
            err = ((p.grad - ge).norm() / ge.norm().clamp_min(1e-20)).item()
            # same kernels either way; only the float atomics of the LayerNorm weight gradients reorder
            assert torch.isfinite(p.grad).all() and err < 1e-3, (i, n, err)


def test_hip_graph_dropout_draws_new_masks_per_replay(dev):
    """Dropout launches captured in a hipGraph read graph_seed_state at run time: each replay (after the word moved)
    draws a new mask, the backward reuses the forward's mask, and call sites inside one graph differ."""
    from src.ppo.hip_ops import _AddLayerNorm, _AttnPacked, graph_seed_state

    torch.manual_seed(3)
    x = torch.zeros(256, 17, 256, device=dev)
    a = (torch.rand(256, 17, 256, device=dev) + 0.5).to(torch.bfloat16).requires_grad_(True)
    qkv = torch.randn(256, 17, 768, device=dev).to(torch.bfloat16).requires_grad_(True)
    gamma, beta = torch.ones(256, device=dev, requires_grad=True), torch.zeros(256, device=dev, requires_grad=True)
    word = graph_seed_state(dev)

    def fwd_bwd():
        x1, _ = _AddLayerNorm.apply(x, a, gamma, beta, 1e-5, 0.25)
        x2, _ = _AddLayerNorm.apply(x, a, gamma, beta, 1e-5, 0.25)
        o = _AttnPacked.apply(qkv, 8, 0.25)
        (da,) = torch.autograd.grad(x1.sum(), a)
        return x1, x2, o, da

    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        fwd_bwd()
    torch.cuda.current_stream().wait_stream(side)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        word.add_(1)
        outs = fwd_bwd()
    seen = []
    for _ in range(3):
        g.replay()
        x1, x2, o, da = [t.clone() for t in outs]
        assert torch.equal(x1 != 0, da != 0)  # backward mask == forward mask
        assert not torch.equal(x1 != 0, x2 != 0)  # two call sites, two masks
        assert abs((x1 != 0).float().mean().item() - 0.75) < 5e-3
        seen.append((x1, o))
    assert not torch.equal(seen[0][0], seen[1][0]) and not torch.equal(seen[1][0], seen[2][0])
    assert not torch.equal(seen[0][1], seen[1][1])


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("use_mask", [True, False])
def test_fused_ppo_loss_matches_torch(dev, dtype, use_mask):
    """g2048_ppo_loss vs the torch restatement of the reference loss (PPOTrainer._compute_ppo_loss): per-sample
    log-probs, the five means and the gradients for logits and values; ratios inside, outside and next to the ends of the clip
    interval, zero advantages (torch.min ties), masked actions, ragged M."""
    from src.g2048 import native as nv

    torch.manual_seed(5)
    for M in (1, 7, 2048, 5000):
        logits = (torch.randn(M, 4, device=dev) * 2).to(dtype)
        values = torch.randn(M, device=dev).to(dtype)
        bits = torch.randint(1, 16, (M,), device=dev, dtype=torch.uint8)
        mask = (bits.unsqueeze(-1) & torch.tensor([1, 2, 4, 8], dtype=torch.uint8, device=dev)) != 0
        actions = torch.multinomial(mask.float(), 1).squeeze(1)  # always a legal action
        adv, ret = torch.randn(M, device=dev), torch.randn(M, device=dev)
        adv[::5] = 0.0
        l32 = logits.float().requires_grad_(True)
        v32 = values.float().requires_grad_(True)
        z = l32 - 1e8 * (1 - mask.float()) if use_mask else l32
        d = torch.distributions.Categorical(logits=z, validate_args=False)
        new_lp = d.log_prob(actions)
        old_lp = (new_lp.detach() + 0.3 * torch.randn(M, device=dev)).contiguous()
        old_lp[1::7] = new_lp.detach()[1::7]  # ratio == 1: inside the clip range, surr1 == surr2
        old_lp[2::11] = new_lp.detach()[2::11] - float(np.log(1.2 * 1.001))  # ratio just outside the clip range
        old_lp[3::11] = new_lp.detach()[3::11] - float(np.log(1.2 * 0.999))  # ... and just inside
        ratio = torch.exp(new_lp - old_lp)
        pl = -torch.min(ratio * adv, torch.clamp(ratio, 0.8, 1.2) * adv)
        vl = (v32 - ret) ** 2
        el = -d.entropy()
        total = (pl + 0.5 * vl + 0.01 * el).mean()
        total.backward()
        got_lp, sums, dl, dv = nv.ppo_loss(logits, values, actions.to(torch.uint8), bits if use_mask else None, old_lp, adv, ret,
                                           0.2, 0.5, 0.01)
        # north_star: losses within 1e-5 of the fp32 path (f32 inputs); bf16 inputs are rounded before either side sees them
        tol = dict(rtol=1e-5, atol=1e-5)
        assert torch.allclose(got_lp, new_lp.detach(), **tol)
        want = torch.stack([pl.mean(), vl.mean(), el.mean(), total, (old_lp - new_lp).mean()]).detach()
        assert torch.allclose(sums, want, **tol), (M, sums, want)
        assert dl.dtype == dtype and dv.dtype == dtype and dl.shape == logits.shape
        gtol = dict(rtol=2e-2, atol=2e-6) if dtype == torch.bfloat16 else dict(rtol=1e-4, atol=1e-8)
        assert torch.allclose(dl.float(), l32.grad, **gtol), (M, (dl.float() - l32.grad).abs().max())
        assert torch.allclose(dv.float(), v32.grad, **gtol)
        assert torch.equal(sums, nv.ppo_loss(logits, values, actions.to(torch.uint8), bits if use_mask else None, old_lp, adv,
                                             ret, 0.2, 0.5, 0.01)[1])  # fixed summation order
        # the running f64 sums the trainer logs from: exactly what `acc += sums.double()` gives, call after call
        run = torch.full((5,), 2.0, dtype=torch.float64, device=dev)
        for _ in range(3):
            nv.ppo_loss(logits, values, actions.to(torch.uint8), bits if use_mask else None, old_lp, adv, ret, 0.2, 0.5, 0.01,
                        running=run)
        assert torch.equal(run, 2.0 + sums.double() + sums.double() + sums.double())


def test_relu_dropout_kernels(dev):
    """g2048_relu_dropout_fwd/bwd: exact ReLU at p = 0; with dropout the kept share of the active units, the 1/(1-p)
    scale and a fresh mask per call; backward = dy/(1-p) where y != 0, plus its column sums (the bias gradient)."""
    from src.g2048 import native as nv

    torch.manual_seed(7)
    for T, Fdim in ((1, 8), (37, 64), (4097, 1024), (130, 2048)):
        x = torch.randn(T, Fdim, device=dev).to(torch.bfloat16)
        y = torch.empty_like(x)
        nv.relu_dropout_fwd(x, y, 0.0, 0)
        assert torch.equal(y, torch.relu(x))
        dy = torch.randn(T, Fdim, device=dev).to(torch.bfloat16)
        dx, db = torch.empty_like(dy), torch.empty(Fdim, device=dev)
        nv.relu_dropout_bwd(dy, y, dx, db, 0.0)
        want = torch.where(y != 0, dy, torch.zeros_like(dy))
        assert torch.equal(dx, want)
        assert torch.allclose(db, want.float().sum(0), rtol=1e-4, atol=1e-3)
    x = (torch.randn(4096, 1024, device=dev)).to(torch.bfloat16)
    y1, y2 = torch.empty_like(x), torch.empty_like(x)
    nv.relu_dropout_fwd(x, y1, 0.1, 123)
    nv.relu_dropout_fwd(x, y2, 0.1, 124)
    active = x > 0
    kept = (y1 != 0)
    assert not (kept & ~active).any()
    assert abs(kept[active].float().mean().item() - 0.9) < 2e-3 and not torch.equal(y1, y2)
    assert torch.allclose(y1[kept].float(), (x[kept].float() / 0.9), rtol=1e-2)
    assert (kept.float().sum(0) / active.float().sum(0) - 0.9).abs().max() < 0.05  # no structure along columns
    dy = torch.randn_like(x)
    dx, db = torch.empty_like(dy), torch.empty(1024, device=dev)
    nv.relu_dropout_bwd(dy, y1, dx, db, 0.1)
    want = torch.where(kept, (dy.float() / 0.9).to(torch.bfloat16), torch.zeros_like(dy))
    assert (dx.float() - want.float()).abs().max() <= 2.0 ** -7 * want.float().abs().max()  # 1 bf16 ulp (rounding of 1/(1-p))
    assert torch.allclose(db, dx.float().sum(0), rtol=1e-4, atol=1e-2)
    nv.relu_dropout_bwd(dy, y1, dy, db, 0.1)  # in place
    assert torch.equal(dy, dx)


def test_fused_linear_blocks_match_torch(dev):
    """_LinearAddLayerNorm and _LinearReluDropout (dropout off) vs the PyTorch composition under bf16 autocast semantics:
    outputs and every gradient, incl. the bias gradients that come out of the add+LN / activation backward kernels and
    the strided [B, 1, 256] residual slice of the CLS-only layer."""
    import torch.nn.functional as F

    from src.ppo.hip_ops import _LinearAddLayerNorm, _LinearReluDropout

    rel = lambda a, b: ((a.float() - b.float()).norm() / b.float().norm().clamp_min(1e-12)).item()
    torch.manual_seed(11)
    for B, S, K, sliced in ((64, 17, 256, False), ((2048, 17, 1024, False)), (300, 1, 256, True)):
        u = torch.randn(B, S, K, device=dev).to(torch.bfloat16).requires_grad_(True)
        w = (torch.randn(256, K, device=dev) / K ** 0.5).requires_grad_(True)
        b = (0.1 * torch.randn(256, device=dev)).requires_grad_(True)
        base = torch.randn(B, 17 if sliced else S, 256, device=dev).requires_grad_(True)
        x = base[:, :1] if sliced else base
        gamma = (1 + 0.1 * torch.randn(256, device=dev)).requires_grad_(True)
        beta = (0.1 * torch.randn(256, device=dev)).requires_grad_(True)
        gx, gh = torch.randn(x.shape, device=dev), torch.randn(x.shape, device=dev).to(torch.bfloat16)
        leaves = (u, w, b, base, gamma, beta)
        x_new, h = _LinearAddLayerNorm.apply(u, w, b, w.detach().to(torch.bfloat16), b.detach().to(torch.bfloat16), x, gamma,
                                             beta, 1e-5, 0.0)
        torch.autograd.backward([x_new, h], [gx, gh])
        got = [t.grad.clone() for t in leaves]
        for t in leaves:
            t.grad = None
        a = F.linear(u, w.to(torch.bfloat16), b.to(torch.bfloat16))
        xr = x + a.float()
        hr = F.layer_norm(xr, (256,), gamma, beta, 1e-5)
        torch.autograd.backward([xr, hr], [gx, gh.float()])
        assert torch.allclose(x_new, xr, atol=1e-5) and rel(h, hr) < 3e-3
        for name, g, t in zip(("u", "w", "b", "x", "gamma", "beta"), got, leaves):
            assert rel(g, t.grad) < (8e-3 if name in ("u", "w", "b") else 2e-4), (name, rel(g, t.grad))
    for T, K, Fdim in ((64, 256, 1024), (34816, 256, 1024), (300, 256, 512)):
        hh = torch.randn(T, K, device=dev).to(torch.bfloat16).requires_grad_(True)
        w = (torch.randn(Fdim, K, device=dev) / K ** 0.5).requires_grad_(True)
        b = (0.1 * torch.randn(Fdim, device=dev)).requires_grad_(True)
        gy = torch.randn(T, Fdim, device=dev).to(torch.bfloat16)
        y = _LinearReluDropout.apply(hh, w, b, w.detach().to(torch.bfloat16), b.detach().to(torch.bfloat16), 0.0)
        y.backward(gy)
        got = [t.grad.clone() for t in (hh, w, b)]
        for t in (hh, w, b):
            t.grad = None
        yr = torch.relu(F.linear(hh, w.to(torch.bfloat16), b.to(torch.bfloat16)))
        yr.backward(gy)
        assert rel(y, yr) < 4e-3  # the wide case runs through g2048_linear_bf16 (f32 bias, one rounding)
        for name, g, t in zip(("h", "w", "b"), got, (hh, w, b)):
            assert rel(g, t.grad) < 2e-2, (name, rel(g, t.grad))  # units with z ~ 0 may land on either side of the ReLU


def test_fused_ffn_kernels(dev):
    """g2048_linear_relu_dropout_bf16 and g2048_linear_mask_bwd_bf16 (GEMM + activation epilogues) vs f32 references: ragged
    T, dropout off (values) and on (keep rate, scaling, no structure, new masks per seed), masked gradient and the bias
    gradient (column sums of the bf16 result, bit-reproducible)."""
    import torch.nn.functional as F

    from src.g2048 import native as nv

    torch.manual_seed(23)
    rel = lambda a, b: ((a.float() - b.float()).norm() / b.float().norm()).item()
    for T in (1, 130, 4096, 34816 + 3):
        for K, N in ((256, 1024), (256, 512), (128, 128), (256, 384)):  # 256-wide slices (8 waves) and 128-wide (4 waves)
            x = torch.randn(T, K, device=dev).to(torch.bfloat16)
            w = (torch.randn(N, K, device=dev) / K ** 0.5).to(torch.bfloat16)
            b = 0.2 * torch.randn(N, device=dev)
            ref = torch.relu(F.linear(x.float(), w.float(), b))
            y = nv.linear_relu_dropout(x, w, b, 0.0)
            assert y.shape == (T, N) and rel(y, ref) < 4e-3, (T, K, N, rel(y, ref))
            assert ((y == 0) == (ref.to(torch.bfloat16) == 0)).float().mean() > 0.999  # pre-activations next to 0 may differ
            if T >= 4096:
                y1, y2 = nv.linear_relu_dropout(x, w, b, 0.1, seed=7), nv.linear_relu_dropout(x, w, b, 0.1, seed=8)
                assert torch.equal(y1, nv.linear_relu_dropout(x, w, b, 0.1, seed=7)) and not torch.equal(y1, y2)
                active, kept = y != 0, y1 != 0
                assert not (kept & ~active).any()
                assert abs(kept[active].float().mean().item() - 0.9) < 3e-3
                assert torch.allclose(y1[kept].float(), ref[kept] / 0.9, rtol=2e-2, atol=2e-2)
                assert (kept.float().sum(0) / active.float().sum(0).clamp_min(1) - 0.9).abs().max() < 0.06
                assert (kept.float().sum(1) / active.float().sum(1).clamp_min(1) - 0.9).abs().mean() < 0.05
            # backward: dz = (dy @ W2) / keep where y_saved != 0
            dy = torch.randn(T, K, device=dev).to(torch.bfloat16)
            w2t = (torch.randn(N, K, device=dev) / K ** 0.5).to(torch.bfloat16)  # = linear2.weight^T, [N, K]
            for p in (0.0, 0.1):
                ys, bits = nv.linear_relu_dropout(x, w, b, p, seed=9, want_mask=True)
                assert torch.equal(ys, nv.linear_relu_dropout(x, w, b, p, seed=9))  # the mask output changes nothing else
                dz, db = nv.linear_mask_bwd(dy, w2t, bits, p)
                want = torch.where(ys != 0, (dy.float() @ w2t.float().t()) / (1 - p), torch.zeros((), device=dev))
                assert rel(dz, want) < 4e-3, (T, K, N, p, rel(dz, want))
                assert not (dz[ys == 0] != 0).any()
                assert torch.allclose(db, dz.float().sum(0), rtol=1e-4, atol=1e-4 * float(dz.float().abs().sum(0).max()) + 1e-6)
                dz2, db2 = nv.linear_mask_bwd(dy, w2t, bits, p)
                assert torch.equal(dz, dz2) and torch.equal(db, db2)
    with pytest.raises(nv.NativeError):
        nv.linear_relu_dropout(torch.zeros(8, 512, device=dev, dtype=torch.bfloat16),
                               torch.zeros(128, 512, device=dev, dtype=torch.bfloat16), torch.zeros(128, device=dev), 0.0)


def test_linked_ffn_block_matches_torch(dev):
    """linear1 -> ReLU -> linear2 -> add -> LayerNorm through _LinearReluDropout + _LinearAddLayerNorm joined by an FFNLink
    (the fused forward epilogue and the masked-gradient GEMM) vs the PyTorch composition: outputs and all gradients."""
    import torch.nn.functional as F

    from src.ppo.hip_ops import FFNLink, _LinearAddLayerNorm, _LinearReluDropout

    rel = lambda a, b: ((a.float() - b.float()).norm() / b.float().norm().clamp_min(1e-12)).item()
    torch.manual_seed(29)
    B, S = 300, 17  # 5100 tokens: the fused kernels engage
    h = torch.randn(B, S, 256, device=dev).to(torch.bfloat16).requires_grad_(True)
    x = torch.randn(B, S, 256, device=dev).requires_grad_(True)
    w1 = (torch.randn(1024, 256, device=dev) / 16).requires_grad_(True)
    b1 = (0.1 * torch.randn(1024, device=dev)).requires_grad_(True)
    w2 = (torch.randn(256, 1024, device=dev) / 32).requires_grad_(True)
    b2 = (0.1 * torch.randn(256, device=dev)).requires_grad_(True)
    gamma = (1 + 0.1 * torch.randn(256, device=dev)).requires_grad_(True)
    beta = (0.1 * torch.randn(256, device=dev)).requires_grad_(True)
    gx, gh = torch.randn(x.shape, device=dev), torch.randn(x.shape, device=dev).to(torch.bfloat16)
    leaves = (h, x, w1, b1, w2, b2, gamma, beta)
    bf = lambda t: t.detach().to(torch.bfloat16)
    link = FFNLink(0.0)
    f = _LinearReluDropout.apply(h, w1, b1, bf(w1), bf(b1), 0.0, link)
    x_new, hn = _LinearAddLayerNorm.apply(f, w2, b2, bf(w2), bf(b2), x, gamma, beta, 1e-5, 0.0, bf(w2).t().contiguous(), link)
    torch.autograd.backward([x_new, hn], [gx, gh])
    assert link.db is None and not link.masked  # consumed by the first node's backward
    got = [t.grad.clone() for t in leaves]
    for t in leaves:
        t.grad = None
    fr = torch.relu(F.linear(h, w1.to(torch.bfloat16), b1.to(torch.bfloat16)))
    xr = x + F.linear(fr, w2.to(torch.bfloat16), b2.to(torch.bfloat16)).float()
    hr = F.layer_norm(xr, (256,), gamma, beta, 1e-5)
    torch.autograd.backward([xr, hr], [gx, gh.float()])
    assert rel(f, fr) < 4e-3 and rel(x_new, xr) < 4e-3 and rel(hn, hr) < 6e-3
    for name, g, t in zip(("h", "x", "w1", "b1", "w2", "b2", "gamma", "beta"), got, leaves):
        assert rel(g, t.grad) < 2e-2, (name, rel(g, t.grad))
    # with dropout: the gradient of h vanishes through dropped units -- same mask forward and backward
    link = FFNLink(0.5)
    f = _LinearReluDropout.apply(h, w1, b1, bf(w1), bf(b1), 0.5, link)
    x_new, hn = _LinearAddLayerNorm.apply(f, w2, b2, bf(w2), bf(b2), x, gamma, beta, 1e-5, 0.0, bf(w2).t().contiguous(), link)
    for t in leaves:
        t.grad = None
    torch.autograd.backward([x_new, hn], [gx, gh])
    kept = (f != 0).float()
    assert abs(kept.mean().item() / (fr != 0).float().mean().item() - 0.5) < 0.01
    active_cols = kept.reshape(-1, 1024).sum(0) > 0
    assert (b1.grad[~active_cols] == 0).all() and torch.isfinite(w1.grad).all() and torch.isfinite(h.grad).all()


def test_linear_bf16_matches_torch(dev):
    """g2048_linear_bf16 (MFMA; K <= 256: weights in registers, X through LDS; K > 256: weights streamed through LDS) vs an f32 reference: as close as hipBLASLt's bf16 GEMM,
    ragged T, all (K, N) of the update, strided inputs/weights (views), with and without bias."""
    import torch.nn.functional as F

    from src.g2048 import native as nv

    torch.manual_seed(13)
    rel = lambda a, b: ((a.float() - b.float()).norm() / b.float().norm()).item()
    for T in (1, 31, 128, 129, 2048, 34816 + 5):
        for K, N in ((256, 256), (256, 768), (256, 1024), (1024, 256), (768, 256), (128, 128), (512, 384), (256, 384), (128, 256)):
            x = torch.randn(T, K, device=dev).to(torch.bfloat16)
            w = (torch.randn(N, K, device=dev) / K ** 0.5).to(torch.bfloat16)
            b = torch.randn(N, device=dev)
            ref = F.linear(x.float(), w.float(), b)
            y = nv.linear_bf16(x, w, b)
            assert y.shape == (T, N) and y.dtype == torch.bfloat16
            assert rel(y, ref) < 4e-3, (T, K, N, rel(y, ref))
            assert (y.float() - ref).abs().max() <= 2.0 ** -7 * ref.abs().max() + 1e-2
            y0 = nv.linear_bf16(x, w, None)
            assert rel(y0, F.linear(x.float(), w.float())) < 4e-3
    # views: a column slice of a wider activation, a row slice of a taller weight
    xx = torch.randn(1000, 768, device=dev).to(torch.bfloat16)
    ww = (torch.randn(768, 256, device=dev) / 16).to(torch.bfloat16)
    y = nv.linear_bf16(xx[:, 256:512], ww[256:512], None)
    assert rel(y, xx[:, 256:512].float() @ ww[256:512].float().t()) < 4e-3
    with pytest.raises(nv.NativeError):
        nv.linear_bf16(torch.zeros(8, 100, device=dev, dtype=torch.bfloat16), torch.zeros(128, 100, device=dev, dtype=torch.bfloat16))


def test_embed_with_first_layernorm_equals_the_two_launches(dev, monkeypatch):
    """g2048_embed_ln_fwd = g2048_embed_fwd followed by g2048_add_ln_fwd over the same rows, bit for bit (tokens, normalised bf16 rows,
    means, rstds; with and without the positional dropout), and the agent's update forward / backward with the embedding kernel
    normalising for layers[0].norm1 (LNPre) equals the run with the separate launch (G2048_EMBED_LN=0) bit for bit."""
    from src.g2048 import native as nv
    from src.ppo import PPOAgent

    torch.manual_seed(23)
    for M, p in ((1, 0.0), (300, 0.0), (2048, 0.1)):
        boards = torch.randint(0, 18, (M, 16), device=dev, dtype=torch.uint8)
        w, pe, cls = torch.randn(256, 31, device=dev), torch.randn(16, 256, device=dev), torch.randn(256, device=dev)
        gamma, beta = torch.rand(256, device=dev) + 0.5, torch.randn(256, device=dev) * 0.1
        x_a, x_b = torch.empty(M, 17, 256, device=dev), torch.empty(M, 17, 256, device=dev)
        h_a, h_b = (torch.empty(M, 17, 256, device=dev, dtype=torch.bfloat16) for _ in range(2))
        st_a, st_b = torch.empty(2, M * 17, device=dev), torch.empty(2, M * 17, device=dev)
        nv.embed_fwd(boards, w, pe, cls, x_a, p, 1234, 0, ln=(gamma, beta, 1e-5, h_a, st_a[0], st_a[1]))
        nv.embed_fwd(boards, w, pe, cls, x_b, p, 1234, 0)
        nv.add_ln_fwd(x_b.data_ptr(), 256, None, gamma, beta, None, h_b, st_b[0], st_b[1], M * 17, 1e-5, 0.0, 0, 0)
        assert torch.equal(x_a, x_b) and torch.equal(h_a, h_b) and torch.equal(st_a, st_b), (M, p)
    with pytest.raises(nv.NativeError):
        nv.embed_fwd(boards, w, pe, cls, x_a, 0.0, 0, 0, ln=(gamma.double(), beta, 1e-5, h_a, st_a[0], st_a[1]))

    torch.manual_seed(5)
    agent = PPOAgent(dropout=0.0, reduction="cls").to(dev).train()  # (no dropout: every launch of the two runs is deterministic)
    boards = torch.randint(0, 12, (256, 16), device=dev, dtype=torch.uint8)

    def run(flag):
        monkeypatch.setenv("G2048_EMBED_LN", flag)
        calls = []
        real = nv.add_ln_fwd
        monkeypatch.setattr(nv, "add_ln_fwd", lambda *a, **k: (calls.append(1), real(*a, **k))[1])
        agent.zero_grad(set_to_none=True)
        with torch.autocast("cuda", dtype=torch.bfloat16):
            lo, va = agent(boards, None)
        (lo.float().square().sum() + va.float().sum()).backward()
        monkeypatch.setattr(nv, "add_ln_fwd", real)
        return lo.detach().clone(), va.detach().clone(), [p.grad.clone() for p in agent.parameters()], len(calls)

    lo1, va1, g1, n1 = run("1")
    lo0, va0, g0, n0 = run("0")
    assert n0 == n1 + 1, (n0, n1)  # one g2048_add_ln_fwd launch less
    assert torch.equal(lo1, lo0) and torch.equal(va1, va0)
    assert all(torch.equal(a, b) for a, b in zip(g1, g0))


def test_embed_boards_matches_torch(dev):
    """g2048_embed_fwd/bwd (gather + segmented sum) vs one-hot Linear + positional add + CLS concat in PyTorch: tokens and
    the gradients of the embedding weight and the CLS token; with dropout only board tokens are dropped, the backward
    reuses the mask, and the sums are bit-reproducible."""
    from src.ppo.hip_ops import _EmbedBoards

    torch.manual_seed(17)
    for M in (1, 5, 2048, 2049):
        boards = torch.randint(0, 31, (M, 16), device=dev, dtype=torch.uint8)
        w = torch.randn(256, 31, device=dev, requires_grad=True)
        pe = torch.randn(16, 256, device=dev)
        cls = torch.randn(1, 1, 256, device=dev, requires_grad=True)
        g = torch.randn(M, 17, 256, device=dev)
        x0 = _EmbedBoards.apply(boards, w, pe, cls, 0.0)
        x0.backward(g)
        got = (w.grad.clone(), cls.grad.clone())
        w.grad = cls.grad = None
        onehot = torch.nn.functional.one_hot(boards.long(), 31).float()
        ref = torch.cat([cls.expand(M, -1, -1), onehot @ w.t() + pe], dim=1)
        ref.backward(g)
        assert torch.equal(x0, ref.detach()) or torch.allclose(x0, ref, atol=1e-6)
        assert torch.allclose(got[0], w.grad, rtol=1e-4, atol=1e-3 * max(1.0, M ** 0.5))
        assert torch.allclose(got[1], cls.grad, rtol=1e-4, atol=1e-3 * max(1.0, M ** 0.5))
        again = _EmbedBoards.apply(boards, w, pe, cls, 0.0)
        w.grad = cls.grad = None
        again.backward(g)
        assert torch.equal(w.grad, got[0]) and torch.equal(cls.grad, got[1])
        w.grad = cls.grad = None
        # with a GradSink both gradients come out of g2048_reduce_jobs (the weight's stored transposed, [256][31])
        from src.ppo.hip_ops import GradSink, grad_sink

        tw, tc = torch.full_like(w, float("nan")), torch.full_like(cls, float("nan"))
        sunk = _EmbedBoards.apply(boards, w, pe, cls, 0.0)
        with grad_sink(GradSink({id(w): tw, id(cls): tc})):
            sunk.backward(g)
        assert w.grad is None and cls.grad is None
        assert torch.allclose(tw, got[0], rtol=1e-5, atol=1e-4 * max(1.0, M ** 0.5))
        assert torch.allclose(tc, got[1], rtol=1e-5, atol=1e-4 * max(1.0, M ** 0.5))
    boards = torch.randint(0, 31, (4096, 16), device=dev, dtype=torch.uint8)
    w = torch.randn(256, 31, device=dev, requires_grad=True)
    pe = torch.zeros(16, 256, device=dev)
    cls = torch.randn(1, 1, 256, device=dev, requires_grad=True)
    x0 = _EmbedBoards.apply(boards, w, pe, cls, 0.1)
    full = _EmbedBoards.apply(boards, w, pe, cls, 0.0)
    assert torch.equal(x0[:, 0], full[:, 0])  # CLS row is never dropped
    kept = x0[:, 1:] != 0
    assert abs(kept.float().mean().item() - 0.9) < 2e-3
    assert torch.allclose(x0[:, 1:][kept], (full[:, 1:] / 0.9)[kept], rtol=1e-6)
    x0.sum().backward()
    counts = torch.zeros(31, device=dev)
    for e in range(31):
        counts[e] = (kept & (boards == e).unsqueeze(-1)).sum() / 256.0
    assert torch.allclose(w.grad.mean(0), counts / 0.9, rtol=1e-2, atol=1e-2)  # masked the same way as the forward
    assert torch.allclose(cls.grad.reshape(-1), torch.full((256,), 4096.0, device=dev))


def test_gather_minibatch_matches_index_select(dev):
    """g2048_gather_minibatch == six index_selects; also straight into pre-allocated (static) outputs."""
    from src.g2048 import native as nv

    torch.manual_seed(19)
    N = 100003
    boards = torch.randint(0, 16, (N, 16), device=dev, dtype=torch.uint8)
    actions = torch.randint(0, 4, (N,), device=dev, dtype=torch.uint8)
    masks = torch.randint(1, 16, (N,), device=dev, dtype=torch.uint8)
    logp, adv, ret = (torch.randn(N, device=dev) for _ in range(3))
    for M in (1, 255, 2048, 5001):
        idx = torch.randint(0, N, (M,), device=dev)
        got = nv.gather_minibatch(idx, boards, actions, masks, logp, adv, ret)
        for k, src in (("obs", boards), ("actions", actions), ("masks", masks), ("old_lp", logp), ("adv", adv), ("ret", ret)):
            assert torch.equal(got[k], src.index_select(0, idx)), k
        again = nv.gather_minibatch(idx.flip(0), boards, actions, masks, logp, adv, ret, out=got)
        assert again is got and torch.equal(got["obs"], boards.index_select(0, idx.flip(0)))


def test_linear_relu_node_matches_torch(dev):
    """``_LinearRelu`` (hidden layers of the heads): forward == relu(F.linear) in bf16 bit for bit; backward == autograd of
    the same expression, with a GradSink (partials summed by g2048_reduce_jobs) and without."""
    from src.ppo.hip_ops import GradSink, _LinearRelu, grad_sink

    torch.manual_seed(11)
    for T, K, N in ((2048, 256, 512), (2048, 512, 512), (300, 64, 64)):
        w = (torch.randn(N, K, device=dev) / K ** 0.5).requires_grad_()
        b = (torch.randn(N, device=dev) * 0.1).requires_grad_()
        x = torch.randn(T, K, device=dev).to(torch.bfloat16).requires_grad_()
        wb, bb = w.detach().to(torch.bfloat16), b.detach().to(torch.bfloat16)
        g = torch.randn(T, N, device=dev).to(torch.bfloat16)
        assert _LinearRelu.ok(x, w, b, wb, bb)
        xr, wr, br = x.detach().clone().requires_grad_(), wb.clone().requires_grad_(), bb.clone().requires_grad_()
        ref = torch.relu(torch.nn.functional.linear(xr, wr, br))
        ref.backward(g)
        with torch.autocast("cuda", dtype=torch.bfloat16):
            y = _LinearRelu.apply(x, w, b, wb, bb)
        assert y.dtype == torch.bfloat16 and torch.equal(y, ref.detach())
        y.backward(g)
        assert torch.allclose(x.grad.float(), xr.grad.float(), rtol=2e-2, atol=2e-2)
        # f32 weight / bias gradients: the bf16 autograd reference rounds its own to bf16
        assert (w.grad - wr.grad.float()).norm() / wr.grad.float().norm() < 6e-3
        assert (b.grad - br.grad.float()).norm() / br.grad.float().norm() < 6e-3
        tw, tb = torch.zeros_like(w), torch.zeros_like(b)
        sink = GradSink({id(w): tw, id(b): tb})
        x2 = x.detach().clone().requires_grad_()
        with torch.autocast("cuda", dtype=torch.bfloat16):
            y2 = _LinearRelu.apply(x2, w, b, wb, bb)
        w.grad = b.grad = None
        with grad_sink(sink):
            y2.backward(g)
        assert w.grad is None and b.grad is None and sink.written == {id(w), id(b)}
        assert torch.equal(x2.grad, x.grad)
        assert (tw - wr.grad.float()).norm() / wr.grad.float().norm() < 6e-3
        assert (tb - br.grad.float()).norm() / br.grad.float().norm() < 6e-3


def test_graphed_rollout_forward_of_the_mlp_policy(dev, tmp_path, monkeypatch):
    """The MLP policy's rollout forward replayed from a hipGraph (TorchActionFunction(graph_cache=...)): same logits/values
    as the eager forward, in-place parameter updates are seen by the replay, the engine then runs without compaction, and
    the trainer's collect uses it (one capture, reused by the next collect)."""
    torch.manual_seed(5)
    agent = MLPAgent().to(dev)
    cache = {}
    fn = TorchActionFunction(agent, use_mask=True, device=dev, amp_dtype=torch.bfloat16, graph_cache=cache)
    assert fn.compact is False
    boards = torch.randint(0, 12, (4096, 16), dtype=torch.uint8, device=dev)
    ref = TorchActionFunction(agent, use_mask=True, device=dev, amp_dtype=torch.bfloat16)
    assert ref.compact is True
    lg, vl = (t.clone() for t in fn.policy_fn(boards, None))  # (the graph's static outputs: overwritten by the next call)
    assert "fallback" not in cache and len(cache) == 1
    lr, vr = ref.policy_fn(boards, None)
    assert torch.equal(lg, lr) and torch.equal(vl, vr)
    boards2 = torch.randint(0, 12, (4096, 16), dtype=torch.uint8, device=dev)
    with torch.no_grad():
        for p in agent.parameters():
            p.mul_(0.5)
    lg2, vl2 = fn.policy_fn(boards2, None)
    lr2, vr2 = ref.policy_fn(boards2, None)
    assert len(cache) == 1 and torch.equal(lg2, lr2) and torch.equal(vl2, vr2) and not torch.equal(lg2, lg)
    # through the trainer: the capture happens in the first collect and is reused by the second
    monkeypatch.chdir(tmp_path)
    tr = PPOTrainer(MLPAgent(), BatchRunner(0, device=dev), RolloutBuffer(31, 16, 4), OPTIM, max_steps=100, device=dev,
                    rollout_amp=True, use_action_mask=True, max_samples_per_epoch=4096, log_dir=str(tmp_path / "lg"))
    tr.collect_rollouts(512, 1)
    assert len(tr._rollout_graphs) == 1 and "fallback" not in tr._rollout_graphs
    n1 = tr.rollout_buffer.buffer_size
    tr.update_policy(batch_size=256, n_epochs=1)
    tr.collect_rollouts(512, 1)
    assert len(tr._rollout_graphs) == 1 and n1 > 512 and tr.rollout_buffer.buffer_size > 512


def test_linear_add_cast_node_matches_torch(dev):
    """``_LinearAddCast`` (g2048_add_ln_fwd/bwd with gamma NULL): bf16(x + dropout(Linear(u))) and its gradients; p = 0
    against autograd of the same expression, p > 0 against the mask the forward itself drew."""
    from src.ppo.hip_ops import GradSink, _LinearAddCast, grad_sink

    torch.manual_seed(13)
    B, K = 2048, 1024
    w = (torch.randn(256, K, device=dev) / K ** 0.5).requires_grad_()
    b = (torch.randn(256, device=dev) * 0.1).requires_grad_()
    wb, bb = w.detach().to(torch.bfloat16), b.detach().to(torch.bfloat16)
    stream = torch.randn(B, 17, 256, device=dev)
    x = stream[:, :1].detach().requires_grad_()  # [B, 1, 256] with row stride 17 * 256: read in place
    assert not x.is_contiguous() or x.stride(0) == 256
    u = torch.randn(B, 1, K, device=dev).to(torch.bfloat16).requires_grad_()
    g = torch.randn(B, 1, 256, device=dev).to(torch.bfloat16)
    with torch.autocast("cuda", dtype=torch.bfloat16):
        h = _LinearAddCast.apply(u, w, b, wb, bb, x, 0.0)
    ur, xr, wr, br = (t.detach().clone().requires_grad_() for t in (u, x, wb, bb))
    ref = (xr + torch.nn.functional.linear(ur, wr, br).float()).to(torch.bfloat16)
    assert h.dtype == torch.bfloat16 and h.shape == (B, 1, 256) and torch.equal(h, ref)
    h.backward(g)
    ref.backward(g)
    assert torch.equal(x.grad, xr.grad) and torch.allclose(u.grad.float(), ur.grad.float(), rtol=2e-2, atol=2e-2)
    assert (w.grad - wr.grad.float()).norm() / wr.grad.float().norm() < 6e-3
    assert (b.grad - br.grad.float()).norm() / br.grad.float().norm() < 6e-3
    # dropout: the backward re-draws the forward's mask (a >= 3 everywhere, so kept / dropped can be read off the output);
    # with a sink the weight / bias gradients land in the targets
    w2 = (torch.randn(256, K, device=dev) * 0.01 / K ** 0.5).requires_grad_()
    b2 = (4.0 + torch.randn(256, device=dev) * 0.1).requires_grad_()
    wb2, bb2 = w2.detach().to(torch.bfloat16), b2.detach().to(torch.bfloat16)
    tw, tb = torch.zeros_like(w2), torch.zeros_like(b2)
    x2 = torch.zeros(B, 1, 256, device=dev, requires_grad=True)
    u2 = u.detach().clone().requires_grad_()
    with torch.autocast("cuda", dtype=torch.bfloat16):
        h2 = _LinearAddCast.apply(u2, w2, b2, wb2, bb2, x2, 0.25)
    a = torch.nn.functional.linear(u2.detach(), wb2, bb2).float()
    assert a.min() > 3
    kept = h2.float() != 0
    assert abs(kept.float().mean().item() - 0.75) < 0.01
    assert torch.allclose(h2.float()[kept], (a / 0.75)[kept], rtol=1e-2)
    with grad_sink(GradSink({id(w2): tw, id(b2): tb})):
        h2.backward(g)
    assert w2.grad is None and b2.grad is None and torch.equal(x2.grad, g.float())
    da = torch.where(kept, g.float() / 0.75, torch.zeros_like(a)).to(torch.bfloat16)
    du_ref = (da.view(B, 256) @ wb2).view(B, 1, K)
    assert torch.allclose(u2.grad.float(), du_ref.float(), rtol=2e-2, atol=2e-3)
    db_ref = da.float().sum((0, 1))
    assert (tb - db_ref).norm() / db_ref.norm() < 1e-3
    dw_ref = da.view(B, 256).float().t() @ u2.detach().view(B, K).float()
    assert (tw - dw_ref).norm() / dw_ref.norm() < 6e-3


def test_no_garbage_collection_while_a_stream_is_capturing(dev, tmp_path, monkeypatch):
    """Both capture sites (the update's _GraphedFwdBwd and the MLP policy's rollout forward) run under capture.capture():
    with the collector's thresholds far below their defaults not one collection starts while the stream is capturing, although
    earlier trainers with captured graphs are garbage at that moment (the situation of the recorded abort, NOTES.md 3)."""
    import gc

    from src.ppo.capture import CollectionsWhileCapturing

    monkeypatch.chdir(tmp_path)

    def make(agent):
        return PPOTrainer(agent, BatchRunner(0, device=dev), RolloutBuffer(31, 16, 4), OPTIM, max_steps=100, device=dev,
                          rollout_amp=True, use_action_mask=True, max_samples_per_epoch=2048, log_dir=str(tmp_path / "lg"))

    old = gc.get_threshold()
    try:
        with CollectionsWhileCapturing() as seen:
            for _ in range(2):  # the second round's captures start with the first round's trainers as cyclic garbage
                gc.set_threshold(100, 5, 5)  # (default 700, 10, 10: a capture allocates thousands of objects)
                tr = make(MLPAgent())
                tr.collect_rollouts(256, 1)  # captures the rollout forward
                m = tr.update_policy(batch_size=256, n_epochs=1)  # captures the update
                assert m["hip_graph"] and "rollout_graph_fallback" not in m and len(tr._rollout_graphs) == 1
                del tr
        assert seen.total > 0 and seen.during_capture == 0
    finally:
        gc.set_threshold(*old)


def test_dweight_parts_match_f32_reference(dev):
    """g2048_dweight_bf16 (token-major operands staged by LDS-DMA, transposed MFMA operand reads): the sum of the partials against
    dY^T X in f32 - as close as the batched hipBLASLt GEMM it replaces -, every (N, K) of the update + the 128-wide block shape,
    1 / 8 / 16 / 32 slices, column-slice views as operands, and the refusals."""
    from src.g2048 import native as nv

    torch.manual_seed(41)
    rel = lambda a, b: ((a.float() - b.float()).norm() / b.float().norm()).item()
    for T, S in ((64, 1), (1024, 8), (34816, 16), (34816, 32)):
        for N, K in ((1024, 256), (256, 1024), (768, 256), (256, 256), (128, 128), (384, 128)):
            if T * N * K > 34816 * 1024 * 256 // 2 and S == 32 and (N, K) != (1024, 256):
                continue  # one full-size case per slice count is enough
            dy = (torch.randn(T, N, device=dev) / 8).to(torch.bfloat16)
            x = torch.randn(T, K, device=dev).to(torch.bfloat16)
            parts = nv.dweight_parts(dy, x, S)
            assert parts.shape == (S, N, K) and parts.dtype == torch.bfloat16
            ref = dy.float().t() @ x.float()
            assert rel(parts.float().sum(0), ref) < 4e-3, (T, S, N, K, rel(parts.float().sum(0), ref))
            # every partial is the product over its own token slice
            s = S // 2
            rows = slice(s * (T // S), (s + 1) * (T // S))
            assert rel(parts[s], dy[rows].float().t() @ x[rows].float()) < 4e-3
            assert torch.equal(parts, nv.dweight_parts(dy, x, S))  # fixed summation order
            parts2, cs = nv.dweight_parts(dy, x, S, colsum=True)  # the bias gradient's first stage rides along
            assert torch.equal(parts2, parts) and cs.shape == (S, N) and cs.dtype == torch.float32
            assert torch.allclose(cs.sum(0), dy.float().sum(0), rtol=1e-4, atol=1e-4 * float(dy.float().abs().sum(0).max()) + 1e-6)
            assert torch.allclose(cs[s], dy[rows].float().sum(0), rtol=1e-4, atol=1e-4 * float(dy[rows].float().abs().sum(0).max()) + 1e-6)
            if N % 256 == 0 and T <= 1024:  # both block shapes
                for rows in (128, 256):
                    assert rel(nv.dweight_parts(dy, x, S, block_rows=rows).float().sum(0), ref) < 4e-3, (T, S, N, K, rows)
    # several products in one launch (what the GradSink does with a backward pass's weight gradients): bit-identical to the single launches
    jobs = []
    for T, N, K, S in ((34816, 1024, 256, 16), (34816, 256, 256, 32), (2048, 768, 128, 8), (34816, 256, 1024, 16)):
        dy = (torch.randn(T, N, device=dev) / 8).to(torch.bfloat16)
        x = torch.randn(T, K, device=dev).to(torch.bfloat16)
        jobs.append((dy, x, torch.empty(S, N, K, dtype=torch.bfloat16, device=dev),
                     torch.empty(S, N, dtype=torch.float32, device=dev) if N != 768 else None))
    nv.dweight_jobs(jobs)
    for dy, x, parts, cs in jobs:
        want = nv.dweight_parts(dy, x, parts.shape[0], block_rows=128, colsum=cs is not None)
        assert torch.equal(parts, want[0] if cs is not None else want)
        if cs is not None:
            assert torch.equal(cs, want[1])
    # f32 partials (the job's parts_f32 flag; round 4's A/B arm G2048_DWEIGHT_PARTS=f32x8): the same accumulators, stored unrounded
    dy, x, parts16, _ = jobs[0]
    S = 8
    p32, p16 = torch.empty(S, 1024, 256, dtype=torch.float32, device=dev), torch.empty(S, 1024, 256, dtype=torch.bfloat16, device=dev)
    nv.dweight_jobs([(dy, x, p32, None), (dy, x, p16, None)])
    assert torch.equal(p32.to(torch.bfloat16), p16), "bf16 partials must be the rounding of the f32 ones"
    assert rel(p32.sum(0), dy.float().t() @ x.float()) < 1e-5
    # views: column slices of wider activations (leading dimension != width)
    wide_dy = (torch.randn(2048, 768, device=dev) / 8).to(torch.bfloat16)
    wide_x = torch.randn(2048, 512, device=dev).to(torch.bfloat16)
    parts = nv.dweight_parts(wide_dy[:, 256:512], wide_x[:, 128:384], 8)
    assert rel(parts.float().sum(0), wide_dy[:, 256:512].float().t() @ wide_x[:, 128:384].float()) < 4e-3
    assert not nv.dweight_ok(wide_dy[:100], wide_x[:100], 1)       # T not a multiple of 64
    assert not nv.dweight_ok(wide_dy[:, :200], wide_x, 8)          # N not a multiple of 128
    with pytest.raises(nv.NativeError):
        nv.dweight_parts(wide_dy, wide_x, 12)
