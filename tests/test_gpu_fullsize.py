"""BASELINE.json's full sizes on the GPU: direct oracle comparison where the C oracle finishes in seconds, and
size-independent properties (tile-sum conservation, frozen-after-done, shard invariance) beyond that."""
import numpy as np
import pytest
import torch

from oracle import c_oracle as orc
from oracle import g2048_oracle as npo
from src.actions import act_drul, act_randomly
from src.g2048 import native as nv
from src.runs import BatchRunner

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("fn,policy", [(act_randomly, 1), (act_drul, 0)])
def test_65536_boards_whole_episodes_vs_oracle(dev, fn, policy):
    """configs[2] size: every one of the 65 536 episodes ends on the oracle's board after the oracle's number of steps."""
    B = 65536
    tr = BatchRunner(init_seed=0, act_fn=fn, rng_mode="partitionable", device=dev).collect(B)
    ref = orc.rollout(npo.key(0), B, 0, B, policy, 1)
    assert (tr.ep_len.cpu().numpy() == ref["ep_len"]).all()
    assert (tr.final_boards.cpu().numpy() == ref["final_boards"]).all()
    valid = tr.valid()
    ret = torch.where(valid, tr.rewards, torch.zeros_like(tr.rewards)).sum(0).cpu().numpy()
    assert (ret == ref["ep_return"]).all()  # integer-valued f32 sums: exact
    assert tr.T == int(ref["ep_len"].max())


def test_shard_of_524288_boards_matches_oracle_slice(dev):
    """configs[3]: 524 288 envs over 8 GPUs -- rank 5's slice equals the same slice of the single-device batch."""
    B_total, per = 524288, 65536
    r = BatchRunner(init_seed=7, act_fn=act_randomly, rng_mode="partitionable", device=dev, env0=5 * per,
                    total_envs=B_total)
    tr = r.collect(per)
    ref = orc.rollout(npo.key(7), B_total, 5 * per, per, 1, 1)
    assert (tr.ep_len.cpu().numpy() == ref["ep_len"]).all()
    assert (tr.final_boards.cpu().numpy() == ref["final_boards"]).all()


def test_shard_of_1M_boards_matches_oracle_slice(dev):
    """configs[4]: 1 048 576 envs over 8 GPUs -- rank 3's 131 072-board slice equals that slice of the single-device batch
    (keys come from the GLOBAL split, so a shard never depends on how many ranks there are)."""
    B_total, per = 1 << 20, 1 << 17
    r = BatchRunner(init_seed=42, act_fn=act_drul, rng_mode="partitionable", device=dev, env0=3 * per, total_envs=B_total)
    tr = r.collect(per)
    ref = orc.rollout(npo.key(42), B_total, 3 * per, per, 0, 1)
    assert (tr.ep_len.cpu().numpy() == ref["ep_len"]).all()
    assert (tr.final_boards.cpu().numpy() == ref["final_boards"]).all()
    # the same slice computed as two half-shards (16 ranks) is the same again
    halves = [BatchRunner(init_seed=42, act_fn=act_drul, rng_mode="partitionable", device=dev, env0=3 * per + h * per // 2,
                          total_envs=B_total).collect(per // 2) for h in (0, 1)]
    assert torch.equal(torch.cat([h.final_boards for h in halves]), tr.final_boards)


@pytest.mark.parametrize("mode", [0, 1])
def test_step_properties_at_4M_boards(dev, mode):
    """2^22 boards through g2048_step (explicit keys): sum of tile values grows by exactly the spawned tile, rewards
    are non-negative multiples of 4 on legal moves, illegal moves give -1 + termination, done boards are frozen."""
    B = 1 << 22
    boards = torch.empty((B, 16), dtype=torch.uint8, device=dev)
    masks = torch.empty(B, dtype=torch.uint8, device=dev)
    done = torch.empty(B, dtype=torch.uint8, device=dev)
    ep = torch.empty(B, dtype=torch.int32, device=dev)
    rew = torch.empty(B, dtype=torch.float32, device=dev)
    nv.reset_fused((11, 12), boards, masks, done, ep, B, 0, mode)
    tile_sum = lambda b: torch.where(b > 0, torch.pow(2.0, b.double()), torch.zeros_like(b, dtype=torch.float64)).sum(1)
    assert ((boards > 0).sum(1) == 2).all()
    g = torch.Generator(device=dev).manual_seed(1)
    for it in range(40):
        keys = nv.split((100 + it, 7), B, mode, dev)
        actions = torch.randint(0, 4, (B,), generator=g, device=dev, dtype=torch.int32)
        before, m_before, d_before = boards.clone(), masks.clone(), done.clone()
        s0 = tile_sum(before)
        nv.step(boards, masks, done, actions, keys, rew, mode)
        s1 = tile_sum(boards)
        legal = ((m_before >> actions.to(torch.uint8)) & 1).bool()
        live = d_before == 0
        frozen = ~live
        assert (boards[frozen] == before[frozen]).all() and (rew[frozen] == 0).all() and (done[frozen] == 1).all()
        grew = (s1 - s0)[live]
        full_illegal = live & ~legal & ((before == 0).sum(1) == 0)
        ok = (grew == 2) | (grew == 4)
        ok[full_illegal[live]] = True  # spawn on a full board overwrites cell 0 (Pgx choice over an all-zero p)
        assert ok.all()
        assert (rew[live & legal] >= 0).all() and (torch.remainder(rew[live & legal], 4) == 0).all()
        assert (rew[live & ~legal] == -1).all() and (done[live & ~legal] == 1).all()
        assert ((masks[done == 1] == 15).all())
        assert (masks[done == 0] != 0).all()
    assert done.float().mean() > 0.5  # random actions mostly die on an illegal move


def test_sample_is_unbiased_and_masked_at_1M(dev):
    """Categorical draw at 2^20 envs: legal-only, and frequencies match softmax(logits) to 4 sigma."""
    B = 1 << 20
    logits = torch.tensor([0.3, -1.0, 1.2, 0.1], device=dev).repeat(B, 1).contiguous()
    masks = torch.full((B,), 0b1101, dtype=torch.uint8, device=dev)
    keys = nv.split((3, 4), B, 1, dev)
    a = torch.empty(B, dtype=torch.int32, device=dev)
    lp = torch.empty(B, dtype=torch.float32, device=dev)
    nv.act_logits(keys, logits, masks, True, True, a, lp, 1)
    counts = torch.bincount(a.long(), minlength=4).double().cpu().numpy()
    assert counts[1] == 0
    p = torch.softmax(torch.tensor([0.3, -1e8, 1.2, 0.1], dtype=torch.float64), 0).numpy()
    sigma = np.sqrt(B * p * (1 - p)) + 1e-9
    assert (np.abs(counts - B * p) < 4 * sigma + 1).all()
    np.testing.assert_allclose(lp.cpu().numpy(), np.log(p[a.cpu().numpy()]), atol=1e-5)
