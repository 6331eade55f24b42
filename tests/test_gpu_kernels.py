"""Parity of every libg2048.so entry point against the C oracle, through the C ABI, on the GPU."""
import numpy as np
import pytest
import torch

from oracle import c_oracle as orc
from oracle import g2048_oracle as npo
from src.g2048 import native as nv

pytestmark = pytest.mark.gpu
MODES = [0, 1]


def _rand_boards(rng, n):
    b = rng.choice(np.arange(0, 7, dtype=np.uint8), size=(n, 16), p=[.35, .2, .15, .1, .08, .07, .05])
    b[: n // 50] = rng.integers(0, 18, size=(n // 50, 16))
    b[n // 50: n // 25] = rng.integers(1, 4, size=(n // 25 - n // 50, 16))  # full boards
    return b.astype(np.uint8)


def _t(a, dev, dtype=None):
    t = torch.from_numpy(np.ascontiguousarray(a))
    return t.to(dev) if dtype is None else t.to(dev).to(dtype)


@pytest.mark.parametrize("mode", MODES)
@pytest.mark.parametrize("n", [1, 2, 5, 64, 1000, 65537])
def test_split(dev, mode, n):
    key = npo.key(1234)
    out = nv.keys_to_numpy(nv.split(key, n, mode, dev))
    assert (out == orc.split(key, n, mode)).all()


@pytest.mark.parametrize("mode", MODES)
def test_chain_keys_host(mode):
    k, subs = nv.chain_keys(npo.key(7), 33, mode)
    k2, subs2 = orc.chain(npo.key(7), 33, mode)
    assert (k == k2).all() and (subs == subs2).all()


@pytest.mark.parametrize("mode", MODES)
@pytest.mark.parametrize("B", [1, 63, 64, 257, 100003])
def test_init(dev, mode, B):
    rng = np.random.default_rng(B)
    keys = rng.integers(0, 2**32, size=(B, 2), dtype=np.uint64).astype(np.uint32)
    boards = torch.empty((B, 16), dtype=torch.uint8, device=dev)
    masks = torch.empty(B, dtype=torch.uint8, device=dev)
    done = torch.empty(B, dtype=torch.uint8, device=dev)
    nv.init(nv.keys_from_numpy(keys, dev), boards, masks, done, mode)
    ob, om, od = orc.init(keys, mode)
    assert (boards.cpu().numpy() == ob).all()
    assert (masks.cpu().numpy() == om).all()
    assert (done.cpu().numpy() == 0).all()


@pytest.mark.parametrize("mode", MODES)
@pytest.mark.parametrize("B", [1, 65, 4096, 200001])
def test_step(dev, mode, B):
    rng = np.random.default_rng(100 + B)
    boards = _rand_boards(rng, B)
    true_mask = (npo.legal_mask(boards) * np.array([1, 2, 4, 8])).sum(1).astype(np.uint8)
    masks = np.where(rng.random(B) < 0.8, true_mask, rng.integers(0, 16, size=B)).astype(np.uint8)
    done = (rng.random(B) < 0.1).astype(np.uint8)
    actions = rng.integers(0, 4, size=B).astype(np.int32)
    keys = rng.integers(0, 2**32, size=(B, 2), dtype=np.uint64).astype(np.uint32)
    tb, tm, td = _t(boards, dev), _t(masks, dev), _t(done, dev)
    rew = torch.empty(B, dtype=torch.float32, device=dev)
    nv.step(tb, tm, td, _t(actions, dev), nv.keys_from_numpy(keys, dev), rew, mode)
    ob, om, od, orw = orc.step(boards, masks, done, actions, keys, mode)
    assert (tb.cpu().numpy() == ob).all()
    assert (tm.cpu().numpy() == om).all()
    assert (td.cpu().numpy() == od).all()
    assert (rew.cpu().numpy() == orw).all()


def test_observe(dev):
    rng = np.random.default_rng(5)
    boards = rng.integers(0, 31, size=(777, 16)).astype(np.uint8)
    obs = torch.empty((777, 4, 4, 31), dtype=torch.uint8, device=dev)
    nv.observe(_t(boards, dev), obs)
    assert (obs.cpu().numpy().astype(bool) == npo.observation(boards)).all()


@pytest.mark.parametrize("mode", MODES)
def test_act_fns(dev, mode):
    B = 50001
    rng = np.random.default_rng(9)
    masks = rng.integers(0, 16, size=B).astype(np.uint8)
    keys = rng.integers(0, 2**32, size=(B, 2), dtype=np.uint64).astype(np.uint32)
    tk, tm = nv.keys_from_numpy(keys, dev), _t(masks, dev)
    a = torch.empty(B, dtype=torch.int32, device=dev)
    lp = torch.empty(B, dtype=torch.float32, device=dev)
    nv.act_drul(tm, a)
    assert (a.cpu().numpy() == orc.act_drul(masks)).all()
    nv.act_random(tk, tm, a, lp, mode)
    oa, olp = orc.act_random(keys, masks, mode)
    assert (a.cpu().numpy() == oa).all() and (lp.cpu().numpy() == olp).all()
    logits = (rng.standard_normal((B, 4)) * 3).astype(np.float32)
    masks_nz = np.where(masks == 0, 15, masks).astype(np.uint8)
    for use_mask in (0, 1):
        for sample in (0, 1):
            nv.act_logits(tk, _t(logits, dev), _t(masks_nz, dev), use_mask, sample, a, lp, mode)
            oa, olp = orc.act_logits(keys, logits, masks_nz, use_mask, sample, mode)
            assert (a.cpu().numpy() == oa).all()
            np.testing.assert_allclose(lp.cpu().numpy(), olp, atol=2e-6, rtol=0)  # f32 expf/logf vs f64


def _fused_rollout(dev, seed, B_total, env0, B, policy, mode, fill_frozen, chunk=32, tcap=1024):
    key, subs = nv.chain_keys(npo.key(seed), 1 + 2 * tcap, mode)
    boards = torch.empty((B, 16), dtype=torch.uint8, device=dev)
    masks = torch.empty(B, dtype=torch.uint8, device=dev)
    done = torch.empty(B, dtype=torch.uint8, device=dev)
    ep_len = torch.empty(B, dtype=torch.int32, device=dev)
    nv.reset_fused(subs[0], boards, masks, done, ep_len, B_total, env0, mode)
    init_boards = boards.cpu().numpy().copy()
    trb = torch.zeros((tcap, B, 16), dtype=torch.uint8, device=dev)
    trm = torch.zeros((tcap, B), dtype=torch.uint8, device=dev)
    trr = torch.zeros((tcap, B), dtype=torch.float32, device=dev)
    trl = torch.zeros((tcap, B), dtype=torch.float32, device=dev)
    live = torch.zeros(1, dtype=torch.int32, device=dev)
    t = 0
    while True:
        live.zero_()
        nv.rollout_fused(subs[1 + 2 * t: 1 + 2 * (t + chunk)].reshape(chunk, 4), t, boards, masks, done, ep_len,
                         trb, trm, trr, trl, B_total, env0, policy, fill_frozen, mode, live)
        t += chunk
        if int(live.item()) == 0:
            break
        assert t + chunk <= tcap
    return dict(T=t, init=init_boards, boards=trb[:t].cpu().numpy(), meta=trm[:t].cpu().numpy(),
                rewards=trr[:t].cpu().numpy(), logp=trl[:t].cpu().numpy(), ep_len=ep_len.cpu().numpy(),
                final=boards.cpu().numpy())


@pytest.mark.parametrize("mode", MODES)
@pytest.mark.parametrize("policy", ["drul", "random"])
def test_rollout_fused_vs_numpy_oracle_full_frames(dev, mode, policy):
    """Whole [T,B] trajectory incl. frozen frames == the lock-step numpy oracle (BatchRunner semantics)."""
    B = 96
    tr = npo.Runner(11, mode).run(B, policy)
    T = tr["actions"].shape[1]
    got = _fused_rollout(dev, 11, B, 0, B, 0 if policy == "drul" else 1, mode, fill_frozen=True)
    assert got["T"] >= T
    assert (got["init"] == tr["init_boards"]).all()
    assert (got["boards"][:T].transpose(1, 0, 2) == tr["boards"]).all()
    meta = got["meta"][:T].T
    assert ((meta & 3) == tr["actions"]).all()
    mk = (tr["masks"] * np.array([1, 2, 4, 8])).sum(2)
    assert (((meta >> 2) & 15) == mk).all()
    assert (((meta >> 6) & 1) == tr["terms"]).all()
    assert (got["rewards"][:T].T == tr["rewards"]).all()
    if policy == "random":
        assert (got["logp"][:T].T == tr["log_probs"]).all()
    assert (got["ep_len"] == npo.episode_lengths(tr["terms"])).all()
    assert (got["final"] == tr["final_boards"]).all()


@pytest.mark.parametrize("mode", MODES)
def test_rollout_fused_shard_invariance_and_c_oracle(dev, mode):
    """A shard [env0, env0+B) of a B_total batch reproduces exactly that slice (multi-GPU sharding rule)."""
    B_total, env0, B = 5000, 1234, 700
    ref = orc.rollout(npo.key(3), B_total, env0, B, 1, mode, max_steps=1024)
    got = _fused_rollout(dev, 3, B_total, env0, B, 1, mode, fill_frozen=False)
    assert (got["ep_len"] == ref["ep_len"]).all()
    assert (got["final"] == ref["final_boards"]).all()
    ret = np.array([got["rewards"][: got["ep_len"][e], e].sum() for e in range(B)], np.float32)
    np.testing.assert_allclose(ret, ref["ep_return"], rtol=0, atol=0)


def test_golden_svg_frames_on_gpu(dev):
    """The reference's own animation frames (seed 0, 4 envs, legacy stream) replayed by the HIP kernels."""
    import os
    here = os.path.dirname(__file__)
    for name, pol in (("drul", 0), ("random", 1)):
        g = np.load(os.path.join(here, "golden", f"svg_{name}_seed0_b4.npy"))
        got = _fused_rollout(dev, 0, 4, 0, 4, pol, 0, fill_frozen=True)
        T = g.shape[0]
        # frame k = boards AFTER step k = observation of step k+1; last frame = final boards
        after = np.concatenate([got["boards"][1:T], got["final"][None]], axis=0)
        assert (after == g).all()


@pytest.mark.parametrize("mode", MODES)
def test_policy_step(dev, mode):
    B, B_total, env0 = 3000, 4000, 500
    rng = np.random.default_rng(21)
    _, subs = nv.chain_keys(npo.key(5), 1 + 2 * 40, mode)
    boards = torch.empty((B, 16), dtype=torch.uint8, device=dev)
    masks = torch.empty(B, dtype=torch.uint8, device=dev)
    done = torch.empty(B, dtype=torch.uint8, device=dev)
    ep_len = torch.empty(B, dtype=torch.int32, device=dev)
    nv.reset_fused(subs[0], boards, masks, done, ep_len, B_total, env0, mode)
    T = 40
    trb = torch.zeros((T, B, 16), dtype=torch.uint8, device=dev)
    trm = torch.zeros((T, B), dtype=torch.uint8, device=dev)
    trr = torch.zeros((T, B), dtype=torch.float32, device=dev)
    trl = torch.zeros((T, B), dtype=torch.float32, device=dev)
    trv = torch.zeros((T, B), dtype=torch.float32, device=dev)
    live = torch.zeros(1, dtype=torch.int32, device=dev)
    ob, om, od = boards.cpu().numpy(), masks.cpu().numpy(), done.cpu().numpy()
    for t in range(T):
        logits = (rng.standard_normal((B, 4)) * 2).astype(np.float32)
        values = rng.standard_normal(B).astype(np.float32)
        live.zero_()
        # no mask: illegal actions do occur and must terminate the env with reward -1
        nv.policy_step(subs[1 + 2 * t], subs[2 + 2 * t], _t(logits, dev), _t(values, dev), False, True, t, boards,
                       masks, done, ep_len, trb, trm, trr, trl, trv, B_total, env0, True, mode, live)
        ak = orc.split(subs[1 + 2 * t], B_total, mode)[env0: env0 + B]
        sk = orc.split(subs[2 + 2 * t], B_total, mode)[env0: env0 + B]
        a, lp = orc.act_logits(ak, logits, om, 0, 1, mode)
        assert (trb[t].cpu().numpy() == ob).all()
        nb, nm, nd, rw = orc.step(ob, om, od, a, sk, mode)
        meta = trm[t].cpu().numpy()
        assert ((meta & 3) == a).all() and (((meta >> 2) & 15) == om).all() and (((meta >> 6) & 1) == nd).all()
        assert (trr[t].cpu().numpy() == rw).all()
        np.testing.assert_allclose(trl[t].cpu().numpy(), lp, atol=2e-6, rtol=0)
        assert (trv[t].cpu().numpy() == values).all()
        assert int(live.item()) == int((nd == 0).sum())
        ob, om, od = nb, nm, nd
        assert (boards.cpu().numpy() == ob).all() and (done.cpu().numpy() == od).all()
    assert od.sum() > 0  # some envs did die from illegal moves


@pytest.mark.parametrize("gamma,lam", [(0.99, 0.95), (0.9, 0.5), (1.0, 1.0)])
def test_gae_flat_bit_exact(dev, gamma, lam):
    rng = np.random.default_rng(1)
    for N in (1, 7, 1000, 123457):
        r = rng.standard_normal(N).astype(np.float32) * 4
        v = rng.standard_normal(N).astype(np.float32)
        term = (rng.random(N) < 0.02).astype(np.uint8)
        adv = torch.empty(N, dtype=torch.float32, device=dev)
        ret = torch.empty(N, dtype=torch.float32, device=dev)
        nv.gae_flat(_t(r, dev), _t(v, dev), _t(term, dev), adv, ret, gamma, lam)
        oa, orr = orc.gae(r, v, term, gamma, lam)
        assert (adv.cpu().numpy() == oa).all() and (ret.cpu().numpy() == orr).all()


def test_gae_tb_and_compact(dev):
    rng = np.random.default_rng(2)
    T, B = 50, 333
    ep_len = rng.integers(0, T + 1, size=B).astype(np.int32)
    r = rng.standard_normal((T, B)).astype(np.float32)
    v = rng.standard_normal((T, B)).astype(np.float32)
    lp = rng.standard_normal((T, B)).astype(np.float32)
    boards = rng.integers(0, 12, size=(T, B, 16)).astype(np.uint8)
    act = rng.integers(0, 4, size=(T, B)).astype(np.uint8)
    msk = rng.integers(0, 16, size=(T, B)).astype(np.uint8)
    dn = (np.arange(T)[:, None] == (ep_len[None, :] - 1)).astype(np.uint8)
    meta = (act | (msk << 2) | (dn << 6)).astype(np.uint8)
    adv = torch.zeros((T, B), dtype=torch.float32, device=dev)
    ret = torch.zeros((T, B), dtype=torch.float32, device=dev)
    nv.gae_tb(_t(r, dev), _t(v, dev), _t(ep_len, dev), adv, ret, T, B, 0.99, 0.95)
    # reference layout: env-major flat buffer with termination flags
    lens = ep_len.astype(np.int64)
    N = int(lens.sum())
    flat = lambda x: npo.compact(np.swapaxes(x, 0, 1), lens)
    oa, orr = orc.gae(flat(r), flat(v), flat(dn), 0.99, 0.95)
    offs = np.concatenate([[0], np.cumsum(lens)[:-1]]).astype(np.int64)
    ob = torch.empty((N, 16), dtype=torch.uint8, device=dev)
    oact = torch.empty(N, dtype=torch.uint8, device=dev)
    omsk = torch.empty(N, dtype=torch.uint8, device=dev)
    otm = torch.empty(N, dtype=torch.uint8, device=dev)
    orw = torch.empty(N, dtype=torch.float32, device=dev)
    olp = torch.empty(N, dtype=torch.float32, device=dev)
    ov = torch.empty(N, dtype=torch.float32, device=dev)
    nv.compact(_t(boards, dev), _t(meta, dev), _t(r, dev), _t(lp, dev), _t(v, dev), _t(ep_len, dev), _t(offs, dev),
               ob, oact, omsk, orw, olp, ov, otm, T, B, N)
    assert (ob.cpu().numpy() == flat(boards)).all()
    assert (oact.cpu().numpy() == flat(act)).all() and (omsk.cpu().numpy() == flat(msk)).all()
    assert (otm.cpu().numpy() == flat(dn)).all()
    assert (orw.cpu().numpy() == flat(r)).all() and (olp.cpu().numpy() == flat(lp)).all()
    assert (ov.cpu().numpy() == flat(v)).all()
    # GAE on the [T][B] layout == the reference's scan over the compacted buffer, bit for bit
    adv_flat = flat(adv.cpu().numpy())
    ret_flat = flat(ret.cpu().numpy())
    assert (adv_flat == oa).all() and (ret_flat == orr).all()


def test_bad_arguments_are_rejected(dev):
    boards = torch.zeros((4, 16), dtype=torch.uint8, device=dev)
    masks = torch.zeros(4, dtype=torch.uint8, device=dev)
    with pytest.raises(nv.NativeError):
        nv.act_drul(masks.cpu(), torch.zeros(4, dtype=torch.int32))  # CPU tensors: no CPU path
    with pytest.raises(nv.NativeError):
        nv.act_drul(masks, torch.zeros(4, dtype=torch.int64, device=dev))  # wrong dtype
    with pytest.raises(nv.NativeError):
        nv.split(npo.key(0), 4, 7, dev)  # bad rng_mode -> G2048_EINVAL
