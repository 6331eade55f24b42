"""Pins the CPU oracle (numpy + C) on the reference's own artifacts.  CPU only."""
import json
import os

import numpy as np
import pytest

from oracle import c_oracle as orc
from oracle import g2048_oracle as npo

G = os.path.join(os.path.dirname(__file__), "golden")


def test_threefry_known_answers():
    # Random123 KATs for threefry2x32-20
    assert orc.threefry(0, 0, 0, 0) == (0x6B200159, 0x99BA4EFE)
    assert orc.threefry(0xFFFFFFFF, 0xFFFFFFFF, 0xFFFFFFFF, 0xFFFFFFFF) == (0x1CB996FC, 0xBB002BE7)
    assert orc.threefry(0x13198A2E, 0x03707344, 0x243F6A88, 0x85A308D3) == (0xC4923A9C, 0x483DF7A0)
    a = npo.threefry2x32(0, 0, 0, 0)
    assert (int(a[0]), int(a[1])) == (0x6B200159, 0x99BA4EFE)


def test_jax_split_known_answers():
    # the well-known legacy value of jax.random.split(jax.random.key(0)) and SURVEY.md A.4
    for impl in (npo.split, orc.split):
        assert impl(npo.key(0), 2, 0).tolist() == [[4146024105, 967050713], [2718843009, 1272950319]]
        assert impl(npo.key(42), 2, 0).tolist() == [[2465931498, 3679230171], [255383827, 267815257]]
    _, subs = orc.chain(npo.key(0), 1, 0)
    assert orc.split(subs[0], 4, 0).tolist() == [[2250432988, 1268406310], [2801332082, 3465333216],
                                                  [516624522, 2826104104], [1997759167, 3564182118]]


@pytest.mark.parametrize("name,policy", [("drul", "drul"), ("random", "random")])
def test_svg_frames_numpy_oracle(name, policy):
    """assets/2048_{drul,random}_actions.svg: seed 0, 4 envs, legacy stream, frame k = boards after step k."""
    g = np.load(os.path.join(G, f"svg_{name}_seed0_b4.npy"))
    tr = npo.Runner(0, npo.MODE_LEGACY).run(4, policy)
    got = tr["next_boards"].transpose(1, 0, 2)
    assert got.shape == g.shape
    assert (got == g).all()
    want_len = {"drul": [274, 172, 274, 285], "random": [123, 91, 123, 73]}[name]
    assert npo.episode_lengths(tr["terms"]).tolist() == want_len


@pytest.mark.parametrize("name,policy", [("drul", 0), ("random", 1)])
def test_svg_final_boards_c_oracle(name, policy):
    g = np.load(os.path.join(G, f"svg_{name}_seed0_b4.npy"))
    res = orc.rollout(npo.key(0), 4, 0, 4, policy, 0)
    for e in range(4):
        assert (res["final_boards"][e] == g[res["ep_len"][e] - 1, e]).all()
    assert res["ep_len"].max() == g.shape[0]


@pytest.mark.parametrize("name,policy", [("drul", 0), ("random", 1)])
def test_readme_histograms_c_oracle(name, policy):
    """README max-tile histograms: 1000 episodes = 10 x 100 envs, seed 42 + 100 i, partitionable stream."""
    want = json.load(open(os.path.join(G, "readme_histograms.json")))[f"{name}_percent"]
    tiles = []
    for i in range(10):
        res = orc.rollout(npo.key(42 + 100 * i), 100, 0, 100, policy, 1)
        tiles += (2 ** res["final_boards"].max(1).astype(int)).tolist()
    vals, counts = np.unique(tiles, return_counts=True)
    got = {str(int(v)): round(100.0 * c / 1000, 1) for v, c in zip(vals, counts)}
    assert got == {k: v for k, v in want.items()}
    assert round(float(np.mean(tiles)), 2) == {"random": 109.17, "drul": 189.44}[name]


@pytest.mark.parametrize("mode", [0, 1])
def test_numpy_and_c_oracle_agree_step_by_step(mode):
    B = 48
    r = npo.Runner(3, mode)
    tr = r.run(B, "random")
    T = tr["actions"].shape[1]
    key, subs = orc.chain(npo.key(3), 1 + 2 * T, mode)
    assert (key == r.key).all()
    b, m, d = orc.init(orc.split(subs[0], B, mode), mode)
    assert (b == tr["init_boards"]).all()
    for t in range(T):
        assert (m == (tr["masks"][:, t] * np.array([1, 2, 4, 8])).sum(1)).all()
        a, lp = orc.act_random(orc.split(subs[1 + 2 * t], B, mode), m, mode)
        assert (a == tr["actions"][:, t]).all() and (lp == tr["log_probs"][:, t]).all()
        b, m, d, rw = orc.step(b, m, d, a, orc.split(subs[2 + 2 * t], B, mode), mode)
        assert (b == tr["next_boards"][:, t]).all() and (rw == tr["rewards"][:, t]).all()
        assert (d == tr["terms"][:, t]).all()


def test_illegal_action_terminates_with_minus_one():
    boards = np.zeros((1, 16), np.uint8)
    boards[0, 0] = 1  # single tile in the corner: left and up are illegal
    masks = (npo.legal_mask(boards) * np.array([1, 2, 4, 8])).sum(1).astype(np.uint8)
    assert masks[0] == 0b1100
    nb, nm, nd, rw = orc.step(boards, masks, np.zeros(1, np.uint8), np.array([0]), np.array([[1, 2]], np.uint32), 1)
    assert nd[0] == 1 and rw[0] == -1.0 and nm[0] == 0xF
    nb2, nm2, nd2, rw2 = orc.step(nb, nm, nd, np.array([3]), np.array([[3, 4]], np.uint32), 1)
    assert (nb2 == nb).all() and rw2[0] == 0.0 and nd2[0] == 1  # frozen afterwards


def test_gae_oracle_matches_reference_fixture_bit_exact():
    """tests/golden/torch_reference.npz holds PPODataset outputs computed by the reference itself."""
    ref = np.load(os.path.join(G, "torch_reference.npz"))
    for i in range(3):
        gamma, lam = ref[f"gae{i}/params"]
        r, v, t = ref[f"gae{i}/rewards"], ref[f"gae{i}/values"], ref[f"gae{i}/terms"]
        for impl in (npo.gae, orc.gae):
            adv, ret = impl(r, v, t, float(gamma), float(lam))
            assert (adv == ref[f"gae{i}/raw_adv"]).all() and (ret == ref[f"gae{i}/raw_ret"]).all()
        np.testing.assert_allclose(npo.normalise(ref[f"gae{i}/raw_adv"]), ref[f"gae{i}/adv"], rtol=1e-5, atol=1e-6)
        np.testing.assert_allclose(npo.normalise(ref[f"gae{i}/raw_ret"]), ref[f"gae{i}/ret"], rtol=1e-5, atol=1e-6)


def test_compaction_oracle_matches_reference_rollout_buffer():
    ref = np.load(os.path.join(G, "torch_reference.npz"))
    lens = npo.episode_lengths(ref["buffer/in_term"])
    assert lens.tolist() == [4, 9, 0, 1, 6] and int(ref["buffer/size"]) == 20
    assert (npo.compact(ref["buffer/in_rew"], lens) == ref["buffer/out_rewards"]).all()
    assert (npo.compact(ref["buffer/in_term"], lens) == ref["buffer/out_terminations"]).all()
    assert (npo.compact(ref["buffer/in_obs"], lens).reshape(20, 16, 31) == ref["buffer/out_observations"]).all()


@pytest.mark.parametrize("mode", [npo.MODE_LEGACY, npo.MODE_PARTITIONABLE])
def test_autoreset_oracle_is_pinned_on_the_lockstep_oracle(mode):
    """The fixed-horizon / auto-reset restatement has no reference artifact of its own; it is pinned on the lock-step
    oracle (which the reference's assets pin): under the same policy every lane's FIRST episode is row for row the
    lock-step episode, and a restarted lane begins on a two-tile board."""
    B = 8

    def pf(keys, boards, masks):
        a, lp = npo.act_randomly(keys, masks, mode)
        return a, lp, np.zeros(len(a), np.float32)

    ref = npo.Runner(3, mode).run(B, "random")
    T = ref["actions"].shape[1] + 60
    ar = npo.AutoResetRunner(3, mode).run(B, T, policy_fn=pf)
    lens = npo.episode_lengths(ref["terms"])
    for e in range(B):
        n = lens[e]
        assert (ar["boards"][:n, e] == ref["boards"][e, :n]).all() and (ar["actions"][:n, e] == ref["actions"][e, :n]).all()
        assert (ar["rewards"][:n, e] == ref["rewards"][e, :n]).all() and ar["terms"][n - 1, e] and not ar["terms"][:n - 1, e].any()
        if n < T:
            assert (ar["boards"][n, e] > 0).sum() == 2 and ar["boards"][n, e].max() <= 2
    assert ar["terms"].sum() > B  # lanes kept playing after their first episode


def test_bootstrapped_gae_restatement_reduces_to_the_reference_scan():
    rng = np.random.default_rng(0)
    T, B = 40, 6
    r, v = rng.standard_normal((T, B)).astype(np.float32), rng.standard_normal((T, B)).astype(np.float32)
    d = rng.random((T, B)) < 0.1
    d[-1] = True  # every lane ends on a terminal step: the bootstrap value cannot matter
    adv, ret = npo.gae_bootstrap(r, v, d, rng.standard_normal(B).astype(np.float32), 0.99, 0.95)
    for e in range(B):
        fa, fr = npo.gae(r[:, e], v[:, e], d[:, e], 0.99, 0.95)
        assert (adv[:, e] == fa).all() and (ret[:, e] == fr).all()
