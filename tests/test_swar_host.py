"""The branch-free SWAR board code of csrc/g2048_device.h, compiled for the host, against the oracle.  CPU only.
(Checks the logic the HIP kernels run; the -m gpu tests check the kernels themselves through the C ABI.)"""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

from oracle import c_oracle as orc
from oracle import g2048_oracle as npo

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))


@pytest.fixture(scope="module")
def swar():
    out_dir = os.path.join(ROOT, "tests", "_build")
    os.makedirs(out_dir, exist_ok=True)
    so = os.path.join(out_dir, "libswar_host.so")
    subprocess.check_call(["g++", "-O1", "-ffp-contract=off", "-shared", "-fPIC", "-I",
                           os.path.join(ROOT, "2048-ppo-agent_amd", "csrc"), "-o", so,
                           os.path.join(ROOT, "tests", "host_swar", "swar_host.cpp")])
    return C.CDLL(so)


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def test_every_row_slides_like_the_oracle(swar):
    """All 18^4 rows (exponents 0..17) in all four directions, via boards whose other rows are empty."""
    vals = np.arange(18, dtype=np.uint8)
    rows = np.stack(np.meshgrid(vals, vals, vals, vals, indexing="ij"), -1).reshape(-1, 4)
    N = rows.shape[0]
    for a in range(4):
        boards = np.zeros((N, 16), np.uint8)
        if a in (0, 2):
            boards[:, 4:8] = rows  # row 1
        else:
            boards[:, 2::4] = rows  # column 2
        acts = np.full(N, a, np.int32)
        got = boards.copy()
        score = np.empty(N, np.float32)
        legal = np.empty(N, np.uint8)
        swar.hst_move(_p(got), _p(acts), _p(score), _p(legal), C.c_int64(N))
        want, wscore = npo.move(boards, acts)
        assert (got == want).all() and (score == wscore).all()
        assert (legal == (npo.legal_mask(boards) * np.array([1, 2, 4, 8])).sum(1)).all()


@pytest.mark.parametrize("mode", [0, 1])
def test_init_step_and_policies(swar, mode):
    rng = np.random.default_rng(mode)
    N = 100000
    boards = rng.choice(np.arange(0, 7, dtype=np.uint8), size=(N, 16), p=[.35, .2, .15, .1, .08, .07, .05])
    boards[:2000] = rng.integers(1, 4, size=(2000, 16))  # full boards
    keys = rng.integers(0, 2**32, size=(N, 2), dtype=np.uint64).astype(np.uint32)
    ob, om, _ = orc.init(keys, mode)
    b, m, d = np.empty((N, 16), np.uint8), np.empty(N, np.uint8), np.empty(N, np.uint8)
    swar.hst_init(_p(keys), _p(b), _p(m), _p(d), C.c_int64(N), C.c_int(mode))
    assert (b == ob).all() and (m == om).all()
    true_mask = (npo.legal_mask(boards) * np.array([1, 2, 4, 8])).sum(1).astype(np.uint8)
    masks = np.where(rng.random(N) < 0.8, true_mask, rng.integers(0, 16, size=N)).astype(np.uint8)
    done = (rng.random(N) < 0.1).astype(np.uint8)
    acts = rng.integers(0, 4, size=N).astype(np.int32)
    wb, wm, wd, wr = orc.step(boards, masks, done, acts, keys, mode)
    b, m, d, r = boards.copy(), masks.copy(), done.copy(), np.empty(N, np.float32)
    swar.hst_step(_p(b), _p(m), _p(d), _p(acts), _p(keys), _p(r), C.c_int64(N), C.c_int(mode))
    assert (b == wb).all() and (m == wm).all() and (d == wd).all() and (r == wr).all()
    a, lp = np.empty(N, np.int32), np.empty(N, np.float32)
    mk = rng.integers(0, 16, size=N).astype(np.uint8)
    swar.hst_act_drul(_p(mk), _p(a), C.c_int64(N))
    assert (a == orc.act_drul(mk)).all()
    swar.hst_act_random(_p(keys), _p(mk), _p(a), _p(lp), C.c_int64(N), C.c_int(mode))
    wa, wlp = orc.act_random(keys, mk, mode)
    assert (a == wa).all() and (lp == wlp).all()
    logits = (rng.standard_normal((N, 4)) * 3).astype(np.float32)
    mk = np.where(mk == 0, 15, mk).astype(np.uint8)
    for um in (0, 1):
        for s in (0, 1):
            swar.hst_act_logits(_p(keys), _p(logits), _p(mk), C.c_int(um), C.c_int(s), _p(a), _p(lp), C.c_int64(N),
                                C.c_int(mode))
            wa, wlp = orc.act_logits(keys, logits, mk, um, s, mode)
            assert (a == wa).all()
            np.testing.assert_allclose(lp, wlp, atol=2e-6, rtol=0)
    for n in (1, 2, 5, 64, 101):
        out = np.empty((n, 2), np.uint32)
        swar.hst_split(_p(npo.key(9)), _p(out), C.c_int64(n), C.c_int(mode))
        assert (out == orc.split(npo.key(9), n, mode)).all()
