"""ctypes loader for the C oracle (oracle/g2048_oracle.c).  TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "libg2048_oracle.so")


def build(force: bool = False) -> str:
    src = os.path.join(_HERE, "g2048_oracle.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-B" if force else "-s"])
    return _SO


_lib = None


def lib():
    global _lib
    if _lib is None:
        _lib = C.CDLL(build())
        _lib.orc_rollout.restype = C.c_int64
        _lib.orc_num_threads.restype = C.c_int
    return _lib


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def threefry(k0, k1, c0, c1):
    o0, o1 = C.c_uint32(), C.c_uint32()
    lib().orc_threefry(C.c_uint32(k0), C.c_uint32(k1), C.c_uint32(c0), C.c_uint32(c1),
                       C.byref(o0), C.byref(o1))
    return o0.value, o1.value


def split(key, n, mode):
    key = np.ascontiguousarray(key, np.uint32)
    out = np.empty((n, 2), np.uint32)
    lib().orc_split(_p(key), _p(out), C.c_int64(n), C.c_int(mode))
    return out


def chain(key, n, mode):
    """n x (key, sub = split(key)); returns (new_key[2], subs[n,2])."""
    key = np.array(key, np.uint32)
    subs = np.empty((n, 2), np.uint32)
    lib().orc_chain(_p(key), _p(subs), C.c_int64(n), C.c_int(mode))
    return key, subs


def init(keys, mode):
    keys = np.ascontiguousarray(keys, np.uint32)
    B = keys.shape[0]
    boards = np.empty((B, 16), np.uint8)
    masks = np.empty(B, np.uint8)
    done = np.empty(B, np.uint8)
    lib().orc_init(_p(keys), _p(boards), _p(masks), _p(done), C.c_int64(B), C.c_int(mode))
    return boards, masks, done


def step(boards, masks, done, actions, keys, mode):
    """Returns NEW (boards, masks, done, rewards); inputs untouched."""
    boards = np.array(boards, np.uint8)
    masks = np.array(masks, np.uint8)
    done = np.array(done, np.uint8)
    actions = np.ascontiguousarray(actions, np.int32)
    keys = np.ascontiguousarray(keys, np.uint32)
    B = boards.shape[0]
    rewards = np.empty(B, np.float32)
    lib().orc_step(_p(boards), _p(masks), _p(done), _p(actions), _p(keys), _p(rewards),
                   C.c_int64(B), C.c_int(mode))
    return boards, masks, done, rewards


def act_drul(masks):
    masks = np.ascontiguousarray(masks, np.uint8)
    a = np.empty(masks.shape[0], np.int32)
    lib().orc_act_drul(_p(masks), _p(a), C.c_int64(masks.shape[0]))
    return a


def act_random(keys, masks, mode):
    keys = np.ascontiguousarray(keys, np.uint32)
    masks = np.ascontiguousarray(masks, np.uint8)
    B = masks.shape[0]
    a = np.empty(B, np.int32)
    lp = np.empty(B, np.float32)
    lib().orc_act_random(_p(keys), _p(masks), _p(a), _p(lp), C.c_int64(B), C.c_int(mode))
    return a, lp


def act_logits(keys, logits, masks, use_mask, sample, mode):
    keys = np.ascontiguousarray(keys, np.uint32)
    logits = np.ascontiguousarray(logits, np.float32)
    masks = np.ascontiguousarray(masks, np.uint8)
    B = masks.shape[0]
    a = np.empty(B, np.int32)
    lp = np.empty(B, np.float32)
    lib().orc_act_logits(_p(keys), _p(logits), _p(masks), C.c_int(int(use_mask)),
                         C.c_int(int(sample)), _p(a), _p(lp), C.c_int64(B), C.c_int(mode))
    return a, lp


def rollout(seed_key, B_total, e0, B, policy, mode, max_steps=8192):
    """Whole episodes for envs [e0, e0+B) of a B_total batch. policy: 0 drul, 1 random."""
    _, subs = chain(seed_key, 1 + 2 * max_steps, mode)
    fb = np.empty((B, 16), np.uint8)
    ln = np.empty(B, np.int32)
    ret = np.empty(B, np.float32)
    total = lib().orc_rollout(_p(subs), C.c_int64(max_steps), C.c_int64(B_total), C.c_int64(e0),
                              C.c_int64(B), C.c_int(policy), C.c_int(mode), _p(fb), _p(ln), _p(ret))
    return dict(final_boards=fb, ep_len=ln, ep_return=ret, total_steps=int(total))


def gae(r, v, term, gamma, lam):
    r = np.ascontiguousarray(r, np.float32)
    v = np.ascontiguousarray(v, np.float32)
    term = np.ascontiguousarray(term, np.uint8)
    adv = np.empty_like(r)
    ret = np.empty_like(r)
    lib().orc_gae(_p(r), _p(v), _p(term), _p(adv), _p(ret), C.c_int64(r.shape[0]),
                  C.c_double(gamma), C.c_double(lam))
    return adv, ret


def num_threads():
    return lib().orc_num_threads()


def set_num_threads(n: int):
    """OpenMP threads used by the batched entry points from now on (bench.py's cpu_baseline matrix)."""
    lib().orc_set_num_threads(C.c_int(int(n)))
