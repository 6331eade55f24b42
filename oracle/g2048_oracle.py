"""CPU oracle (numpy) for the 2048 rollout hot path.  TEST INFRASTRUCTURE ONLY.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may
import this module.  The product (``2048-ppo-agent_amd/``) never does.

What it restates
----------------
The reference's env arithmetic is not in its tree: it calls third-party ``pgx==2.6.0``
(``pgx.make("2048")``) on ``jax==0.5.3`` (reference ``uv.lock:1564-1565``, ``:701-702``) from
``src/runs/batch_runner.py:32-37,105-128`` and ``src/runs/run_actions_batch.py:31-54``.
This file restates the published algorithm of that env and of ``jax.random`` (threefry2x32)
as specified in SURVEY.md Appendix A, and the in-tree plug-ins/drivers:

* ``act_drul``            <- reference ``src/actions/act_drul.py:40-44``
* ``act_randomly``        <- reference ``src/actions/act_randomly.py:40-54``
* driver key schedule     <- reference ``src/runs/batch_runner.py:105-128``
* keep-through-first-termination compaction <- ``src/ppo/rollout_buffer.py:164-187``
* GAE reverse scan + normalisation          <- ``src/ppo/data_loader.py:103-130``, ``:61-67``

Parity pin (NOT "parity unpinned"): ``tests/test_oracle_golden.py`` checks this oracle against
board-frames derived from the reference's own assets (``assets/2048_drul_actions.svg`` 285x4
frames, ``assets/2048_random_actions.svg`` 123x4 frames, legacy threefry stream) and against the
1000-episode max-tile histograms published in the reference README (partitionable stream).

One deliberate restatement choice: the two ``log`` calls of the Gumbel draw (jax: XLA's f32 log) are
evaluated with a fixed f32 polynomial in individually rounded IEEE mul/add (``_log_f32_poly``) so that
the C oracle, this oracle and the HIP kernels agree bit-for-bit with each other; the reference's assets
are still reproduced exactly with it.
"""
from __future__ import annotations

import numpy as np

MODE_LEGACY = 0  # jax_threefry_partitionable=False (matches the SVG assets)
MODE_PARTITIONABLE = 1  # default of jax 0.5.3 (matches the README histograms)

U32 = np.uint32
_F32_TINY = np.float32(np.finfo(np.float32).tiny)
_F32_MIN = np.float32(np.finfo(np.float32).min)

# --------------------------------------------------------------------------------------
# threefry2x32 (20 rounds) -- SURVEY.md Appendix A.2
# --------------------------------------------------------------------------------------
_ROT = ((13, 15, 26, 6), (17, 29, 16, 24))


def _rotl(x, r):
    return (x << U32(r)) | (x >> U32(32 - r))


def threefry2x32(k0, k1, c0, c1):
    """One threefry2x32 block per element; all args broadcast as uint32 arrays."""
    k0 = np.asarray(k0, dtype=U32)
    k1 = np.asarray(k1, dtype=U32)
    x0 = np.asarray(c0, dtype=U32)
    x1 = np.asarray(c1, dtype=U32)
    with np.errstate(over="ignore"):
        ks = (k0, k1, k0 ^ k1 ^ U32(0x1BD11BDA))
        x0 = x0 + ks[0]
        x1 = x1 + ks[1]
        for g in range(5):
            for r in _ROT[g % 2]:
                x0 = x0 + x1
                x1 = _rotl(x1, r)
                x1 = x1 ^ x0
            x0 = x0 + ks[(g + 1) % 3]
            x1 = x1 + ks[(g + 2) % 3] + U32(g + 1)
    return x0, x1


def key(seed: int) -> np.ndarray:
    """jax.random.key(seed) for 0 <= seed < 2**32 -> (0, seed)."""
    return np.array([0, seed & 0xFFFFFFFF], dtype=U32)


def split(k: np.ndarray, n: int, mode: int) -> np.ndarray:
    """jax.random.split(k, n) for ONE key k[2] -> [n, 2]."""
    k = np.asarray(k, dtype=U32)
    j = np.arange(n, dtype=U32)
    if mode == MODE_PARTITIONABLE:
        o0, o1 = threefry2x32(k[0], k[1], np.zeros(n, U32), j)
        return np.stack([o0, o1], axis=1)
    o0, o1 = threefry2x32(k[0], k[1], j, U32(n) + j)
    flat = np.concatenate([o0, o1])
    return flat.reshape(n, 2)


def split_each(keys: np.ndarray, mode: int):
    """vmap(jax.random.split)(keys) with n=2: keys [B,2] -> (first [B,2], second [B,2])."""
    keys = np.asarray(keys, dtype=U32)
    k0, k1 = keys[:, 0], keys[:, 1]
    if mode == MODE_PARTITIONABLE:
        a0, a1 = threefry2x32(k0, k1, U32(0), U32(0))
        b0, b1 = threefry2x32(k0, k1, U32(0), U32(1))
        return np.stack([a0, a1], 1), np.stack([b0, b1], 1)
    # legacy: blocks (0,2) and (1,3); flat = [b0.o0, b1.o0, b0.o1, b1.o1]
    p0, p1 = threefry2x32(k0, k1, U32(0), U32(2))
    q0, q1 = threefry2x32(k0, k1, U32(1), U32(3))
    return np.stack([p0, q0], 1), np.stack([p1, q1], 1)


def bits_scalar(keys: np.ndarray, mode: int) -> np.ndarray:
    """32 random bits of shape () per key: keys [B,2] -> [B] uint32."""
    keys = np.asarray(keys, dtype=U32)
    o0, o1 = threefry2x32(keys[:, 0], keys[:, 1], U32(0), U32(0))
    return (o0 ^ o1) if mode == MODE_PARTITIONABLE else o0


def bits_vec4(keys: np.ndarray, mode: int) -> np.ndarray:
    """32 random bits of shape (4,) per key: keys [B,2] -> [B,4] uint32."""
    keys = np.asarray(keys, dtype=U32)
    k0, k1 = keys[:, 0], keys[:, 1]
    if mode == MODE_PARTITIONABLE:
        out = [np.bitwise_xor(*threefry2x32(k0, k1, U32(0), U32(i))) for i in range(4)]
        return np.stack(out, axis=1)
    p0, p1 = threefry2x32(k0, k1, U32(0), U32(2))
    q0, q1 = threefry2x32(k0, k1, U32(1), U32(3))
    return np.stack([p0, q0, p1, q1], axis=1)


def uniform_f32(bits: np.ndarray) -> np.ndarray:
    """jax.random.uniform's bit trick: [0,1) float32 from 32 random bits."""
    b = (np.asarray(bits, dtype=U32) >> U32(9)) | U32(0x3F800000)
    return b.view(np.float32) - np.float32(1.0)


def _log_f32(x: np.ndarray) -> np.ndarray:
    """Correctly rounded f32 log (via f64): where the reference takes log(1/n) in act_randomly."""
    return np.log(x.astype(np.float64)).astype(np.float32)


_LOG_P = [np.float32(c) for c in (7.0376836292E-2, -1.1514610310E-1, 1.1676998740E-1, -1.2420140846E-1,
                                  1.4249322787E-1, -1.6668057665E-1, 2.0000714765E-1, -2.4999993993E-1,
                                  3.3333331174E-1)]


def _log_f32_poly(x: np.ndarray) -> np.ndarray:
    """f32 log in pure IEEE f32 mul/add (Cephes logf polynomial), operation-for-operation the same as
    oracle/g2048_oracle.c:log_f32_poly and the HIP kernels' log_f32 (used for the Gumbel noise)."""
    x = np.asarray(x, np.float32)
    bits = x.view(U32)
    e = ((bits >> U32(23)) & U32(0xFF)).astype(np.int32) - 126
    m = ((bits & U32(0x007FFFFF)) | U32(0x3F000000)).view(np.float32)
    small = m < np.float32(0.707106781186547524)
    e = np.where(small, e - 1, e)
    m = np.where(small, (m + m) + np.float32(-1.0), m + np.float32(-1.0)).astype(np.float32)
    z = m * m
    y = np.full_like(m, _LOG_P[0])
    for c in _LOG_P[1:]:
        y = y * m + c  # numpy rounds the product and the sum separately (no FMA)
    y = (y * m) * z
    fe = e.astype(np.float32)
    y = y + np.float32(-2.12194440e-4) * fe
    y = y + np.float32(-0.5) * z
    return ((m + y) + np.float32(0.693359375) * fe).astype(np.float32)


def gumbel_f32(bits: np.ndarray) -> np.ndarray:
    """jax.random.gumbel from raw bits: -log(-log(u)), u in [tiny, 1)."""
    f = uniform_f32(bits)
    u = np.maximum(_F32_TINY, f * np.float32(1.0) + _F32_TINY)
    return -_log_f32_poly(-_log_f32_poly(u))


def categorical4(keys: np.ndarray, logits: np.ndarray, mode: int) -> np.ndarray:
    """jax.random.categorical(key, logits[4]) per env: argmax(gumbel + logits), first max wins."""
    g = gumbel_f32(bits_vec4(keys, mode))
    return np.argmax(g + np.asarray(logits, dtype=np.float32), axis=1).astype(np.int32)


# --------------------------------------------------------------------------------------
# board arithmetic -- SURVEY.md Appendix A.1
# boards: [B,16] uint8 of log2(tile), 0 = empty, row-major.
# actions: 0 = left, 1 = up, 2 = right, 3 = down.
# --------------------------------------------------------------------------------------
def _compact_left(rows: np.ndarray) -> np.ndarray:
    order = np.argsort(rows == 0, axis=1, kind="stable")
    return np.take_along_axis(rows, order, axis=1)


def slide_rows_left(rows: np.ndarray):
    """rows [N,4] uint8 -> (slid rows [N,4] uint8, merge score [N] float32)."""
    a = _compact_left(rows).astype(np.int32)
    score = np.zeros(a.shape[0], dtype=np.int64)
    for i in range(3):
        m = (a[:, i] == a[:, i + 1]) & (a[:, i] != 0)
        a[:, i] += m
        a[:, i + 1] = np.where(m, 0, a[:, i + 1])
        score += np.where(m, np.int64(1) << a[:, i].astype(np.int64), 0)
    return _compact_left(a.astype(np.uint8)), score.astype(np.float32)


def move(boards: np.ndarray, actions: np.ndarray):
    """Apply slide/merge in direction actions[e] -> (new boards, merge score f32)."""
    B = boards.shape[0]
    out = np.empty_like(boards)
    score = np.zeros(B, dtype=np.float32)
    grid = boards.reshape(B, 4, 4)
    for a in range(4):
        sel = np.nonzero(actions == a)[0]
        if sel.size == 0:
            continue
        rot = np.rot90(grid[sel], k=a, axes=(1, 2))
        slid, sc = slide_rows_left(rot.reshape(-1, 4))
        slid = np.rot90(slid.reshape(-1, 4, 4), k=-a, axes=(1, 2))
        out[sel] = slid.reshape(-1, 16)
        score[sel] = sc.reshape(-1, 4).sum(axis=1, dtype=np.float32)
    return out, score


def legal_mask(boards: np.ndarray) -> np.ndarray:
    """legal[a] = moving in direction a changes the board. -> [B,4] bool."""
    B = boards.shape[0]
    m = np.zeros((B, 4), dtype=bool)
    for a in range(4):
        moved, _ = move(boards, np.full(B, a, dtype=np.int32))
        m[:, a] = (moved != boards).any(axis=1)
    return m


def spawn(boards: np.ndarray, keys: np.ndarray, mode: int) -> np.ndarray:
    """Place one 2 (p=.9) / 4 (p=.1) on a uniformly chosen empty cell (jax.random.choice)."""
    kpos, kval = split_each(keys, mode)
    p = (boards == 0).astype(np.float32)
    c = np.cumsum(p, axis=1, dtype=np.float32)
    u = uniform_f32(bits_scalar(kpos, mode))
    r = c[:, 15] * (np.float32(1.0) - u)
    pos = (c < r[:, None]).sum(axis=1)  # searchsorted(c, r, side="left")
    pos = np.minimum(pos, 15)
    u2 = uniform_f32(bits_scalar(kval, mode))
    r2 = np.float32(1.0) - u2
    val = np.where(r2 <= np.float32(0.9), 1, 2).astype(np.uint8)
    out = boards.copy()
    out[np.arange(boards.shape[0]), pos] = val
    return out


def env_init(keys: np.ndarray, mode: int) -> np.ndarray:
    """pgx 2048 init: two spawns on an empty board. keys [B,2] -> boards [B,16].

    The initial legal_action_mask is the TRUE legal mask of the initial board (``legal_mask``),
    not all-True: the first frame of assets/2048_random_actions.svg only reproduces that way
    (SURVEY.md Appendix A.1 guessed "all True"; the asset refutes it).
    """
    k1, k2 = split_each(keys, mode)
    b = np.zeros((keys.shape[0], 16), dtype=np.uint8)
    b = spawn(b, k1, mode)
    return spawn(b, k2, mode)


def env_step(boards, masks, done, actions, keys, mode):
    """pgx ``env.step`` incl. the core wrapper semantics.

    boards [B,16] u8, masks [B,4] bool (incoming legal mask), done [B] bool,
    actions [B] int, keys [B,2] u32.  Returns (boards', rewards f32, masks', done').
    """
    actions = np.asarray(actions).astype(np.int64)
    B = boards.shape[0]
    illegal = ~masks[np.arange(B), actions]
    moved, score = move(boards, actions)
    spawned = spawn(moved, keys, mode)
    new_mask = legal_mask(spawned)
    terminated = ~new_mask.any(axis=1)
    rewards = score.copy()
    terminated = terminated | illegal
    rewards = np.where(illegal, np.float32(-1.0), rewards)
    new_mask = np.where(terminated[:, None], True, new_mask)
    # already-terminated envs: frozen, zero reward
    out_b = np.where(done[:, None], boards, spawned)
    out_r = np.where(done, np.float32(0.0), rewards).astype(np.float32)
    out_m = np.where(done[:, None], masks, new_mask)
    out_d = done | terminated
    return out_b, out_r, out_m, out_d


def observation(boards: np.ndarray) -> np.ndarray:
    """obs[e, r, c, k] = (board[e, 4r+c] == k) -> [B,4,4,31] bool."""
    return (boards.reshape(-1, 4, 4, 1) == np.arange(31, dtype=np.uint8)).astype(bool)


# --------------------------------------------------------------------------------------
# act_fn plug-ins
# --------------------------------------------------------------------------------------
def act_drul(masks: np.ndarray) -> np.ndarray:
    """First legal of [3,2,1,0]; all-false mask -> 3 (argmax of all-False is index 0)."""
    order = np.array([3, 2, 1, 0])
    return order[np.argmax(masks[:, order], axis=1)].astype(np.int32)


def act_randomly(keys: np.ndarray, masks: np.ndarray, mode: int):
    n = masks.sum(axis=1).astype(np.float32)
    m = masks.astype(np.float32)
    probs = np.where((n > 0)[:, None], m / np.maximum(n, 1)[:, None], np.float32(0.25))
    with np.errstate(divide="ignore"):
        logits = np.maximum(_log_f32(probs.astype(np.float32)), _F32_MIN)
    action = categorical4(keys, logits, mode)
    with np.errstate(divide="ignore"):
        logp = _log_f32(probs[np.arange(len(action)), action].astype(np.float32))
    return action, logp


def sample_policy(keys, logits, mode, sample=True):
    """TorchActionFunction tail (reference src/ppo/torch_action_wrapper.py:85-102)."""
    logits = np.maximum(np.asarray(logits, np.float32), _F32_MIN)
    if sample:
        action = categorical4(keys, logits, mode)
    else:
        action = np.argmax(logits, axis=1).astype(np.int32)
    mx = logits.max(axis=1, keepdims=True)
    lse = (mx[:, 0].astype(np.float64)
           + np.log(np.exp((logits - mx).astype(np.float64)).sum(axis=1)))
    logp = logits[np.arange(len(action)), action] - lse.astype(np.float32)  # f32, as jax
    return action, logp


# --------------------------------------------------------------------------------------
# driver (BatchRunner key schedule) -- reference src/runs/batch_runner.py:105-128
# --------------------------------------------------------------------------------------
class Runner:
    """Lock-step rollout of B envs until all terminate, with the reference key schedule."""

    def __init__(self, seed: int, mode: int):
        self.key = key(seed)
        self.mode = mode

    def _next_sub(self):
        ks = split(self.key, 2, self.mode)
        self.key = ks[0]
        return ks[1]

    def run(self, B: int, policy: str = "drul", policy_fn=None, max_steps: int = 100000):
        """Returns dict of [B,T] trajectories (+ 'init_boards', 'final_boards').

        policy: 'drul' | 'random' | 'callable' (policy_fn(act_keys, boards, masks) -> (a, lp, v)).
        boards[:, t]/masks[:, t] are PRE-step, rewards/terms POST-step (batch_runner.py:121-136).
        """
        mode = self.mode
        boards = env_init(split(self._next_sub(), B, mode), mode)
        masks = legal_mask(boards)
        done = np.zeros(B, dtype=bool)
        tr = {k: [] for k in ("boards", "actions", "masks", "log_probs", "values",
                              "rewards", "terms", "next_boards")}
        init_boards = boards.copy()
        while not done.all():
            if len(tr["actions"]) >= max_steps:
                raise RuntimeError("max_steps exceeded")
            act_keys = split(self._next_sub(), B, mode)
            if policy == "drul":
                a = act_drul(masks)
                lp = np.zeros(B, np.float32)
                v = np.zeros(B, np.float32)
            elif policy == "random":
                a, lp = act_randomly(act_keys, masks, mode)
                v = np.zeros(B, np.float32)
            else:
                a, lp, v = policy_fn(act_keys, boards, masks)
            step_keys = split(self._next_sub(), B, mode)
            nb, r, nm, nd = env_step(boards, masks, done, a, step_keys, mode)
            tr["boards"].append(boards)
            tr["actions"].append(np.asarray(a, np.int32))
            tr["masks"].append(masks)
            tr["log_probs"].append(np.asarray(lp, np.float32))
            tr["values"].append(np.asarray(v, np.float32))
            tr["rewards"].append(r)
            tr["terms"].append(nd)
            tr["next_boards"].append(nb)
            boards, masks, done = nb, nm, nd
        out = {k: np.stack(v, axis=1) for k, v in tr.items()}
        out["init_boards"] = init_boards
        out["final_boards"] = boards
        return out


def fold_in(k: np.ndarray, data: int) -> np.ndarray:
    """jax.random.fold_in(key, data) for 32-bit data: one threefry block over the counter words (0, data)."""
    a, b = threefry2x32(np.uint32(k[0]), np.uint32(k[1]), np.uint32(0), np.uint32(data))
    return np.array([a, b], dtype=np.uint32)


class AutoResetRunner(Runner):
    """Fixed-horizon rollout with per-lane auto-reset: the throughput mode of the build.  There is no
    reference code for it (the reference only has the lock-step loop, src/runs/batch_runner.py:117); this restatement
    defines it in terms of the pinned primitives: the BatchRunner key chain (one init split, then act / step splits per
    lock-step), ``env_step`` for every lane at every step, and for a lane whose step terminated
    ``env_init(split(fold_in(step_sub, 0xFFFFFFFF), B)[e])`` as its state for the next step.  Env state and the key chain
    persist across ``run`` calls."""

    def __init__(self, seed: int, mode: int):
        super().__init__(seed, mode)
        self.state = None

    def run(self, B: int, T: int, policy_fn=None, actions=None):
        """T steps.  Either ``policy_fn(act_keys, boards, masks) -> (a, lp, v)`` or recorded ``actions`` [T, B].
        Returns step-major arrays [T, B]: boards/masks BEFORE step t, rewards/terms AFTER it, + final_boards/final_masks."""
        mode = self.mode
        if self.state is None:
            boards = env_init(split(self._next_sub(), B, mode), mode)
            self.state = (boards, legal_mask(boards))
        boards, masks = self.state
        tr = {k: [] for k in ("boards", "actions", "masks", "log_probs", "values", "rewards", "terms")}
        alive = np.zeros(B, dtype=bool)  # "done" input of env_step: every lane is live at every step
        for t in range(T):
            act_keys = split(self._next_sub(), B, mode)
            if actions is not None:
                a, lp, v = np.asarray(actions[t], np.int32), np.zeros(B, np.float32), np.zeros(B, np.float32)
            else:
                a, lp, v = policy_fn(act_keys, boards, masks)
            step_sub = self._next_sub()
            nb, r, nm, nd = env_step(boards, masks, alive, a, split(step_sub, B, mode), mode)
            tr["boards"].append(boards)
            tr["actions"].append(np.asarray(a, np.int32))
            tr["masks"].append(masks)
            tr["log_probs"].append(np.asarray(lp, np.float32))
            tr["values"].append(np.asarray(v, np.float32))
            tr["rewards"].append(r)
            tr["terms"].append(nd)
            if nd.any():
                fresh = env_init(split(fold_in(step_sub, 0xFFFFFFFF), B, mode), mode)
                nb = np.where(nd[:, None], fresh, nb)
                nm = np.where(nd[:, None], legal_mask(fresh), nm)
            boards, masks = nb, nm
        self.state = (boards, masks)
        out = {k: np.stack(v, axis=0) for k, v in tr.items()}
        out["final_boards"], out["final_masks"] = boards, masks
        return out


# --------------------------------------------------------------------------------------
# rollout buffer compaction + GAE
# --------------------------------------------------------------------------------------
def episode_lengths(terms: np.ndarray) -> np.ndarray:
    """Per env: first termination index + 1, or 0 if it never terminates (dropped)."""
    has = terms.any(axis=1)
    return np.where(has, terms.argmax(axis=1) + 1, 0).astype(np.int64)


def compact(arr: np.ndarray, lens: np.ndarray) -> np.ndarray:
    """[B,T,...] -> env-major flat [sum(lens), ...] keeping steps 0..len_e-1 of each env."""
    return np.concatenate([arr[e, : lens[e]] for e in range(arr.shape[0])], axis=0)


def gae(rewards, values, terms, gamma: float, lam: float):
    """Reverse scan over the flat buffer, float32 op-for-op as torch does it on 0-d tensors."""
    r = np.asarray(rewards, np.float32)
    v = np.asarray(values, np.float32)
    t = np.asarray(terms, bool)
    g = np.float32(gamma)
    gl = np.float32(gamma * lam)
    adv = np.zeros_like(r)
    ret = np.zeros_like(r)
    last_gae = np.float32(0.0)
    last_v = np.float32(0.0)
    for i in range(len(r) - 1, -1, -1):
        if t[i]:
            last_v = np.float32(0.0)
            last_gae = np.float32(0.0)
        delta = np.float32(np.float32(r[i] + np.float32(g * last_v)) - v[i])
        last_gae = np.float32(delta + np.float32(gl * last_gae))
        adv[i] = last_gae
        ret[i] = np.float32(last_gae + v[i])
        last_v = v[i]
    return adv, ret


def gae_bootstrap(rewards, values, terms, last_values, gamma: float, lam: float):
    """The same scan per lane over step-major [T, B] arrays of a fixed-horizon rollout, started from
    last_values[e] = V(state after step T-1) (the reference's loop with last_value initialised to the bootstrap)."""
    r = np.asarray(rewards, np.float32)
    v = np.asarray(values, np.float32)
    d = np.asarray(terms, bool)
    T, B = r.shape
    adv, ret = np.zeros_like(r), np.zeros_like(r)
    g, gl = np.float32(gamma), np.float32(gamma * lam)
    for e in range(B):
        last_gae, last_v = np.float32(0.0), np.float32(last_values[e])
        for t in range(T - 1, -1, -1):
            if d[t, e]:
                last_v = np.float32(0.0)
                last_gae = np.float32(0.0)
            delta = np.float32(np.float32(r[t, e] + np.float32(g * last_v)) - v[t, e])
            last_gae = np.float32(delta + np.float32(gl * last_gae))
            adv[t, e] = last_gae
            ret[t, e] = np.float32(last_gae + v[t, e])
            last_v = v[t, e]
    return adv, ret


def normalise(x: np.ndarray) -> np.ndarray:
    """(x - mean) / (unbiased std + 1e-8), as torch .mean()/.std() (float64 accumulate here)."""
    x64 = x.astype(np.float64)
    return ((x64 - x64.mean()) / (x64.std(ddof=1) + 1e-8)).astype(np.float32)
