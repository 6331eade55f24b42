/*
 * CPU oracle (plain C) for the 2048 rollout hot path.  TEST INFRASTRUCTURE ONLY.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library.
 * The product (2048-ppo-agent_amd/) never links or calls it.
 *
 * It restates, scalar and un-optimised, the algorithm the reference runs through third-party
 * pgx==2.6.0 / jax==0.5.3 (reference uv.lock:1564-1565,701-702; call sites
 * src/runs/batch_runner.py:32-37,105-128) as specified in SURVEY.md Appendix A, plus the in-tree
 * plug-ins (src/actions/act_drul.py:40-44, src/actions/act_randomly.py:40-54), the driver key
 * schedule (src/runs/batch_runner.py:105-128), the keep-through-first-termination rule
 * (src/ppo/rollout_buffer.py:164-187) and the GAE reverse scan (src/ppo/data_loader.py:103-130).
 *
 * Parity pin: tests/test_oracle_golden.py replays the reference's own assets through this file
 * (assets/2048_{drul,random}_actions.svg frame-for-frame, README max-tile histograms exactly).
 *
 * Array formats are those of include/g2048.h so tests compare buffers byte-for-byte:
 *   boards u8[B][16] log2 tiles row-major; masks u8[B] bit a = legal[a]; done u8[B];
 *   actions i32[B]; keys u32[B][2]; rewards f32[B].   actions: 0 left, 1 up, 2 right, 3 down.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define MODE_LEGACY 0
#define MODE_PARTITIONABLE 1

/* ---------------------------------------------------------------- threefry2x32, 20 rounds */
static inline uint32_t rotl32(uint32_t x, int r) { return (x << r) | (x >> (32 - r)); }

void orc_threefry(uint32_t k0, uint32_t k1, uint32_t c0, uint32_t c1, uint32_t *o0, uint32_t *o1) {
    static const int R[2][4] = {{13, 15, 26, 6}, {17, 29, 16, 24}};
    uint32_t ks[3] = {k0, k1, k0 ^ k1 ^ 0x1BD11BDAu};
    uint32_t x0 = c0 + ks[0], x1 = c1 + ks[1];
    for (int g = 0; g < 5; ++g) {
        for (int i = 0; i < 4; ++i) {
            x0 += x1;
            x1 = rotl32(x1, R[g & 1][i]);
            x1 ^= x0;
        }
        x0 += ks[(g + 1) % 3];
        x1 += ks[(g + 2) % 3] + (uint32_t)(g + 1);
    }
    *o0 = x0;
    *o1 = x1;
}

/* jax.random.split(key, n)[j] */
static void split_at(const uint32_t key[2], int64_t n, int64_t j, int mode, uint32_t out[2]) {
    if (mode == MODE_PARTITIONABLE) {
        orc_threefry(key[0], key[1], 0u, (uint32_t)j, &out[0], &out[1]);
        return;
    }
    /* legacy: flat = [b_0.o0 .. b_{n-1}.o0, b_0.o1 .. b_{n-1}.o1], b_i = TF(key,(i, n+i)) */
    for (int w = 0; w < 2; ++w) {
        int64_t f = 2 * j + w;
        uint32_t a, b;
        int64_t i = f < n ? f : f - n;
        orc_threefry(key[0], key[1], (uint32_t)i, (uint32_t)(n + i), &a, &b);
        out[w] = f < n ? a : b;
    }
}

void orc_split(const uint32_t key[2], uint32_t *out, int64_t n, int mode) {
    for (int64_t j = 0; j < n; ++j) split_at(key, n, j, mode, out + 2 * j);
}

/* n times: key, sub = split(key); subs[i] = sub.  key is updated in place. */
void orc_chain(uint32_t key[2], uint32_t *subs, int64_t n, int mode) {
    for (int64_t i = 0; i < n; ++i) {
        uint32_t two[4];
        orc_split(key, two, 2, mode);
        key[0] = two[0];
        key[1] = two[1];
        subs[2 * i] = two[2];
        subs[2 * i + 1] = two[3];
    }
}

static uint32_t bits_scalar(const uint32_t key[2], int mode) {
    uint32_t a, b;
    orc_threefry(key[0], key[1], 0u, 0u, &a, &b);
    return mode == MODE_PARTITIONABLE ? (a ^ b) : a;
}

static void bits_vec4(const uint32_t key[2], int mode, uint32_t out[4]) {
    if (mode == MODE_PARTITIONABLE) {
        for (uint32_t i = 0; i < 4; ++i) {
            uint32_t a, b;
            orc_threefry(key[0], key[1], 0u, i, &a, &b);
            out[i] = a ^ b;
        }
    } else {
        uint32_t p0, p1, q0, q1;
        orc_threefry(key[0], key[1], 0u, 2u, &p0, &p1);
        orc_threefry(key[0], key[1], 1u, 3u, &q0, &q1);
        out[0] = p0; out[1] = q0; out[2] = p1; out[3] = q1;
    }
}

static float uniform_f32(uint32_t bits) {
    uint32_t u = (bits >> 9) | 0x3F800000u;
    float f;
    memcpy(&f, &u, 4);
    return f - 1.0f;
}

/* correctly rounded f32 log (via f64): used where the reference takes log of 1/n (act_randomly) */
static float log_f32(float x) { return (float)log((double)x); }

/* f32 log evaluated entirely in IEEE f32 mul/add (Cephes logf polynomial).  jax draws its Gumbel noise
 * with XLA's f32 log; this is the faithful f32 log all three implementations (this file, the numpy
 * oracle, the HIP kernels) evaluate operation-for-operation, so they agree bit-for-bit.  The compile
 * flags (-ffp-contract=off) and the volatile temporaries keep every operation individually rounded. */
static float log_f32_poly(float x) {
    static const float P[9] = {7.0376836292E-2f, -1.1514610310E-1f, 1.1676998740E-1f, -1.2420140846E-1f,
                               1.4249322787E-1f, -1.6668057665E-1f, 2.0000714765E-1f, -2.4999993993E-1f,
                               3.3333331174E-1f};
    uint32_t bits;
    memcpy(&bits, &x, 4);
    int e = (int)((bits >> 23) & 0xFFu) - 126;
    uint32_t mb = (bits & 0x007FFFFFu) | 0x3F000000u;
    volatile float m, z, y, t;
    float mf;
    memcpy(&mf, &mb, 4);
    m = mf;
    if (m < 0.707106781186547524f) { e -= 1; t = m + m; m = t + -1.0f; }
    else { m = m + -1.0f; }
    z = m * m;
    y = P[0];
    for (int i = 1; i < 9; ++i) { t = y * m; y = t + P[i]; }
    t = y * m;
    y = t * z;
    const float fe = (float)e;
    t = -2.12194440e-4f * fe;
    y = y + t;
    t = -0.5f * z;
    y = y + t;
    t = m + y;
    z = 0.693359375f * fe;
    return t + z;
}

static float gumbel_f32(uint32_t bits) {
    const float tiny = 1.17549435e-38f;
    volatile float f = uniform_f32(bits) * 1.0f + tiny;
    float u = f > tiny ? f : tiny;
    return -log_f32_poly(-log_f32_poly(u));
}

static int categorical4(const uint32_t key[2], const float logits[4], int mode) {
    uint32_t bits[4];
    bits_vec4(key, mode, bits);
    int best = 0;
    float bv = 0.f;
    for (int i = 0; i < 4; ++i) {
        volatile float s = gumbel_f32(bits[i]) + logits[i];
        if (i == 0 || s > bv) { bv = s; best = i; }
    }
    return best;
}

/* ---------------------------------------------------------------- board arithmetic */
static float slide_row_left(uint8_t r[4]) {
    uint8_t t[4] = {0, 0, 0, 0};
    int n = 0;
    float score = 0.f;
    for (int i = 0; i < 4; ++i) if (r[i]) t[n++] = r[i];
    for (int i = 0; i + 1 < 4; ++i) {
        if (t[i] && t[i] == t[i + 1]) {
            t[i] += 1;
            t[i + 1] = 0;
            score += (float)(1u << t[i]);
        }
    }
    n = 0;
    uint8_t o[4] = {0, 0, 0, 0};
    for (int i = 0; i < 4; ++i) if (t[i]) o[n++] = t[i];
    memcpy(r, o, 4);
    return score;
}

/* cell index of the i-th element of line l when sliding in direction a (element 0 = wall side) */
static int line_cell(int a, int l, int i) {
    switch (a) {
        case 0: return 4 * l + i;          /* left:  rows, left -> right */
        case 1: return 4 * i + l;          /* up:    columns, top -> bottom */
        case 2: return 4 * l + (3 - i);    /* right: rows, right -> left */
        default: return 4 * (3 - i) + l;   /* down:  columns, bottom -> top */
    }
}

static float move_board(uint8_t b[16], int a) {
    float score = 0.f;
    for (int l = 0; l < 4; ++l) {
        uint8_t r[4];
        for (int i = 0; i < 4; ++i) r[i] = b[line_cell(a, l, i)];
        score += slide_row_left(r);
        for (int i = 0; i < 4; ++i) b[line_cell(a, l, i)] = r[i];
    }
    return score;
}

static uint8_t legal_bits(const uint8_t b[16]) {
    uint8_t m = 0;
    for (int a = 0; a < 4; ++a) {
        uint8_t t[16];
        memcpy(t, b, 16);
        move_board(t, a);
        if (memcmp(t, b, 16) != 0) m |= (uint8_t)(1u << a);
    }
    return m;
}

static void spawn(uint8_t b[16], const uint32_t key[2], int mode) {
    uint32_t two[4];
    orc_split(key, two, 2, mode);
    const uint32_t *kpos = two, *kval = two + 2;
    float c[16], acc = 0.f;
    for (int i = 0; i < 16; ++i) { acc += (b[i] == 0) ? 1.0f : 0.0f; c[i] = acc; }
    float u = uniform_f32(bits_scalar(kpos, mode));
    volatile float r = c[15] * (1.0f - u);
    int pos = 0;
    while (pos < 16 && c[pos] < r) ++pos;   /* searchsorted(c, r, side="left") */
    if (pos > 15) pos = 15;
    float u2 = uniform_f32(bits_scalar(kval, mode));
    float r2 = 1.0f - u2;
    b[pos] = (r2 <= 0.9f) ? 1 : 2;
}

void orc_init(const uint32_t *keys, uint8_t *boards, uint8_t *masks, uint8_t *done, int64_t B, int mode) {
#pragma omp parallel for schedule(static)
    for (int64_t e = 0; e < B; ++e) {
        uint32_t two[4];
        orc_split(keys + 2 * e, two, 2, mode);
        uint8_t *b = boards + 16 * e;
        memset(b, 0, 16);
        spawn(b, two, mode);
        spawn(b, two + 2, mode);
        masks[e] = legal_bits(b);
        done[e] = 0;
    }
}

static float step_one(uint8_t *b, uint8_t *mask, uint8_t *done, int a, const uint32_t key[2], int mode) {
    if (*done) return 0.0f;              /* frozen: board/mask unchanged, reward 0 */
    int illegal = !((*mask >> a) & 1);
    float reward = move_board(b, a);
    spawn(b, key, mode);
    uint8_t m = legal_bits(b);
    int term = (m == 0);
    if (illegal) { term = 1; reward = -1.0f; }
    if (term) m = 0xF;
    *mask = m;
    *done = (uint8_t)term;
    return reward;
}

void orc_step(uint8_t *boards, uint8_t *masks, uint8_t *done, const int32_t *actions,
              const uint32_t *keys, float *rewards, int64_t B, int mode) {
#pragma omp parallel for schedule(static)
    for (int64_t e = 0; e < B; ++e)
        rewards[e] = step_one(boards + 16 * e, masks + e, done + e, actions[e] & 3, keys + 2 * e, mode);
}

/* ---------------------------------------------------------------- act_fn plug-ins */
static int drul_one(uint8_t m) {
    for (int a = 3; a >= 0; --a) if ((m >> a) & 1) return a;
    return 3;
}

void orc_act_drul(const uint8_t *masks, int32_t *actions, int64_t B) {
    for (int64_t e = 0; e < B; ++e) actions[e] = drul_one(masks[e]);
}

static int random_one(const uint32_t key[2], uint8_t m, int mode, float *logp) {
    int n = 0;
    for (int a = 0; a < 4; ++a) n += (m >> a) & 1;
    float probs[4], logits[4];
    for (int a = 0; a < 4; ++a) {
        probs[a] = n > 0 ? (float)((m >> a) & 1) / (float)n : 0.25f;
        float l = probs[a] > 0.f ? log_f32(probs[a]) : -INFINITY;
        logits[a] = l > -3.40282347e38f ? l : -3.40282347e38f;
    }
    int act = categorical4(key, logits, mode);
    *logp = probs[act] > 0.f ? log_f32(probs[act]) : -INFINITY;
    return act;
}

void orc_act_random(const uint32_t *keys, const uint8_t *masks, int32_t *actions, float *logp,
                    int64_t B, int mode) {
#pragma omp parallel for schedule(static)
    for (int64_t e = 0; e < B; ++e) actions[e] = random_one(keys + 2 * e, masks[e], mode, logp + e);
}

/* TorchActionFunction tail (reference src/ppo/torch_action_wrapper.py:85-102).
 * logits f32[B][4] are the agent's raw actor outputs; if use_mask, the agent's own masking
 * (src/ppo/ppo_agent.py:117-121: logits - 1e8*(1-mask)) is applied first, in f32. */
void orc_act_logits(const uint32_t *keys, const float *logits, const uint8_t *masks, int use_mask,
                    int sample, int32_t *actions, float *logp, int64_t B, int mode) {
#pragma omp parallel for schedule(static)
    for (int64_t e = 0; e < B; ++e) {
        float l[4];
        for (int a = 0; a < 4; ++a) {
            volatile float v = logits[4 * e + a];
            if (use_mask) v = v - 1e8f * (1.0f - (float)((masks[e] >> a) & 1));
            l[a] = v > -3.40282347e38f ? v : -3.40282347e38f;
        }
        int act = 0;
        if (sample) act = categorical4(keys + 2 * e, l, mode);
        else for (int a = 1; a < 4; ++a) if (l[a] > l[act]) act = a;
        double mx = l[0];
        for (int a = 1; a < 4; ++a) if (l[a] > mx) mx = l[a];
        double s = 0;
        for (int a = 0; a < 4; ++a) s += exp((double)l[a] - mx);
        actions[e] = act;
        volatile float lse = (float)(mx + log(s)); /* f32 logsumexp, as jax computes it */
        logp[e] = l[act] - lse;
    }
}

/* ---------------------------------------------------------------- fused lock-step rollout
 * BatchRunner.run_actions_batch semantics for the two naive policies (policy 0 = drul, 1 = random)
 * with the reference key schedule; every env runs to ITS termination (lanes are independent: keys
 * are batch-wide splits).  Used as the CPU baseline and as a whole-episode checker.
 * Outputs per env: ep_len (steps through first termination), final board, sum of rewards.
 * Returns total live env-steps.  chain_subs = orc_chain() output, [1 + 2*max_steps][2].
 */
int64_t orc_rollout(const uint32_t *chain_subs, int64_t max_steps, int64_t B_total, int64_t e0,
                    int64_t B, int policy, int mode, uint8_t *final_boards, int32_t *ep_len,
                    float *ep_return) {
    int64_t total = 0;
#pragma omp parallel for schedule(static) reduction(+ : total)
    for (int64_t i = 0; i < B; ++i) {
        int64_t g = e0 + i;
        uint32_t k[2];
        uint8_t b[16], mask, done = 0;
        split_at(chain_subs, B_total, g, mode, k);
        {
            uint32_t two[4];
            orc_split(k, two, 2, mode);
            memset(b, 0, 16);
            spawn(b, two, mode);
            spawn(b, two + 2, mode);
            mask = legal_bits(b);
        }
        float ret = 0.f;
        int64_t t = 0;
        while (!done && t < max_steps) {
            int a;
            float lp;
            if (policy == 0) a = drul_one(mask);
            else {
                split_at(chain_subs + 2 * (1 + 2 * t), B_total, g, mode, k);
                a = random_one(k, mask, mode, &lp);
            }
            split_at(chain_subs + 2 * (2 + 2 * t), B_total, g, mode, k);
            ret += step_one(b, &mask, &done, a, k, mode);
            ++t;
        }
        memcpy(final_boards + 16 * i, b, 16);
        ep_len[i] = done ? (int32_t)t : 0;
        ep_return[i] = ret;
        total += t;
    }
    return total;
}

/* ---------------------------------------------------------------- GAE (flat buffer) */
void orc_gae(const float *r, const float *v, const uint8_t *term, float *adv, float *ret, int64_t N,
             double gamma, double lam) {
    const float g = (float)gamma, gl = (float)(gamma * lam);
    volatile float last_gae = 0.f, last_v = 0.f, t1, delta;
    for (int64_t i = N - 1; i >= 0; --i) {
        if (term[i]) { last_v = 0.f; last_gae = 0.f; }
        t1 = g * last_v;
        t1 = r[i] + t1;
        delta = t1 - v[i];
        t1 = gl * last_gae;
        last_gae = delta + t1;
        adv[i] = last_gae;
        ret[i] = last_gae + v[i];
        last_v = v[i];
    }
}

void orc_set_num_threads(int n) {
    if (n > 0) omp_set_num_threads(n);
}

int orc_num_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}
