// Per-lane primitives of the 2048 board step for gfx950: one board per wavefront lane, held as
// four 32-bit VGPRs (row r = one dword, cell (r,c) = byte c, value = log2(tile), 0 = empty).
//
// Everything here is branch-free SWAR on those four dwords plus threefry2x32 for the JAX-compatible
// counter RNG.  What each piece replaces in the reference (paths relative to the reference repo):
//   tf2x32 / split / bits      jax.random.key/split/uniform/categorical as called from
//                              src/runs/batch_runner.py:32,105-106,118-119,126-127 and
//                              src/actions/act_randomly.py:48
//   board_move/legal/spawn     pgx.make("2048") init/step, src/runs/batch_runner.py:33-35,107,128
//   policy_drul/random/logits  src/actions/act_drul.py:40-44, src/actions/act_randomly.py:40-54,
//                              src/ppo/torch_action_wrapper.py:85-102
//
// The file also compiles as plain host C++ (G2048_HOST_TEST) so tests/ can exercise the SWAR logic
// exhaustively without a GPU; the product never runs that build.
#pragma once
#include <stdint.h>

#if defined(__HIPCC__) && !defined(G2048_HOST_TEST)
#include <hip/hip_runtime.h>
#define G_DEV __host__ __device__ __forceinline__
#else
#define G_DEV static inline
#endif
#include <math.h>
#include <string.h>
// device-only intrinsics are used in the device pass; the host pass (threefry for the host-side key
// chain) and the G2048_HOST_TEST build take the portable spelling of the same operation
#if defined(__HIP_DEVICE_COMPILE__) && !defined(G2048_HOST_TEST)
#define G2048_ON_DEVICE 1
#else
#define G2048_ON_DEVICE 0
#endif

namespace g2048 {

typedef uint32_t u32;

enum { RNG_LEGACY = 0, RNG_PARTITIONABLE = 1 };

struct Board {
    u32 r[4];
};

// ------------------------------------------------------------------------------------------------
// small helpers
// ------------------------------------------------------------------------------------------------
G_DEV u32 rotl(u32 x, int r) {
#if G2048_ON_DEVICE
    return __builtin_rotateleft32(x, r);
#else
    return (x << r) | (x >> (32 - r));
#endif
}

// v_perm_b32: result byte i is chosen by selector byte i from the 8-byte pool {lo: 0..3, hi: 4..7};
// selector 8/9/10/11 = sign of pool byte 1/3/5/7 replicated, 12 = 0x00, >= 13 = 0xFF.
G_DEV u32 perm(u32 hi, u32 lo, u32 sel) {
#if G2048_ON_DEVICE
    return __builtin_amdgcn_perm(hi, lo, sel);
#else
    uint64_t pool = ((uint64_t)hi << 32) | lo;
    u32 out = 0;
    for (int i = 0; i < 4; ++i) {
        const u32 s = (sel >> (8 * i)) & 0xFF;
        u32 v;
        if (s < 8) v = (u32)((pool >> (8 * s)) & 0xFF);
        else if (s < 12) v = ((pool >> (8 * (2 * (s - 8) + 1) + 7)) & 1) ? 0xFFu : 0u;
        else v = s == 12 ? 0u : 0xFFu;
        out |= v << (8 * i);
    }
    return out;
#endif
}

G_DEV u32 popc(u32 x) {
#if G2048_ON_DEVICE
    return __popc(x);
#else
    return (u32)__builtin_popcount(x);
#endif
}

G_DEV float u32_as_float(u32 x) {
#if G2048_ON_DEVICE
    return __uint_as_float(x);
#else
    float f;
    memcpy(&f, &x, 4);
    return f;
#endif
}

// ------------------------------------------------------------------------------------------------
// threefry2x32, 20 rounds
// ------------------------------------------------------------------------------------------------
G_DEV void tf2x32(u32 k0, u32 k1, u32 c0, u32 c1, u32 &o0, u32 &o1) {
    const u32 k2 = k0 ^ k1 ^ 0x1BD11BDAu;
    u32 x0 = c0 + k0, x1 = c1 + k1;
#define G2048_R(r) x0 += x1; x1 = rotl(x1, r); x1 ^= x0;
    G2048_R(13) G2048_R(15) G2048_R(26) G2048_R(6)
    x0 += k1; x1 += k2 + 1u;
    G2048_R(17) G2048_R(29) G2048_R(16) G2048_R(24)
    x0 += k2; x1 += k0 + 2u;
    G2048_R(13) G2048_R(15) G2048_R(26) G2048_R(6)
    x0 += k0; x1 += k1 + 3u;
    G2048_R(17) G2048_R(29) G2048_R(16) G2048_R(24)
    x0 += k1; x1 += k2 + 4u;
    G2048_R(13) G2048_R(15) G2048_R(26) G2048_R(6)
    x0 += k2; x1 += k0 + 5u;
#undef G2048_R
    o0 = x0;
    o1 = x1;
}

// jax.random.split(key, n)[j]
template <int MODE>
G_DEV void split_at(u32 k0, u32 k1, u32 n, u32 j, u32 &o0, u32 &o1) {
    if (MODE == RNG_PARTITIONABLE) {
        tf2x32(k0, k1, 0u, j, o0, o1);
    } else {
        // flat = [b_0.o0 .. b_{n-1}.o0, b_0.o1 .. b_{n-1}.o1], b_i = TF(key,(i, n+i)); out = flat[2j], flat[2j+1]
        u32 f0 = 2u * j, f1 = 2u * j + 1u, a, b;
        u32 i0 = f0 < n ? f0 : f0 - n;
        tf2x32(k0, k1, i0, n + i0, a, b);
        o0 = f0 < n ? a : b;
        u32 i1 = f1 < n ? f1 : f1 - n;
        tf2x32(k0, k1, i1, n + i1, a, b);
        o1 = f1 < n ? a : b;
    }
}

// ka, kb = jax.random.split(key)
template <int MODE>
G_DEV void split2(u32 k0, u32 k1, u32 &a0, u32 &a1, u32 &b0, u32 &b1) {
    if (MODE == RNG_PARTITIONABLE) {
        tf2x32(k0, k1, 0u, 0u, a0, a1);
        tf2x32(k0, k1, 0u, 1u, b0, b1);
    } else {
        u32 p0, p1, q0, q1;
        tf2x32(k0, k1, 0u, 2u, p0, p1);
        tf2x32(k0, k1, 1u, 3u, q0, q1);
        a0 = p0; a1 = q0; b0 = p1; b1 = q1;
    }
}

// 32 random bits, shape ()
template <int MODE>
G_DEV u32 bits_scalar(u32 k0, u32 k1) {
    u32 a, b;
    tf2x32(k0, k1, 0u, 0u, a, b);
    return MODE == RNG_PARTITIONABLE ? (a ^ b) : a;
}

// 32 random bits, shape (4,)
template <int MODE>
G_DEV void bits_vec4(u32 k0, u32 k1, u32 out[4]) {
    if (MODE == RNG_PARTITIONABLE) {
        for (u32 i = 0; i < 4; ++i) {
            u32 a, b;
            tf2x32(k0, k1, 0u, i, a, b);
            out[i] = a ^ b;
        }
    } else {
        u32 p0, p1, q0, q1;
        tf2x32(k0, k1, 0u, 2u, p0, p1);
        tf2x32(k0, k1, 1u, 3u, q0, q1);
        out[0] = p0; out[1] = q0; out[2] = p1; out[3] = q1;
    }
}

// ------------------------------------------------------------------------------------------------
// board arithmetic
// ------------------------------------------------------------------------------------------------
// 0xFF in every byte of x that holds a tile, 0x00 in empty bytes (cell values <= 0x80).
// x + 0x7F.. puts "non-zero" into bit 7 of each byte; v_perm's sign-replicating selectors widen bit 7 of
// pool bytes 1/3/5/7 to full bytes, so one shift + one perm finish the job.
G_DEV u32 nz_mask(u32 x) {
    const u32 y = x + 0x7F7F7F7Fu;
    return perm(y << 8, y, 0x090B080Au);
}

// (m & x) | (~m & y)  -> v_bfi_b32
G_DEV u32 bfi(u32 m, u32 x, u32 y) { return (m & x) | (~m & y); }

// Slide the four columns of the board toward row register a (2048 merge rules), all four columns at once:
// byte lane c of (a, b, c, d) is column c from the wall outward.  Returns the merge score of the four lines.
G_DEV u32 slide_lines(u32 &a, u32 &b, u32 &c, u32 &d) {
    // squeeze out empty cells, outermost gap first (afterwards every line is packed against a)
    u32 m = nz_mask(c);
    c = bfi(m, c, d);  d &= m;
    m = nz_mask(b);
    b = bfi(m, b, c);  c = bfi(m, c, d);  d &= m;
    m = nz_mask(a);
    a = bfi(m, a, b);  b = bfi(m, b, c);  c = bfi(m, c, d);  d &= m;
    // a tile merges with its outer neighbour when equal; a merged tile does not merge again (left-to-right scan)
    const u32 e01 = ~nz_mask(a ^ b) & nz_mask(b);
    const u32 e12 = ~nz_mask(b ^ c) & nz_mask(c) & ~e01;
    const u32 e23 = ~nz_mask(c ^ d) & nz_mask(d) & ~e12;
    const u32 both = e01 & e23, ones = 0x01010101u;
    const u32 na = a + (e01 & ones);                                   // [a+1 | a]
    const u32 nb = bfi(e01, c, b) + ((e12 | both) & ones);             // [c(+1) | b+1 | b]
    const u32 nc = (bfi(e01 | e12, d, c) + (e23 & ~e01 & ones)) & ~both;  // [0 | d | c+1 | c]
    const u32 nd = d & ~(e01 | e12 | e23);
    // score: every merge creates one tile 2^e, e = the new exponent (>= 2)
    const u32 first = (na & e01) | (nb & e12) | (nc & e23 & ~e01);   // one merge per line ...
    const u32 second = nb & both;                                     // ... plus the (c,d) merge when both fire
    const u32 vf = (e01 | e12 | e23) & ones, vs = both & ones;
    u32 score = 0;
    for (int i = 0; i < 4; ++i) {
        score += ((vf >> (8 * i)) & 0xFFu) << ((first >> (8 * i)) & 0xFFu);
        score += ((vs >> (8 * i)) & 0xFFu) << ((second >> (8 * i)) & 0xFFu);
    }
    a = na; b = nb; c = nc; d = nd;
    return score;
}

// 4x4 byte transpose of the four row dwords (8 v_perm_b32)
G_DEV void transpose(Board &bd) {
    const u32 t0 = perm(bd.r[1], bd.r[0], 0x05010400u);  // [r0.b0 r1.b0 r0.b1 r1.b1]
    const u32 t1 = perm(bd.r[1], bd.r[0], 0x07030602u);  // [r0.b2 r1.b2 r0.b3 r1.b3]
    const u32 t2 = perm(bd.r[3], bd.r[2], 0x05010400u);
    const u32 t3 = perm(bd.r[3], bd.r[2], 0x07030602u);
    bd.r[0] = perm(t2, t0, 0x05040100u);
    bd.r[1] = perm(t2, t0, 0x07060302u);
    bd.r[2] = perm(t3, t1, 0x05040100u);
    bd.r[3] = perm(t3, t1, 0x07060302u);
}

// Slide/merge the whole board in direction a (0 left, 1 up, 2 right, 3 down); returns merge score.
// Divergence-free: horizontal moves are the vertical move of the transposed board, right/down are the
// left/up move with the line order reversed -- the direction only selects permutations around one
// column-parallel slide.
G_DEV u32 board_move(Board &bd, u32 a) {
    const bool horizontal = (a & 1u) == 0, reverse = (a & 2u) != 0;
    Board t = bd;
    transpose(t);
    for (int i = 0; i < 4; ++i) t.r[i] = horizontal ? t.r[i] : bd.r[i];
    u32 x0 = reverse ? t.r[3] : t.r[0], x1 = reverse ? t.r[2] : t.r[1];
    u32 x2 = reverse ? t.r[1] : t.r[2], x3 = reverse ? t.r[0] : t.r[3];
    const u32 score = slide_lines(x0, x1, x2, x3);
    t.r[0] = reverse ? x3 : x0; t.r[1] = reverse ? x2 : x1;
    t.r[2] = reverse ? x1 : x2; t.r[3] = reverse ? x0 : x3;
    Board u = t;
    transpose(u);
    for (int i = 0; i < 4; ++i) bd.r[i] = horizontal ? u.r[i] : t.r[i];
    return score;
}

// bit 7 of every byte that holds a tile (valid for cell values <= 0x80)
G_DEV u32 tile_bits(u32 row) { return (row + 0x7F7F7F7Fu) & 0x80808080u; }

// legal_action_mask: bit a set iff moving in direction a changes the board.
G_DEV u32 board_legal(const Board &bd) {
    u32 nz[4], s[4];
    for (int i = 0; i < 4; ++i) {
        nz[i] = tile_bits(bd.r[i]);
        // give empty cells checkerboard sentinels (0x80 / 0xC0) so they never equal a neighbour
        const u32 z = nz[i] ^ 0x80808080u;
        const u32 chk = (i & 1) ? 0x00400040u : 0x40004000u;
        s[i] = bd.r[i] | z | ((z >> 1) & chk);
    }
    // mergeable neighbours: a zero byte in s ^ neighbour
    u32 hz = 0, vz = 0;
    for (int i = 0; i < 4; ++i) {
        const u32 y = s[i] ^ (s[i] >> 8);  // byte 3 = s.b3 != 0
        hz |= (y - 0x01010101u) & ~y;
    }
    for (int i = 0; i < 3; ++i) {
        const u32 y = s[i] ^ s[i + 1];
        vz |= (y - 0x01010101u) & ~y;
    }
    const bool hmerge = (hz & 0x80808080u) != 0, vmerge = (vz & 0x80808080u) != 0;
    // occupancy: byte c of P holds column c, bit r = tile at (r, c)
    const u32 P = (nz[0] >> 7) | (nz[1] >> 6) | (nz[2] >> 5) | (nz[3] >> 4);
    const u32 Z = ~P & 0x0F0F0F0Fu;
    const bool up_gap = ((P + 0x01010101u) & P) != 0;                                   // empty above a tile
    const bool down_gap = ((((P | 0x10101010u) - 0x01010101u) | P) & 0x0F0F0F0Fu) != 0x0F0F0F0Fu;
    const bool left_gap = (Z & ((P >> 8) | (P >> 16) | (P >> 24))) != 0;                // empty left of a tile
    const bool right_gap = (Z & ((P << 8) | (P << 16) | (P << 24))) != 0;
    return (u32)(left_gap | hmerge) | ((u32)(up_gap | vmerge) << 1) | ((u32)(right_gap | hmerge) << 2) |
           ((u32)(down_gap | vmerge) << 3);
}

// jax.random.choice spawn: uniformly chosen empty cell (row-major order), value 1 (p=.9) or 2.
template <int MODE>
G_DEV void board_spawn(Board &bd, u32 k0, u32 k1) {
#ifdef G2048_RNG_STUB
    // Diagnostic build only (tools/step_rng_floor.py; never compiled into the product library): the two random words of a
    // spawn are taken from the key as they are instead of through split + 2 x bits = four threefry2x32 blocks.  Same bytes
    // per env-step, board logic untouched: what k_step would cost if the reference's RNG schedule were free.
    const u32 bpos = k0, bval = k1;
#else
    u32 p0, p1, v0, v1;
    split2<MODE>(k0, k1, p0, p1, v0, v1);
    const u32 bpos = bits_scalar<MODE>(p0, p1);
    const u32 bval = bits_scalar<MODE>(v0, v1);
#endif
    // empties per row and running totals (row-major)
    u32 z[4], c[4];
    for (int i = 0; i < 4; ++i) z[i] = tile_bits(bd.r[i]) ^ 0x80808080u;
    c[0] = popc(z[0]);
    c[1] = c[0] + popc(z[1]);
    c[2] = c[1] + popc(z[2]);
    c[3] = c[2] + popc(z[3]);
    // r = c[15] * (1 - u) in f32, pos = first index with cumsum >= r  <=>  the ceil(r)-th empty cell
    const float one_minus_u = 2.0f - u32_as_float((bpos >> 9) | 0x3F800000u);  // exact: 1 - (f - 1)
#if G2048_ON_DEVICE
    const float rr = __fmul_rn((float)c[3], one_minus_u);
#else
    volatile float rr_v = (float)c[3] * one_minus_u;
    const float rr = rr_v;
#endif
    const u32 k = (u32)ceilf(rr);  // 0 only when the board is full (illegal move on a full board): cell 0
    const u32 row = (u32)(k > c[0]) + (u32)(k > c[1]) + (u32)(k > c[2]);
    const u32 before = row == 0 ? 0u : (row == 1 ? c[0] : (row == 2 ? c[1] : c[2]));
    const u32 zr = row == 0 ? z[0] : (row == 1 ? z[1] : (row == 2 ? z[2] : z[3]));
    const u32 kk = k - before;  // kk-th empty cell of that row (1-based; 0 -> column 0)
    const u32 f0 = (zr >> 7) & 1u, f1 = f0 + ((zr >> 15) & 1u), f2 = f1 + ((zr >> 23) & 1u);
    const u32 col = (u32)(kk > f0) + (u32)(kk > f1) + (u32)(kk > f2);
    // 1 - u2 <= 0.9f  <=>  (bits >> 9) >= 838861  (1 - 0.9f = 838861 * 2^-23 exactly)
    const u32 val = ((bval >> 9) >= 838861u) ? 1u : 2u;
    const u32 sh = 8u * col;
    for (int i = 0; i < 4; ++i) {
        const u32 nv = (bd.r[i] & ~(0xFFu << sh)) | (val << sh);
        bd.r[i] = (row == (u32)i) ? nv : bd.r[i];
    }
}

// pgx init: two spawns on an empty board; mask = true legal mask of that board.
template <int MODE>
G_DEV void env_init(Board &bd, u32 &mask, u32 k0, u32 k1) {
    u32 a0, a1, b0, b1;
    split2<MODE>(k0, k1, a0, a1, b0, b1);
    bd.r[0] = bd.r[1] = bd.r[2] = bd.r[3] = 0;
    board_spawn<MODE>(bd, a0, a1);
    board_spawn<MODE>(bd, b0, b1);
    mask = board_legal(bd);
}

// pgx step incl. wrapper semantics. done lanes are frozen: reward 0.
template <int MODE>
G_DEV float env_step(Board &bd, u32 &mask, u32 &done, u32 a, u32 k0, u32 k1) {
    Board nb = bd;
    const u32 score = board_move(nb, a);
    board_spawn<MODE>(nb, k0, k1);
    u32 m = board_legal(nb);
    const bool illegal = ((mask >> a) & 1u) == 0;
    const bool term = (m == 0) | illegal;
    float reward = illegal ? -1.0f : (float)score;
    m = term ? 0xFu : m;
    const bool frozen = done != 0;
    for (int i = 0; i < 4; ++i) bd.r[i] = frozen ? bd.r[i] : nb.r[i];
    mask = frozen ? mask : m;
    done = frozen ? done : (u32)term;
    return frozen ? 0.0f : reward;
}

// ------------------------------------------------------------------------------------------------
// policies
// ------------------------------------------------------------------------------------------------
G_DEV u32 policy_drul(u32 mask) {
    return (mask & 8u) ? 3u : ((mask & 4u) ? 2u : ((mask & 2u) ? 1u : ((mask & 1u) ? 0u : 3u)));
}

// Exactly-rounded single f32 multiply / add (never contracted into an FMA), so the host oracle and the
// device evaluate the same sequence of IEEE operations.
G_DEV float mul_rn(float a, float b) {
#if G2048_ON_DEVICE
    return __fmul_rn(a, b);
#else
    volatile float r = a * b;
    return r;
#endif
}
G_DEV float add_rn(float a, float b) {
#if G2048_ON_DEVICE
    return __fadd_rn(a, b);
#else
    volatile float r = a + b;
    return r;
#endif
}

// Natural log of a positive normal f32 in pure f32 arithmetic (Cephes logf polynomial, < 1 ulp typical).
// jax computes the Gumbel noise with XLA's f32 log; any faithful f32 log reproduces the reference's
// assets (tests/test_oracle_golden.py).  The oracle evaluates the identical operation sequence, so device
// and oracle agree bit-for-bit; an f64 log here made the fused random-policy kernel latency-bound.
G_DEV float log_f32(float x) {
    u32 bits;
#if G2048_ON_DEVICE
    bits = __float_as_uint(x);
#else
    memcpy(&bits, &x, 4);
#endif
    int e = (int)((bits >> 23) & 0xFFu) - 126;
    float m = u32_as_float((bits & 0x007FFFFFu) | 0x3F000000u);  // [0.5, 1)
    if (m < 0.707106781186547524f) {
        e -= 1;
        m = add_rn(add_rn(m, m), -1.0f);
    } else {
        m = add_rn(m, -1.0f);
    }
    const float z = mul_rn(m, m);
    float y = 7.0376836292E-2f;
    y = add_rn(mul_rn(y, m), -1.1514610310E-1f);
    y = add_rn(mul_rn(y, m), 1.1676998740E-1f);
    y = add_rn(mul_rn(y, m), -1.2420140846E-1f);
    y = add_rn(mul_rn(y, m), 1.4249322787E-1f);
    y = add_rn(mul_rn(y, m), -1.6668057665E-1f);
    y = add_rn(mul_rn(y, m), 2.0000714765E-1f);
    y = add_rn(mul_rn(y, m), -2.4999993993E-1f);
    y = add_rn(mul_rn(y, m), 3.3333331174E-1f);
    y = mul_rn(mul_rn(y, m), z);
    const float fe = (float)e;
    y = add_rn(y, mul_rn(-2.12194440e-4f, fe));
    y = add_rn(y, mul_rn(-0.5f, z));
    return add_rn(add_rn(m, y), mul_rn(0.693359375f, fe));
}

G_DEV float gumbel(u32 bits) {
    const float tiny = 1.17549435e-38f;
    const float f = u32_as_float((bits >> 9) | 0x3F800000u) - 1.0f;
    const float u = f > 0.0f ? f : tiny;  // max(tiny, f * (1 - tiny) + tiny) in f32
    return -log_f32(-log_f32(u));
}

// jax.random.categorical over 4 logits: argmax(gumbel + logits), first maximum wins
template <int MODE>
G_DEV u32 categorical4(u32 k0, u32 k1, const float l[4]) {
    u32 bits[4];
    bits_vec4<MODE>(k0, k1, bits);
    u32 best = 0;
#if G2048_ON_DEVICE
    float bv = __fadd_rn(gumbel(bits[0]), l[0]);
    for (u32 i = 1; i < 4; ++i) {
        const float s = __fadd_rn(gumbel(bits[i]), l[i]);
#else
    volatile float bv = gumbel(bits[0]) + l[0];
    for (u32 i = 1; i < 4; ++i) {
        volatile float s = gumbel(bits[i]) + l[i];
#endif
        if (s > bv) { bv = s; best = i; }
    }
    return best;
}

// act_randomly: uniform over legal actions through the same categorical draw; logp = log(1/n).
template <int MODE>
G_DEV u32 policy_random(u32 k0, u32 k1, u32 mask, float &logp) {
    const u32 n = popc(mask & 0xFu);
    const float FMIN = -3.40282347e38f;
    // log(f32(1/n)) for n = 1..4, and log(0.25) when nothing is legal
    const float lp = n == 1 ? 0.0f : (n == 2 ? -0x1.62e43p-1f : (n == 3 ? -0x1.193ea8p+0f : -0x1.62e43p+0f));
    float l[4];
    for (u32 a = 0; a < 4; ++a) l[a] = (n == 0 || ((mask >> a) & 1u)) ? lp : FMIN;
    logp = lp;
    return categorical4<MODE>(k0, k1, l);
}

// TorchActionFunction tail: optional agent masking (logits - 1e8*(1-mask), f32), clamp to finfo.min,
// categorical sample or argmax, logp = logit[a] - logsumexp(logits).
template <int MODE>
G_DEV u32 policy_logits(u32 k0, u32 k1, const float raw[4], u32 mask, bool use_mask, bool sample, float &logp) {
    const float FMIN = -3.40282347e38f;
    float l[4];
    for (u32 a = 0; a < 4; ++a) {
        float v = raw[a];
        if (use_mask) {
#if G2048_ON_DEVICE
            v = __fsub_rn(v, ((mask >> a) & 1u) ? 0.0f : 1e8f);
#else
            volatile float vv = v - (((mask >> a) & 1u) ? 0.0f : 1e8f);
            v = vv;
#endif
        }
        l[a] = v > FMIN ? v : FMIN;
    }
    u32 act = 0;
    if (sample) {
        act = categorical4<MODE>(k0, k1, l);
    } else {
        for (u32 a = 1; a < 4; ++a) act = l[a] > l[act] ? a : act;
    }
    float mx = fmaxf(fmaxf(l[0], l[1]), fmaxf(l[2], l[3]));
    float s = expf(l[0] - mx) + expf(l[1] - mx) + expf(l[2] - mx) + expf(l[3] - mx);
    const float la = act == 0 ? l[0] : (act == 1 ? l[1] : (act == 2 ? l[2] : l[3]));
    logp = la - (mx + logf(s));
    return act;
}

}  // namespace g2048
