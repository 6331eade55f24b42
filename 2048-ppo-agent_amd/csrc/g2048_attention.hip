// Attention forward/backward for the policy's tiny sequences (17 tokens, head_dim 32) in the PPO update.
//
// Replaces torch's scaled_dot_product_attention in training (reference: nn.TransformerEncoderLayer inside
// src/ppo/transformer_encoder.py:138-148, attention dropout 0.1).  The flash kernels PyTorch dispatches to spend
// 55 us forward and 217 us backward per layer at minibatch 2048 on problems of 17 x 17 scores per (sample, head);
// here one lane owns one query row (3 (sample, head) pairs per wavefront), K/V/Q/dO rows sit in LDS as f32, the
// 17 x 17 probabilities live in registers, and the whole thing is a few thousand VALU FMAs per lane: memory-bound.
// Dropout masks are recomputed in the backward pass from (seed, element index) instead of being stored.
//
// Layout: q/k/v/dq/dk/dv are bf16 with arbitrary (batch, token) strides and heads contiguous inside a token
// ([.., H, 32]), so they can point into the packed in_proj output / its gradient; o/dout are bf16 [B, Sq, H, 32]
// contiguous; lse f32 [B, H, Sq].  Sq = 17 (full layer) or 1 (CLS-only last layer), Sk = 17.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/g2048.h"

namespace {

constexpr int HD = 32, SK = 17, PAIRS = 3;  // (sample, head) pairs per wavefront when Sq = 17
constexpr int ROW = HD + 4;                  // f32 row stride in LDS (pad: pairs land on different banks)
constexpr int PSTRIDE = SK * ROW + 8;        // per-pair stride of a [17][32] f32 tile

struct Params {
    const uint16_t *q, *k, *v;
    int64_t B;
    int H;
    int64_t q_sb, q_ss, k_sb, k_ss, v_sb, v_ss;
    float scale, p_drop, inv_keep;
    uint32_t seed0, seed1, thr;
    const uint64_t *seed_state;  // optional device-resident word mixed into the seed (advanced between hipGraph replays)
};

__device__ __forceinline__ float bf2f(uint16_t b) { return __uint_as_float((uint32_t)b << 16); }
__device__ __forceinline__ uint16_t f2bf(float f) {
    const __bf16 b = (__bf16)f;
    return *reinterpret_cast<const uint16_t *>(&b);
}

// 32 bf16 (64 B) -> 32 f32
__device__ __forceinline__ void load_row(const uint16_t *p, float out[HD]) {
    const uint4 *p4 = reinterpret_cast<const uint4 *>(p);
    for (int c = 0; c < 4; ++c) {
        const uint4 u = p4[c];
        const uint32_t w[4] = {u.x, u.y, u.z, u.w};
        for (int e = 0; e < 4; ++e) {
            out[8 * c + 2 * e] = __uint_as_float(w[e] << 16);
            out[8 * c + 2 * e + 1] = __uint_as_float(w[e] & 0xFFFF0000u);
        }
    }
}
__device__ __forceinline__ void store_row(uint16_t *p, const float in[HD]) {
    uint4 *p4 = reinterpret_cast<uint4 *>(p);
    for (int c = 0; c < 4; ++c) {
        uint32_t w[4];
        for (int e = 0; e < 4; ++e) w[e] = (uint32_t)f2bf(in[8 * c + 2 * e]) | ((uint32_t)f2bf(in[8 * c + 2 * e + 1]) << 16);
        p4[c] = make_uint4(w[0], w[1], w[2], w[3]);
    }
}

__device__ __forceinline__ void mix_seed_state(Params &P) {
    if (P.seed_state) {
        const uint64_t s = *P.seed_state;
        P.seed0 ^= (uint32_t)s * 0x9E3779B1u;
        P.seed1 += (uint32_t)(s >> 32) * 0x85EBCA77u + (uint32_t)s;
    }
}

// Workgroup -> work item, XCD-aware.  The dispatcher deals consecutive workgroup ids round-robin over the 8 XCDs, each with its
// own L2, while consecutive (sample, head) pairs share cache lines (the 64-byte q/k/v rows of neighbouring heads are halves
// of one 128-byte line, the 8 heads of a token one 512-byte run).  Give every XCD a contiguous range of items instead: XCD x
// gets the workgroups b with b % 8 == x, count_x = n / 8 + (x < n % 8) of them, and they take the items
// [start_x, start_x + count_x) with start_x = x * (n / 8) + min(x, n % 8) - a bijection of [0, n).
// (rocprofv3 FETCH_SIZE before: 1.35x the algorithmic reads.)
__device__ __forceinline__ int64_t xcd_block() {
    const int64_t n = gridDim.x, q = n / 8, r = n % 8, x = blockIdx.x % 8;
    return x * q + (x < r ? x : r) + blockIdx.x / 8;
}

// dropout keep decision for probability element `idx`: a 32-bit mix of (seed, idx), deterministic across fwd/bwd
__device__ __forceinline__ bool keep_mask(const Params &P, uint64_t idx) {
    uint32_t x = (uint32_t)idx * 0x9E3779B1u ^ P.seed0;
    x ^= (uint32_t)(idx >> 32) * 0x85EBCA77u + P.seed1;
    x ^= x >> 16; x *= 0x7FEB352Du; x ^= x >> 15; x *= 0x846CA68Bu; x ^= x >> 16;
    return (x >> 8) >= P.thr;
}

__device__ __forceinline__ float dot_lds(const float a[HD], const float *row) {
    float s = 0.f;
    for (int c = 0; c < HD / 4; ++c) {
        const float4 kv = *reinterpret_cast<const float4 *>(row + 4 * c);
        s = fmaf(a[4 * c], kv.x, s); s = fmaf(a[4 * c + 1], kv.y, s);
        s = fmaf(a[4 * c + 2], kv.z, s); s = fmaf(a[4 * c + 3], kv.w, s);
    }
    return s;
}
__device__ __forceinline__ void axpy_lds(float acc[HD], float a, const float *row) {
    for (int c = 0; c < HD / 4; ++c) {
        const float4 kv = *reinterpret_cast<const float4 *>(row + 4 * c);
        acc[4 * c] = fmaf(a, kv.x, acc[4 * c]); acc[4 * c + 1] = fmaf(a, kv.y, acc[4 * c + 1]);
        acc[4 * c + 2] = fmaf(a, kv.z, acc[4 * c + 2]); acc[4 * c + 3] = fmaf(a, kv.w, acc[4 * c + 3]);
    }
}
__device__ __forceinline__ void put_row(float *row, const float v[HD]) {
    for (int c = 0; c < HD / 4; ++c) *reinterpret_cast<float4 *>(row + 4 * c) = make_float4(v[4 * c], v[4 * c + 1], v[4 * c + 2], v[4 * c + 3]);
}

// ------------------------------------------------------------------------------------------------ Sq = 17
__global__ void __launch_bounds__(64) k_attn_fwd17(Params P, uint16_t *__restrict__ o, float *__restrict__ lse) {
    mix_seed_state(P);
    __shared__ float Ks[PAIRS * PSTRIDE], Vs[PAIRS * PSTRIDE];
    const int lane = threadIdx.x, pl = lane / SK, i = lane - pl * SK;
    const int64_t pair = xcd_block() * PAIRS + pl;
    const bool active = pl < PAIRS && pair < P.B * P.H;
    const int64_t b = active ? pair / P.H : 0;
    const int h = active ? (int)(pair - b * P.H) : 0;
    float q[HD], t[HD];
    if (active) {
        load_row(P.k + b * P.k_sb + i * P.k_ss + h * HD, t);
        put_row(Ks + pl * PSTRIDE + i * ROW, t);
        load_row(P.v + b * P.v_sb + i * P.v_ss + h * HD, t);
        put_row(Vs + pl * PSTRIDE + i * ROW, t);
        load_row(P.q + b * P.q_sb + i * P.q_ss + h * HD, q);
    }
    __syncthreads();
    if (!active) return;
    // The loops over the 17 keys stay ROLLED (s[] lives in registers through indexed moves): fully unrolled, the
    // compiler hoists the LDS reads of every row, needs all 512 registers plus scratch and runs one wave per SIMD.
    float s[SK], m = -3.0e38f;
#pragma unroll 1
    for (int j = 0; j < SK; ++j) {
        s[j] = dot_lds(q, Ks + pl * PSTRIDE + j * ROW) * P.scale;
        m = fmaxf(m, s[j]);
    }
    float l = 0.f;
    for (int j = 0; j < SK; ++j) {
        s[j] = __expf(s[j] - m);
        l += s[j];
    }
    const float inv = 1.0f / l;
    const uint64_t base = ((uint64_t)pair * SK + i) * 32;
    float acc[HD];
    for (int d = 0; d < HD; ++d) acc[d] = 0.f;
#pragma unroll 1
    for (int j = 0; j < SK; ++j) {
        const float pj = (P.p_drop > 0.f && !keep_mask(P, base + j)) ? 0.f : s[j] * inv * P.inv_keep;
        axpy_lds(acc, pj, Vs + pl * PSTRIDE + j * ROW);
    }
    store_row(o + ((b * SK + i) * P.H + h) * HD, acc);
    lse[(b * P.H + h) * SK + i] = m + __logf(l);
}

__global__ void __launch_bounds__(64) k_attn_bwd17(Params P, const uint16_t *__restrict__ dout, const float *__restrict__ lse,
                                                   uint16_t *__restrict__ dq, uint16_t *__restrict__ dk,
                                                   uint16_t *__restrict__ dv) {
    mix_seed_state(P);
    // K/V tiles serve the first phase (dS, dQ), then the same LDS holds Q/dO for the second (dK, dV): 22 KB per
    // workgroup instead of 37, i.e. 7 resident waves per CU instead of 4
    __shared__ float Ks[PAIRS * PSTRIDE], Vs[PAIRS * PSTRIDE];
    __shared__ float dSs[PAIRS * SK * (SK + 1)], Pt[PAIRS * SK * (SK + 1)];
    float *const Qs = Ks, *const Gs = Vs;
    const int lane = threadIdx.x, pl = lane / SK, i = lane - pl * SK;
    const int64_t pair = xcd_block() * PAIRS + pl;
    const bool active = pl < PAIRS && pair < P.B * P.H;
    const int64_t b = active ? pair / P.H : 0;
    const int h = active ? (int)(pair - b * P.H) : 0;
    float q[HD], g[HD], t[HD];
    if (active) {
        load_row(P.k + b * P.k_sb + i * P.k_ss + h * HD, t);
        put_row(Ks + pl * PSTRIDE + i * ROW, t);
        load_row(P.v + b * P.v_sb + i * P.v_ss + h * HD, t);
        put_row(Vs + pl * PSTRIDE + i * ROW, t);
        load_row(P.q + b * P.q_sb + i * P.q_ss + h * HD, q);
        load_row(dout + ((b * SK + i) * P.H + h) * HD, g);
    }
    __syncthreads();
    if (active) {
        const float L = lse[(b * P.H + h) * SK + i];
        const uint64_t base = ((uint64_t)pair * SK + i) * 32;
        float p[SK], dp[SK], delta = 0.f;
#pragma unroll 1
        for (int j = 0; j < SK; ++j) {
            p[j] = __expf(dot_lds(q, Ks + pl * PSTRIDE + j * ROW) * P.scale - L);
            const bool keep = !(P.p_drop > 0.f) || keep_mask(P, base + j);
            dp[j] = keep ? dot_lds(g, Vs + pl * PSTRIDE + j * ROW) * P.inv_keep : 0.f;
            delta = fmaf(p[j], dp[j], delta);
            Pt[(pl * SK + i) * (SK + 1) + j] = keep ? p[j] * P.inv_keep : 0.f;
        }
        float acc[HD];
        for (int d = 0; d < HD; ++d) acc[d] = 0.f;
#pragma unroll 1
        for (int j = 0; j < SK; ++j) {
            const float ds = p[j] * (dp[j] - delta) * P.scale;
            dSs[(pl * SK + i) * (SK + 1) + j] = ds;
            axpy_lds(acc, ds, Ks + pl * PSTRIDE + j * ROW);
        }
        store_row(dq + b * P.q_sb + i * P.q_ss + h * HD, acc);
    }
    __syncthreads();  // every lane is done with K/V
    if (active) {
        put_row(Qs + pl * PSTRIDE + i * ROW, q);
        put_row(Gs + pl * PSTRIDE + i * ROW, g);
    }
    __syncthreads();
    if (!active) return;
    // this lane now owns key/value row j = i of its pair
    float ak[HD], av[HD];
    for (int d = 0; d < HD; ++d) ak[d] = av[d] = 0.f;
#pragma unroll 1
    for (int r = 0; r < SK; ++r) {
        axpy_lds(ak, dSs[(pl * SK + r) * (SK + 1) + i], Qs + pl * PSTRIDE + r * ROW);
        axpy_lds(av, Pt[(pl * SK + r) * (SK + 1) + i], Gs + pl * PSTRIDE + r * ROW);
    }
    store_row(dk + b * P.k_sb + i * P.k_ss + h * HD, ak);
    store_row(dv + b * P.v_sb + i * P.v_ss + h * HD, av);
}

// ------------------------------------------------------------------------------------------------ Sq = 1 (CLS row)
__global__ void __launch_bounds__(64) k_attn_fwd1(Params P, uint16_t *__restrict__ o, float *__restrict__ lse) {
    mix_seed_state(P);
    const int64_t pair = (int64_t)blockIdx.x * 64 + threadIdx.x;
    if (pair >= P.B * P.H) return;
    const int64_t b = pair / P.H;
    const int h = (int)(pair - b * P.H);
    float q[HD], t[HD], s[SK], m = -3.0e38f;
    load_row(P.q + b * P.q_sb + h * HD, q);
    for (int j = 0; j < SK; ++j) {
        load_row(P.k + b * P.k_sb + j * P.k_ss + h * HD, t);
        float a = 0.f;
        for (int d = 0; d < HD; ++d) a = fmaf(q[d], t[d], a);
        s[j] = a * P.scale;
        m = fmaxf(m, s[j]);
    }
    float l = 0.f;
    for (int j = 0; j < SK; ++j) {
        s[j] = __expf(s[j] - m);
        l += s[j];
    }
    const float inv = 1.0f / l;
    float acc[HD];
    for (int d = 0; d < HD; ++d) acc[d] = 0.f;
    for (int j = 0; j < SK; ++j) {
        const float pj = (P.p_drop > 0.f && !keep_mask(P, (uint64_t)pair * 32 + j)) ? 0.f : s[j] * inv * P.inv_keep;
        load_row(P.v + b * P.v_sb + j * P.v_ss + h * HD, t);
        for (int d = 0; d < HD; ++d) acc[d] = fmaf(pj, t[d], acc[d]);
    }
    store_row(o + (b * P.H + h) * HD, acc);
    lse[b * P.H + h] = m + __logf(l);
}

__global__ void __launch_bounds__(64) k_attn_bwd1(Params P, const uint16_t *__restrict__ dout, const float *__restrict__ lse,
                                                  uint16_t *__restrict__ dq, uint16_t *__restrict__ dk,
                                                  uint16_t *__restrict__ dv) {
    mix_seed_state(P);
    const int64_t pair = (int64_t)blockIdx.x * 64 + threadIdx.x;
    if (pair >= P.B * P.H) return;
    const int64_t b = pair / P.H;
    const int h = (int)(pair - b * P.H);
    float q[HD], g[HD], t[HD], p[SK], dp[SK], delta = 0.f;
    load_row(P.q + b * P.q_sb + h * HD, q);
    load_row(dout + (b * P.H + h) * HD, g);
    const float L = lse[b * P.H + h];
    for (int j = 0; j < SK; ++j) {
        load_row(P.k + b * P.k_sb + j * P.k_ss + h * HD, t);
        float a = 0.f;
        for (int d = 0; d < HD; ++d) a = fmaf(q[d], t[d], a);
        p[j] = __expf(a * P.scale - L);
        const bool keep = !(P.p_drop > 0.f) || keep_mask(P, (uint64_t)pair * 32 + j);
        load_row(P.v + b * P.v_sb + j * P.v_ss + h * HD, t);
        float c = 0.f;
        for (int d = 0; d < HD; ++d) c = fmaf(g[d], t[d], c);
        dp[j] = keep ? c * P.inv_keep : 0.f;
        delta = fmaf(p[j], dp[j], delta);
        // dV_j = Ptilde_j * dO
        const float pt = keep ? p[j] * P.inv_keep : 0.f;
        for (int d = 0; d < HD; ++d) t[d] = pt * g[d];
        store_row(dv + b * P.v_sb + j * P.v_ss + h * HD, t);
    }
    float acc[HD];
    for (int d = 0; d < HD; ++d) acc[d] = 0.f;
    for (int j = 0; j < SK; ++j) {
        const float ds = p[j] * (dp[j] - delta) * P.scale;
        load_row(P.k + b * P.k_sb + j * P.k_ss + h * HD, t);
        for (int d = 0; d < HD; ++d) acc[d] = fmaf(ds, t[d], acc[d]);
        for (int d = 0; d < HD; ++d) t[d] = ds * q[d];
        store_row(dk + b * P.k_sb + j * P.k_ss + h * HD, t);
    }
    store_row(dq + b * P.q_sb + h * HD, acc);
}

inline bool fill(Params &P, const void *q, const void *k, const void *v, int64_t B, int H, int Sq, int64_t q_sb,
                 int64_t q_ss, int64_t k_sb, int64_t k_ss, int64_t v_sb, int64_t v_ss, float scale, float p_drop,
                 uint64_t seed, const uint64_t *seed_state) {
    if (!q || !k || !v || B <= 0 || H <= 0 || (Sq != 1 && Sq != SK) || !(p_drop >= 0.f && p_drop < 1.f)) return false;
    // rows are read/written as 4 x 16 bytes
    if (((uintptr_t)q | (uintptr_t)k | (uintptr_t)v) & 15) return false;
    if ((q_sb | q_ss | k_sb | k_ss | v_sb | v_ss) & 7) return false;
    P.q = (const uint16_t *)q; P.k = (const uint16_t *)k; P.v = (const uint16_t *)v;
    P.B = B; P.H = H;
    P.q_sb = q_sb; P.q_ss = q_ss; P.k_sb = k_sb; P.k_ss = k_ss; P.v_sb = v_sb; P.v_ss = v_ss;
    P.scale = scale; P.p_drop = p_drop; P.inv_keep = 1.0f / (1.0f - p_drop);
    P.seed0 = (uint32_t)seed; P.seed1 = (uint32_t)(seed >> 32); P.seed_state = seed_state;
    P.thr = (uint32_t)(p_drop * 16777216.0f);
    return true;
}
inline int done() {
    const hipError_t e = hipGetLastError();
    return e == hipSuccess ? 0 : -(1000 + (int)e);
}

}  // namespace

extern "C" int g2048_attn_fwd(const void *q, const void *k, const void *v, void *o, float *lse, int64_t B, int H, int Sq,
                              int64_t q_sb, int64_t q_ss, int64_t k_sb, int64_t k_ss, int64_t v_sb, int64_t v_ss,
                              float scale, float p_drop, uint64_t seed, const uint64_t *seed_state, void *stream) {
    Params P;
    if (!o || !lse || ((uintptr_t)o & 15) || !fill(P, q, k, v, B, H, Sq, q_sb, q_ss, k_sb, k_ss, v_sb, v_ss, scale, p_drop, seed, seed_state))
        return G2048_EINVAL;
    const int64_t pairs = B * H;
    if (Sq == SK)
        hipLaunchKernelGGL(k_attn_fwd17, dim3((unsigned)((pairs + PAIRS - 1) / PAIRS)), dim3(64), 0, (hipStream_t)stream, P,
                           (uint16_t *)o, lse);
    else
        hipLaunchKernelGGL(k_attn_fwd1, dim3((unsigned)((pairs + 63) / 64)), dim3(64), 0, (hipStream_t)stream, P,
                           (uint16_t *)o, lse);
    return done();
}

extern "C" int g2048_attn_bwd(const void *q, const void *k, const void *v, const void *dout, const float *lse, void *dq,
                              void *dk, void *dv, int64_t B, int H, int Sq, int64_t q_sb, int64_t q_ss, int64_t k_sb,
                              int64_t k_ss, int64_t v_sb, int64_t v_ss, float scale, float p_drop, uint64_t seed,
                              const uint64_t *seed_state, void *stream) {
    Params P;
    if (!dout || !lse || !dq || !dk || !dv || (((uintptr_t)dout | (uintptr_t)dq | (uintptr_t)dk | (uintptr_t)dv) & 15) ||
        !fill(P, q, k, v, B, H, Sq, q_sb, q_ss, k_sb, k_ss, v_sb, v_ss, scale, p_drop, seed, seed_state))
        return G2048_EINVAL;
    const int64_t pairs = B * H;
    if (Sq == SK)
        hipLaunchKernelGGL(k_attn_bwd17, dim3((unsigned)((pairs + PAIRS - 1) / PAIRS)), dim3(64), 0, (hipStream_t)stream, P,
                           (const uint16_t *)dout, lse, (uint16_t *)dq, (uint16_t *)dk, (uint16_t *)dv);
    else
        hipLaunchKernelGGL(k_attn_bwd1, dim3((unsigned)((pairs + 63) / 64)), dim3(64), 0, (hipStream_t)stream, P,
                           (const uint16_t *)dout, lse, (uint16_t *)dq, (uint16_t *)dk, (uint16_t *)dv);
    return done();
}
