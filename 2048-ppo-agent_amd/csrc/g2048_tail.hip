// The 2048-row tail of the PPO update as two kernels (+ one grouped weight-gradient kernel) on gfx950.
//
// With the "cls" reduction only the CLS row of the LAST encoder layer is needed, so everything after that layer's attention
// works on one row per board: out_proj + residual + LayerNorm, the feed-forward block, the residual, and both heads
// (reference: nn.TransformerEncoderLayer(norm_first=True) built at src/ppo/transformer_encoder.py:138-148, read out at
// :150-190; actor / critic at src/ppo/ppo_agent.py:62-92).  In round 2 that was ~20 launches forward and ~35 backward of
// 4-10 us each, every one of them at the launch-latency floor of dependent kernels inside a hipGraph: 0.45 ms of a 2.3 ms
// minibatch for 1 % of its FLOPs.  Here:
//   k_tail_fwd  o (attention output of the CLS rows), x (residual CLS rows)  ->  logits, values (+ what the backward needs)
//   k_tail_bwd  d logits, d values  ->  d o, d x, LayerNorm gradient partials, and every Linear's dY^T
//   k_dweight_t all weight and bias gradients of the tail from the transposed operands the two kernels left behind
//
// Decomposition: one workgroup (4 waves) owns 32 boards and walks the whole chain; between Linears the activations of
// those 32 rows live in LDS (row-major bf16, padded rows), the four waves split every Linear's OUTPUT features.  Every GEMM
// is computed transposed, Y^T[out][row] = W[out][in] . X^T, with v_mfma_f32_32x32x16_bf16: the A operand is a weight tile
// in nn.Linear's own [out][in] layout, read straight from global memory (L2-resident: 2.7 MB of bf16 weights shared by
// all workgroups; each wave streams only the rows of its own output tiles, so there is no LDS staging of weights and no
// barrier inside a Linear), the B operand comes from the LDS activation tile, rows sit on lanes.  The backward needs
// W^T as the A operand: the optimiser kernel maintains transposed bf16 shadows of these weights (g2048_opt_step).
// Time is set by streaming the weights through each CU's vector-memory path once (2.7 MB at ~64 B/clk), not by the MFMAs.
//
// Numerics mirror torch.autocast(bf16) as the unfused path did: bf16 GEMM inputs, f32 accumulation, Linear outputs rounded
// to bf16 before dropout / residual add, f32 residual and LayerNorm statistics, bf16 gradients between Linears.
// Dropout masks are functions of (seed, *seed_state, site, element index) like every other kernel of the update; ReLU (and
// ReLU o dropout) patterns travel to the backward as one bit per element in the accumulator layout.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/g2048.h"

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int D = 256, FF = 1024, HID = 512, TB = 32, THREADS = 256, FC = 128;
constexpr int S256 = 2 * 256 + 16, S512 = 2 * 512 + 16, S128 = 2 * 128 + 16;  // LDS row strides in bytes (+16: bank spread)
constexpr int XM_S = S512 / 4;                                                  // the f32 [32][256] tile uses the same rows
constexpr int MT_FFN = 0, MT_A1 = 32, MT_A2 = 48, MT_C1 = 64, MT_C2 = 80, N_MASK_TILES = G2048_TAIL_MASK_TILES;
static_assert(N_MASK_TILES == 96, "mask tile table");

// accumulator register i of lane (r, h) holds row rowof(i, h), column r of a 32 x 32 tile
__device__ __forceinline__ int rowof(int i, int h) { return (i & 3) + 8 * (i >> 2) + 4 * h; }
__device__ __forceinline__ f32x16 mfma(bf16x8 a, bf16x8 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
}
__device__ __forceinline__ float wave_sum(float v) {
    for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m);
    return v;
}
// the update's dropout hash (same function as csrc/g2048_layernorm.hip)
__device__ __forceinline__ bool keep_elem(uint32_t s0, uint32_t s1, uint32_t thr, uint64_t idx) {
    uint32_t x = (uint32_t)idx * 0x9E3779B1u ^ s0;
    x ^= (uint32_t)(idx >> 32) * 0x85EBCA77u + s1;
    x ^= x >> 16; x *= 0x7FEB352Du; x ^= x >> 15; x *= 0x846CA68Bu; x ^= x >> 16;
    return (x >> 8) >= thr;
}
struct Drop {
    uint32_t s0, s1, thr;
    float inv_keep;
    __device__ __forceinline__ Drop site(uint32_t k) const { return Drop{s0 + k * 0x632BE5ABu, s1 ^ (k * 0x7F4A7C15u), thr, inv_keep}; }
    __device__ __forceinline__ float apply(float v, uint64_t idx) const {
        return thr ? (keep_elem(s0, s1, thr, idx) ? v * inv_keep : 0.0f) : v;
    }
};
__device__ __forceinline__ Drop make_drop(uint64_t seed, const uint64_t *seed_state, float p_drop) {
    uint32_t s0 = (uint32_t)seed, s1 = (uint32_t)(seed >> 32);
    if (seed_state) {
        const uint64_t s = *seed_state;
        s0 ^= (uint32_t)s * 0x9E3779B1u;
        s1 += (uint32_t)(s >> 32) * 0x85EBCA77u + (uint32_t)s;
    }
    return Drop{s0, s1, (uint32_t)(p_drop * 16777216.0f), 1.0f / (1.0f - p_drop)};
}

// acc[i] = b[row0 + rowof(i, h)]: the bias enters through the accumulator's initial value
__device__ __forceinline__ f32x16 bias_tile(const float *b, int row0, int h) {
    f32x16 a;
    for (int g = 0; g < 4; ++g) {
        const f32x4 v = *reinterpret_cast<const f32x4 *>(b + row0 + 8 * g + 4 * h);
        for (int q = 0; q < 4; ++q) a[4 * g + q] = v[q];
    }
    return a;
}
__device__ __forceinline__ f32x16 zero_tile() {
    f32x16 a;
    for (int i = 0; i < 16; ++i) a[i] = 0.f;
    return a;
}
// acc += W[row0 .. row0+31][k0 .. k0 + 16 NK) . X, X = NK operand fragments (rows on lanes).  W row-major with leading
// dimension ld (elements); the A fragment of k-step ks is 16 bytes of row row0 + r at column k0 + 16 ks + 8 h.
template <int NK>
__device__ __forceinline__ f32x16 tile_gemm(const __bf16 *__restrict__ W, int ld, int row0, int k0, const bf16x8 *xf, f32x16 acc, int r,
                                            int h) {
    const __bf16 *p = W + (size_t)(row0 + r) * ld + k0 + 8 * h;
    bf16x8 a[NK];
#pragma unroll
    for (int ks = 0; ks < NK; ++ks) a[ks] = *reinterpret_cast<const bf16x8 *>(p + 16 * ks);
#pragma unroll
    for (int ks = 0; ks < NK; ++ks) acc = mfma(a[ks], xf[ks], acc);
    return acc;
}
// operand fragments of an LDS activation tile (row-major bf16, byte stride `stride`): columns k0 .. k0 + 16 NK of row r
template <int NK>
__device__ __forceinline__ void load_frags(const char *buf, int stride, int k0, bf16x8 *xf, int r, int h) {
#pragma unroll
    for (int ks = 0; ks < NK; ++ks) xf[ks] = *reinterpret_cast<const bf16x8 *>(buf + r * stride + 2 * (k0 + 16 * ks + 8 * h));
}
// four consecutive features (accumulator group g) of row r into a row-major LDS tile
__device__ __forceinline__ void put4(char *buf, int stride, int r, int col, const float v[4]) {
    bf16x4 pk;
    for (int q = 0; q < 4; ++q) pk[q] = (__bf16)v[q];
    *reinterpret_cast<bf16x4 *>(buf + r * stride + 2 * col) = pk;
}
// dstT[(f0 + f) * ld + m0 + row] = buf[row][f] for f < nfeat, row < 32: the transposed copy the weight-gradient kernel reads
// (16-byte stores of 8 rows each).  Call between two barriers; nfeat a multiple of 64.
__device__ __forceinline__ void lds_to_T(const char *buf, int stride, int nfeat, __bf16 *__restrict__ dstT, int64_t ld, int64_t m0,
                                         int tid) {
    for (int e = tid; e < nfeat * 4; e += THREADS) {
        const int f = e % nfeat, g = e / nfeat;
        bf16x8 v;
        for (int j = 0; j < 8; ++j) v[j] = *reinterpret_cast<const __bf16 *>(buf + (8 * g + j) * stride + 2 * f);
        *reinterpret_cast<bf16x8 *>(dstT + (int64_t)f * ld + m0 + 8 * g) = v;
    }
}

struct TailLds {
    char xa[TB * S512];     // bf16 activations, rows of up to 512
    char xb[TB * S512];
    char xc[TB * S512];     // first the f32 [32][256] residual tile (same row stride), later bf16 activations
    char u[2][TB * S128];   // feed-forward hidden chunk (128 units), double-buffered
    float red[4][2][D];     // backward: LayerNorm gradient partials of the four waves
    float dl[TB][8];        // backward: d logits (4) and d value of the tile's rows
};
static_assert(sizeof(TailLds) <= 160 * 1024, "LDS budget");

// ---------------------------------------------------------------------------------------------------------------------
// forward
// ---------------------------------------------------------------------------------------------------------------------
// Linear(256 -> 512) + ReLU, Linear(512 -> 512) + ReLU, Linear(512 -> n_out, no bias) on the bf16 features in L.xa
template <int N_OUT>
__device__ __forceinline__ void head_fwd(TailLds &L, const __bf16 *w1, const float *b1, const __bf16 *w2, const float *b2,
                                         const __bf16 *w3, int mt1, int mt2, uint16_t *masks_wg, __bf16 *h1T, __bf16 *h2T, int64_t ld,
                                         int64_t m0, bool valid, float *out, int64_t M, int tid, int lane, int r, int h, int w) {
    {
        bf16x8 ff[16];
        load_frags<16>(L.xa, S256, 0, ff, r, h);
        for (int t = 0; t < 4; ++t) {
            const int mt = 4 * w + t;
            f32x16 acc = tile_gemm<16>(w1, D, 32 * mt, 0, ff, bias_tile(b1, 32 * mt, h), r, h);
            uint32_t bits = 0;
            for (int g = 0; g < 4; ++g) {
                float v[4];
                for (int q = 0; q < 4; ++q) {
                    v[q] = valid ? fmaxf(acc[4 * g + q], 0.f) : 0.f;
                    bits |= (uint32_t)((float)(__bf16)v[q] != 0.f) << (4 * g + q);
                }
                put4(L.xb, S512, r, 32 * mt + 8 * g + 4 * h, v);
            }
            masks_wg[(mt1 + mt) * 64 + lane] = (uint16_t)bits;
        }
    }
    __syncthreads();
    lds_to_T(L.xb, S512, HID, h1T, ld, m0, tid);
    {
        bf16x8 af[32];
        load_frags<32>(L.xb, S512, 0, af, r, h);
        for (int t = 0; t < 4; ++t) {
            const int mt = 4 * w + t;
            f32x16 acc = tile_gemm<16>(w2, HID, 32 * mt, 0, af, bias_tile(b2, 32 * mt, h), r, h);
            acc = tile_gemm<16>(w2, HID, 32 * mt, 256, af + 16, acc, r, h);
            uint32_t bits = 0;
            for (int g = 0; g < 4; ++g) {
                float v[4];
                for (int q = 0; q < 4; ++q) {
                    v[q] = valid ? fmaxf(acc[4 * g + q], 0.f) : 0.f;
                    bits |= (uint32_t)((float)(__bf16)v[q] != 0.f) << (4 * g + q);
                }
                put4(L.xc, S512, r, 32 * mt + 8 * g + 4 * h, v);
            }
            masks_wg[(mt2 + mt) * 64 + lane] = (uint16_t)bits;
        }
    }
    __syncthreads();
    lds_to_T(L.xc, S512, HID, h2T, ld, m0, tid);
    {   // the output layer (4 logits / 1 value) on the vector ALU: 8 threads per row, 64 inputs each
        const int row = tid >> 3, part = tid & 7;
        float s[N_OUT];
        for (int o = 0; o < N_OUT; ++o) s[o] = 0.f;
        for (int c = 0; c < 8; ++c) {
            const bf16x8 x = *reinterpret_cast<const bf16x8 *>(L.xc + row * S512 + 2 * (64 * part + 8 * c));
            for (int o = 0; o < N_OUT; ++o) {
                const bf16x8 wv = *reinterpret_cast<const bf16x8 *>(w3 + o * HID + 64 * part + 8 * c);
                for (int j = 0; j < 8; ++j) s[o] = __builtin_fmaf((float)x[j], (float)wv[j], s[o]);
            }
        }
        for (int o = 0; o < N_OUT; ++o) {
            s[o] += __shfl_xor(s[o], 1);
            s[o] += __shfl_xor(s[o], 2);
            s[o] += __shfl_xor(s[o], 4);
        }
        if (part == 0 && m0 + row < M)
            for (int o = 0; o < N_OUT; ++o) out[(m0 + row) * N_OUT + o] = s[o];
    }
    __syncthreads();  // xb / xc are free again
}

__global__ void __launch_bounds__(THREADS, 1)
k_tail_fwd(const __bf16 *__restrict__ o, const float *__restrict__ x_cls, int64_t x_rs, g2048_tail_weights W, g2048_tail_saved S,
           float *__restrict__ logits, float *__restrict__ values, int64_t M, float eps, float p_drop, uint64_t seed,
           const uint64_t *seed_state) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    TailLds &L = *reinterpret_cast<TailLds *>(smem);
    const int tid = threadIdx.x, lane = tid & 63, r = lane & 31, h = lane >> 5;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int64_t m0 = (int64_t)blockIdx.x * TB, ld = S.ld;
    const bool valid = m0 + r < M;  // this lane's row in the accumulator layout
    const Drop drop = make_drop(seed, seed_state, p_drop);
    float *const xm = reinterpret_cast<float *>(L.xc);
    uint16_t *const masks_wg = reinterpret_cast<uint16_t *>(S.masks) + (int64_t)blockIdx.x * N_MASK_TILES * 64;
    const __bf16 *wo = (const __bf16 *)W.wo, *w1 = (const __bf16 *)W.w1, *w2 = (const __bf16 *)W.w2;

    // ---- the tile's inputs: attention output rows -> xa (bf16), residual CLS rows -> xm (f32); rows past M are zero
    for (int p = 0; p < 4; ++p) {
        const int row = 8 * p + (tid >> 5), ch = tid & 31;
        uint4 v = make_uint4(0u, 0u, 0u, 0u);
        if (m0 + row < M) v = *reinterpret_cast<const uint4 *>(o + (m0 + row) * D + 8 * ch);
        *reinterpret_cast<uint4 *>(L.xa + row * S256 + 16 * ch) = v;
    }
    for (int p = 0; p < 8; ++p) {
        const int row = 4 * p + (tid >> 6), c4 = tid & 63;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (m0 + row < M) v = *reinterpret_cast<const float4 *>(x_cls + (m0 + row) * x_rs + 4 * c4);
        *reinterpret_cast<float4 *>(xm + row * XM_S + 4 * c4) = v;
    }
    __syncthreads();
    lds_to_T(L.xa, S256, D, (__bf16 *)S.oT, ld, m0, tid);

    // ---- out_proj, dropout, residual add: x_mid = x + dropout(bf16(Wo o + bo))
    {
        bf16x8 xf[16];
        load_frags<16>(L.xa, S256, 0, xf, r, h);
        const Drop d1 = drop.site(1);
        for (int j = 0; j < 2; ++j) {
            const int mt = 2 * w + j;
            const f32x16 acc = tile_gemm<16>(wo, D, 32 * mt, 0, xf, bias_tile(W.bo, 32 * mt, h), r, h);
            for (int g = 0; g < 4; ++g) {
                const int f0 = 32 * mt + 8 * g + 4 * h;
                f32x4 x = *reinterpret_cast<const f32x4 *>(xm + r * XM_S + f0);
                for (int q = 0; q < 4; ++q)
                    x[q] += d1.apply((float)(__bf16)acc[4 * g + q], (uint64_t)(m0 + r) * D + f0 + q);
                *reinterpret_cast<f32x4 *>(xm + r * XM_S + f0) = x;
            }
        }
    }
    __syncthreads();
    // ---- LayerNorm of the 32 rows, one wavefront per row (the arithmetic of k_add_ln_fwd): h2 -> xb, x_mid + statistics saved
    {
        const float4 gm = reinterpret_cast<const float4 *>(W.ln_g)[lane], bt = reinterpret_cast<const float4 *>(W.ln_b)[lane];
        for (int i = 0; i < 8; ++i) {
            const int row = 8 * w + i;
            const float4 v = *reinterpret_cast<const float4 *>(xm + row * XM_S + 4 * lane);
            const float mean = wave_sum(v.x + v.y + v.z + v.w) * (1.0f / D);
            const float dx = v.x - mean, dy = v.y - mean, dz = v.z - mean, dw = v.w - mean;
            const float rstd = rsqrtf(wave_sum(dx * dx + dy * dy + dz * dz + dw * dw) * (1.0f / D) + eps);
            const bool ok = m0 + row < M;
            float hv[4] = {dx * rstd * gm.x + bt.x, dy * rstd * gm.y + bt.y, dz * rstd * gm.z + bt.z, dw * rstd * gm.w + bt.w};
            if (!ok) hv[0] = hv[1] = hv[2] = hv[3] = 0.f;
            put4(L.xb, S256, row, 4 * lane, hv);
            if (ok) {
                reinterpret_cast<float4 *>(S.x_mid + (m0 + row) * D)[lane] = v;
                if (lane == 0) {
                    S.mean[m0 + row] = mean;
                    S.rstd[m0 + row] = rstd;
                }
            }
        }
    }
    __syncthreads();
    lds_to_T(L.xb, S256, D, (__bf16 *)S.h2T, ld, m0, tid);

    // ---- feed-forward: u = dropout(relu(W1 h2 + b1)) in chunks of 128 hidden units, f = W2 u + b2 accumulated per chunk
    {
        bf16x8 xf[16];
        load_frags<16>(L.xb, S256, 0, xf, r, h);
        const Drop d2 = drop.site(2), d3 = drop.site(3);
        f32x16 acc2[2];
        for (int j = 0; j < 2; ++j) acc2[j] = bias_tile(W.b2, 32 * (2 * w + j), h);
#pragma nounroll
        for (int c = 0; c < FF / FC; ++c) {
            const int ht = 4 * c + w;  // hidden tile of this wave
            char *ub = L.u[c & 1];
            const f32x16 z = tile_gemm<16>(w1, D, 32 * ht, 0, xf, bias_tile(W.b1, 32 * ht, h), r, h);
            uint32_t bits = 0;
            for (int g = 0; g < 4; ++g) {
                float v[4];
                for (int q = 0; q < 4; ++q) {
                    const int hid = 32 * ht + 8 * g + 4 * h + q;
                    v[q] = valid ? d2.apply(fmaxf(z[4 * g + q], 0.f), (uint64_t)(m0 + r) * FF + hid) : 0.f;
                    bits |= (uint32_t)((float)(__bf16)v[q] != 0.f) << (4 * g + q);
                }
                put4(ub, S128, r, 32 * w + 8 * g + 4 * h, v);
            }
            masks_wg[(MT_FFN + ht) * 64 + lane] = (uint16_t)bits;
            __syncthreads();  // chunk c of every wave visible (the other buffer is still being read by nobody: see below)
            lds_to_T(ub, S128, FC, (__bf16 *)S.uT + (int64_t)(FC * c) * ld, ld, m0, tid);
            bf16x8 uf[8];
            load_frags<8>(ub, S128, 0, uf, r, h);
            for (int j = 0; j < 2; ++j) acc2[j] = tile_gemm<8>(w2, FF, 32 * (2 * w + j), FC * c, uf, acc2[j], r, h);
            // a wave reaches the writes of chunk c + 2 (same buffer) only after the barrier of chunk c + 1, which every wave
            // passes after these reads
        }
        // ---- features = bf16(x_mid + dropout(bf16(f))) -> xa
        for (int j = 0; j < 2; ++j) {
            const int mt = 2 * w + j;
            for (int g = 0; g < 4; ++g) {
                const int f0 = 32 * mt + 8 * g + 4 * h;
                const f32x4 x = *reinterpret_cast<const f32x4 *>(xm + r * XM_S + f0);
                float v[4];
                for (int q = 0; q < 4; ++q)
                    v[q] = valid ? x[q] + d3.apply((float)(__bf16)acc2[j][4 * g + q], (uint64_t)(m0 + r) * D + f0 + q) : 0.f;
                put4(L.xa, S256, r, f0, v);
            }
        }
    }
    __syncthreads();
    lds_to_T(L.xa, S256, D, (__bf16 *)S.featsT, ld, m0, tid);
    __syncthreads();  // xc (the f32 residual tile) is dead from here on: the heads write bf16 rows into it

    head_fwd<4>(L, (const __bf16 *)W.a1, W.ab1, (const __bf16 *)W.a2, W.ab2, (const __bf16 *)W.a3, MT_A1, MT_A2, masks_wg,
                (__bf16 *)S.a1T, (__bf16 *)S.a2T, ld, m0, valid, logits, M, tid, lane, r, h, w);
    head_fwd<1>(L, (const __bf16 *)W.c1, W.cb1, (const __bf16 *)W.c2, W.cb2, (const __bf16 *)W.c3, MT_C1, MT_C2, masks_wg,
                (__bf16 *)S.c1T, (__bf16 *)S.c2T, ld, m0, valid, values, M, tid, lane, r, h, w);
}

// ---------------------------------------------------------------------------------------------------------------------
// backward
// ---------------------------------------------------------------------------------------------------------------------
// d(features) contribution of one head.  dl_col0 .. dl_col0 + N_OUT - 1: this head's columns of L.dl.
// WT1 = W1^T [256][512], WT2 = W2^T [512][512] (transposed shadows), w3 [N_OUT][512] as it is.
template <int N_OUT>
__device__ __forceinline__ void head_bwd(TailLds &L, const __bf16 *wT1, const __bf16 *wT2, const __bf16 *w3, int dl_col0, int mt1, int mt2,
                                         const uint16_t *masks_wg, __bf16 *d3T, __bf16 *d2T, __bf16 *d1T, int64_t ld, int64_t m0,
                                         f32x16 dfeat[2], int tid, int lane, int r, int h, int w) {
    // d a2 = (W3^T d out) where a2 > 0 -> xa; d out^T (bf16, rows N_OUT.. of the 32-row buffer stay zero) for the weight gradient
    if (tid < 32 * N_OUT) {
        const int o = tid / 32, row = tid % 32;
        d3T[(int64_t)o * ld + m0 + row] = (__bf16)L.dl[row][dl_col0 + o];
    }
    for (int t = 0; t < 4; ++t) {
        const int mt = 4 * w + t;
        const uint32_t bits = masks_wg[(mt2 + mt) * 64 + lane];
        float dlr[N_OUT];
        for (int o = 0; o < N_OUT; ++o) dlr[o] = (float)(__bf16)L.dl[r][dl_col0 + o];
        for (int g = 0; g < 4; ++g) {
            const int f0 = 32 * mt + 8 * g + 4 * h;
            float v[4] = {0.f, 0.f, 0.f, 0.f};
            for (int o = 0; o < N_OUT; ++o) {
                const bf16x4 wv = *reinterpret_cast<const bf16x4 *>(w3 + o * HID + f0);
                for (int q = 0; q < 4; ++q) v[q] = __builtin_fmaf((float)wv[q], dlr[o], v[q]);
            }
            for (int q = 0; q < 4; ++q) v[q] = ((bits >> (4 * g + q)) & 1u) ? v[q] : 0.f;
            put4(L.xa, S512, r, f0, v);
        }
    }
    __syncthreads();
    lds_to_T(L.xa, S512, HID, d2T, ld, m0, tid);
    {   // d a1 = (W2^T d a2) where a1 > 0 -> xb
        bf16x8 af[32];
        load_frags<32>(L.xa, S512, 0, af, r, h);
        for (int t = 0; t < 4; ++t) {
            const int mt = 4 * w + t;
            f32x16 acc = tile_gemm<16>(wT2, HID, 32 * mt, 0, af, zero_tile(), r, h);
            acc = tile_gemm<16>(wT2, HID, 32 * mt, 256, af + 16, acc, r, h);
            const uint32_t bits = masks_wg[(mt1 + mt) * 64 + lane];
            for (int g = 0; g < 4; ++g) {
                float v[4];
                for (int q = 0; q < 4; ++q) v[q] = ((bits >> (4 * g + q)) & 1u) ? acc[4 * g + q] : 0.f;
                put4(L.xb, S512, r, 32 * mt + 8 * g + 4 * h, v);
            }
        }
    }
    __syncthreads();
    lds_to_T(L.xb, S512, HID, d1T, ld, m0, tid);
    {   // d features += W1^T d a1
        bf16x8 af[32];
        load_frags<32>(L.xb, S512, 0, af, r, h);
        for (int j = 0; j < 2; ++j) {
            const int mt = 2 * w + j;
            dfeat[j] = tile_gemm<16>(wT1, HID, 32 * mt, 0, af, dfeat[j], r, h);
            dfeat[j] = tile_gemm<16>(wT1, HID, 32 * mt, 256, af + 16, dfeat[j], r, h);
        }
    }
    __syncthreads();  // xa / xb are free again (every wave holds its fragments in registers)
}

__global__ void __launch_bounds__(THREADS, 1)
k_tail_bwd(const float *__restrict__ dlogits, const float *__restrict__ dvalues, g2048_tail_weights_t WT, g2048_tail_saved S,
           g2048_tail_grads G, __bf16 *__restrict__ d_o, float *__restrict__ dx_cls, int64_t M, float p_drop, uint64_t seed,
           const uint64_t *seed_state) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    TailLds &L = *reinterpret_cast<TailLds *>(smem);
    const int tid = threadIdx.x, lane = tid & 63, r = lane & 31, h = lane >> 5;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int64_t m0 = (int64_t)blockIdx.x * TB, ld = S.ld;
    const Drop drop = make_drop(seed, seed_state, p_drop);
    float *const xm = reinterpret_cast<float *>(L.xc);
    const uint16_t *const masks_wg = reinterpret_cast<const uint16_t *>(S.masks) + (int64_t)blockIdx.x * N_MASK_TILES * 64;

    if (tid < 32) {
        const bool ok = m0 + tid < M;
        const float4 v = ok ? reinterpret_cast<const float4 *>(dlogits)[m0 + tid] : make_float4(0.f, 0.f, 0.f, 0.f);
        L.dl[tid][0] = v.x; L.dl[tid][1] = v.y; L.dl[tid][2] = v.z; L.dl[tid][3] = v.w;
        L.dl[tid][4] = ok ? dvalues[m0 + tid] : 0.f;
    }
    __syncthreads();

    // ---- both heads -> d features (f32, this wave's two tiles)
    f32x16 dfeat[2] = {zero_tile(), zero_tile()};
    head_bwd<4>(L, (const __bf16 *)WT.a1T, (const __bf16 *)WT.a2T, (const __bf16 *)WT.a3, 0, MT_A1, MT_A2, masks_wg, (__bf16 *)G.dlT,
                (__bf16 *)G.da2T, (__bf16 *)G.da1T, ld, m0, dfeat, tid, lane, r, h, w);
    head_bwd<1>(L, (const __bf16 *)WT.c1T, (const __bf16 *)WT.c2T, (const __bf16 *)WT.c3, 4, MT_C1, MT_C2, masks_wg, (__bf16 *)G.dvT,
                (__bf16 *)G.dc2T, (__bf16 *)G.dc1T, ld, m0, dfeat, tid, lane, r, h, w);

    // ---- features = bf16(x_mid + dropout(f)): g = bf16(d features) flows into the residual (-> xm, f32) and, masked, into f (-> xa)
    {
        const Drop d3 = drop.site(3);
        for (int j = 0; j < 2; ++j) {
            const int mt = 2 * w + j;
            for (int g = 0; g < 4; ++g) {
                const int f0 = 32 * mt + 8 * g + 4 * h;
                f32x4 gx;
                float v[4];
                for (int q = 0; q < 4; ++q) {
                    gx[q] = (float)(__bf16)dfeat[j][4 * g + q];
                    v[q] = d3.apply(gx[q], (uint64_t)(m0 + r) * D + f0 + q);
                }
                *reinterpret_cast<f32x4 *>(xm + r * XM_S + f0) = gx;
                put4(L.xa, S256, r, f0, v);
            }
        }
    }
    __syncthreads();
    lds_to_T(L.xa, S256, D, (__bf16 *)G.df2T, ld, m0, tid);

    // ---- feed-forward backward, in the forward's chunks: dz = (W2^T df) * [u != 0] / keep;  d h2 += W1^T dz
    {
        bf16x8 xf[16];
        load_frags<16>(L.xa, S256, 0, xf, r, h);
        const __bf16 *w2T = (const __bf16 *)WT.w2T, *w1T = (const __bf16 *)WT.w1T;
        f32x16 dh[2] = {zero_tile(), zero_tile()};
#pragma nounroll
        for (int c = 0; c < FF / FC; ++c) {
            const int ht = 4 * c + w;
            char *ub = L.u[c & 1];
            const f32x16 du = tile_gemm<16>(w2T, D, 32 * ht, 0, xf, zero_tile(), r, h);
            const uint32_t bits = masks_wg[(MT_FFN + ht) * 64 + lane];
            for (int g = 0; g < 4; ++g) {
                float v[4];
                for (int q = 0; q < 4; ++q) v[q] = ((bits >> (4 * g + q)) & 1u) ? du[4 * g + q] * drop.inv_keep : 0.f;
                put4(ub, S128, r, 32 * w + 8 * g + 4 * h, v);
            }
            __syncthreads();
            lds_to_T(ub, S128, FC, (__bf16 *)G.dzT + (int64_t)(FC * c) * ld, ld, m0, tid);
            bf16x8 uf[8];
            load_frags<8>(ub, S128, 0, uf, r, h);
            for (int j = 0; j < 2; ++j) dh[j] = tile_gemm<8>(w1T, FF, 32 * (2 * w + j), FC * c, uf, dh[j], r, h);
        }
        // d h2 (bf16, as the unfused path hands it to the LayerNorm backward) -> xb
        for (int j = 0; j < 2; ++j)
            for (int g = 0; g < 4; ++g) {
                float v[4];
                for (int q = 0; q < 4; ++q) v[q] = dh[j][4 * g + q];
                put4(L.xb, S256, r, 32 * (2 * w + j) + 8 * g + 4 * h, v);
            }
    }
    __syncthreads();

    // ---- LayerNorm backward + residual, one wavefront per row (the arithmetic of k_add_ln_bwd):
    //      dx = g + dLN(d h2);  d(out_proj output) = dropout-masked dx (bf16) -> xa;  gamma / beta partials of this workgroup
    {
        const Drop d1 = drop.site(1);
        const float4 gm = reinterpret_cast<const float4 *>(WT.ln_g)[lane];
        const float gg[4] = {gm.x, gm.y, gm.z, gm.w};
        float dg[4] = {0, 0, 0, 0}, db[4] = {0, 0, 0, 0};
        for (int i = 0; i < 8; ++i) {
            const int row = 8 * w + i;
            const bool ok = m0 + row < M;
            const int64_t m = ok ? m0 + row : M - 1;  // clamped: loads stay in bounds; rows past M carry zero gradients
            const float4 v = reinterpret_cast<const float4 *>(S.x_mid + m * D)[lane];
            const float mean = S.mean[m], rstd = S.rstd[m];
            const bf16x4 ghb = *reinterpret_cast<const bf16x4 *>(L.xb + row * S256 + 8 * lane);
            const float4 gx = *reinterpret_cast<const float4 *>(xm + row * XM_S + 4 * lane);
            const float xh[4] = {(v.x - mean) * rstd, (v.y - mean) * rstd, (v.z - mean) * rstd, (v.w - mean) * rstd};
            float dxh[4], s1 = 0.f, s2 = 0.f;
            for (int q = 0; q < 4; ++q) {
                const float gh = ok ? (float)ghb[q] : 0.f;
                dxh[q] = gh * gg[q];
                s1 += dxh[q];
                s2 += dxh[q] * xh[q];
                dg[q] += gh * xh[q];
                db[q] += gh;
            }
            const float c1 = wave_sum(s1) * (1.0f / D), c2 = wave_sum(s2) * (1.0f / D);
            float ov[4] = {gx.x, gx.y, gx.z, gx.w};
            for (int q = 0; q < 4; ++q) ov[q] += rstd * (dxh[q] - c1 - xh[q] * c2);
            if (ok) reinterpret_cast<float4 *>(dx_cls + (m0 + row) * D)[lane] = make_float4(ov[0], ov[1], ov[2], ov[3]);
            for (int q = 0; q < 4; ++q) ov[q] = ok ? d1.apply(ov[q], (uint64_t)(m0 + row) * D + 4 * lane + q) : 0.f;
            put4(L.xa, S256, row, 4 * lane, ov);
        }
        for (int q = 0; q < 4; ++q) {
            L.red[w][0][4 * lane + q] = dg[q];
            L.red[w][1][4 * lane + q] = db[q];
        }
    }
    __syncthreads();
    for (int c = tid; c < 2 * D; c += THREADS) {
        const int which = c / D, col = c - which * D;
        G.ln_partial[(int64_t)blockIdx.x * 2 * D + c] = L.red[0][which][col] + L.red[1][which][col] + L.red[2][which][col] + L.red[3][which][col];
    }
    lds_to_T(L.xa, S256, D, (__bf16 *)G.daoT, ld, m0, tid);

    // ---- d o = Wo^T d(out_proj output), bf16 rows for the attention backward
    {
        bf16x8 xf[16];
        load_frags<16>(L.xa, S256, 0, xf, r, h);
        const __bf16 *woT = (const __bf16 *)WT.woT;
        for (int j = 0; j < 2; ++j) {
            const int mt = 2 * w + j;
            const f32x16 acc = tile_gemm<16>(woT, D, 32 * mt, 0, xf, zero_tile(), r, h);
            if (m0 + r < M)
                for (int g = 0; g < 4; ++g) {
                    bf16x4 pk;
                    for (int q = 0; q < 4; ++q) pk[q] = (__bf16)acc[4 * g + q];
                    *reinterpret_cast<bf16x4 *>(d_o + (m0 + r) * D + 32 * mt + 8 * g + 4 * h) = pk;
                }
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// weight gradients from transposed operands: dW[n][k] = sum_m dY^T[n][m] X^T[k][m], bias gradient = row sums of dY^T
// ---------------------------------------------------------------------------------------------------------------------
struct DwTable {
    g2048_dw_job jobs[G2048_DW_MAX_JOBS];
    int first_item[G2048_DW_MAX_JOBS + 1];  // prefix sums of (N/32) * (K/32 + has_bias) * slices
    int n_jobs, slices;
    int64_t ld, m_per_slice;
};

// One wavefront per (job, 32 x 32 output tile, slice of the row axis): both operand fragments are plain 16-byte global
// loads (the row axis is the contiguous one of both transposed operands), f32 partial tile stored as it is; the slices are
// summed by g2048_reduce_jobs.  k-tile K/32 of a job with a bias is the bias tile: B = ones, column 0 of the result.
__global__ void __launch_bounds__(THREADS)
k_dweight_t(DwTable T) {
    const int lane = threadIdx.x & 63, r = lane & 31, h = lane >> 5;
    const int item = blockIdx.x * (THREADS / 64) + (threadIdx.x >> 6);
    if (item >= T.first_item[T.n_jobs]) return;
    int j = 0;
    while (item >= T.first_item[j + 1]) ++j;
    const g2048_dw_job J = T.jobs[j];
    const int kt_n = J.K / 32 + (J.db ? 1 : 0), nt_n = J.N / 32;
    int e = item - T.first_item[j];
    const int slice = e / (nt_n * kt_n);
    e -= slice * nt_n * kt_n;
    const int nt = e / kt_n, kt = e - nt * kt_n;
    const bool bias = kt == J.K / 32;
    const int64_t mb = slice * T.m_per_slice;
    const __bf16 *pa = (const __bf16 *)J.dyT + (int64_t)(32 * nt + r) * T.ld + mb + 8 * h;
    const __bf16 *pb = bias ? pa : (const __bf16 *)J.xT + (int64_t)(32 * kt + r) * T.ld + mb + 8 * h;
    bf16x8 ones;
    for (int q = 0; q < 8; ++q) ones[q] = (__bf16)1.0f;
    f32x16 acc = zero_tile();
    const int steps = (int)(T.m_per_slice / 16);
    int s = 0;
    for (; s + 4 <= steps; s += 4) {
        bf16x8 a[4], b[4];
        for (int u = 0; u < 4; ++u) {
            a[u] = *reinterpret_cast<const bf16x8 *>(pa + 16 * (s + u));
            b[u] = bias ? ones : *reinterpret_cast<const bf16x8 *>(pb + 16 * (s + u));
        }
        for (int u = 0; u < 4; ++u) acc = mfma(a[u], b[u], acc);
    }
    for (; s < steps; ++s)
        acc = mfma(*reinterpret_cast<const bf16x8 *>(pa + 16 * s), bias ? ones : *reinterpret_cast<const bf16x8 *>(pb + 16 * s), acc);
    if (bias) {
        if (r == 0)
            for (int i = 0; i < 16; ++i) J.db[(int64_t)slice * J.N + 32 * nt + rowof(i, h)] = acc[i];
    } else {
        float *out = J.dw + ((int64_t)slice * J.N + 32 * nt) * J.K + 32 * kt + r;
        for (int i = 0; i < 16; ++i) out[(int64_t)rowof(i, h) * J.K] = acc[i];
    }
}

inline int done() {
    const hipError_t e = hipGetLastError();
    return e == hipSuccess ? 0 : -(1000 + (int)e);
}
inline bool mis16(const void *p) { return !p || ((uintptr_t)p & 15); }

}  // namespace

extern "C" int g2048_cls_tail_fwd(const void *o, const float *x_cls, int64_t x_row_stride, const g2048_tail_weights *W,
                                  const g2048_tail_saved *S, float *logits, float *values, int64_t M, float eps, float p_drop,
                                  uint64_t seed, const uint64_t *seed_state, void *stream) {
    if (!W || !S || M <= 0 || !(p_drop >= 0.f && p_drop < 1.f) || (x_row_stride & 3) || x_row_stride < D) return G2048_EINVAL;
    const void *ptrs[] = {o, x_cls, W->wo, W->w1, W->w2, W->a1, W->a2, W->a3, W->c1, W->c2, W->c3, W->bo, W->b1, W->b2, W->ab1,
                          W->ab2, W->cb1, W->cb2, W->ln_g, W->ln_b, S->x_mid, S->masks, S->oT, S->h2T, S->uT, S->featsT, S->a1T,
                          S->a2T, S->c1T, S->c2T, logits};
    for (const void *p : ptrs)
        if (mis16(p)) return G2048_EINVAL;
    if (!values || !S->mean || !S->rstd || ((uintptr_t)values & 3)) return G2048_EINVAL;
    const int64_t blocks = (M + TB - 1) / TB;
    if (S->ld < blocks * TB || (S->ld & 7)) return G2048_EINVAL;
    if (hipFuncSetAttribute(reinterpret_cast<const void *>(k_tail_fwd), hipFuncAttributeMaxDynamicSharedMemorySize,
                            (int)sizeof(TailLds)) != hipSuccess)
        return -(1000 + (int)hipGetLastError());
    hipLaunchKernelGGL(k_tail_fwd, dim3((unsigned)blocks), dim3(THREADS), sizeof(TailLds), (hipStream_t)stream, (const __bf16 *)o, x_cls,
                       x_row_stride, *W, *S, logits, values, M, eps, p_drop, seed, seed_state);
    return done();
}

extern "C" int g2048_cls_tail_bwd(const float *dlogits, const float *dvalues, const g2048_tail_weights_t *WT,
                                  const g2048_tail_saved *S, const g2048_tail_grads *G, void *d_o, float *dx_cls, int64_t M,
                                  float p_drop, uint64_t seed, const uint64_t *seed_state, void *stream) {
    if (!WT || !S || !G || M <= 0 || !(p_drop >= 0.f && p_drop < 1.f)) return G2048_EINVAL;
    const void *ptrs[] = {dlogits, WT->woT, WT->w1T, WT->w2T, WT->a1T, WT->a2T, WT->a3, WT->c1T, WT->c2T, WT->c3, WT->ln_g, S->x_mid,
                          S->masks, G->daoT, G->dzT, G->df2T, G->da1T, G->da2T, G->dlT, G->dc1T, G->dc2T, G->dvT, G->ln_partial, d_o,
                          dx_cls};
    for (const void *p : ptrs)
        if (mis16(p)) return G2048_EINVAL;
    if (!dvalues || !S->mean || !S->rstd || ((uintptr_t)dvalues & 3)) return G2048_EINVAL;
    const int64_t blocks = (M + TB - 1) / TB;
    if (S->ld < blocks * TB || (S->ld & 7)) return G2048_EINVAL;
    if (hipFuncSetAttribute(reinterpret_cast<const void *>(k_tail_bwd), hipFuncAttributeMaxDynamicSharedMemorySize,
                            (int)sizeof(TailLds)) != hipSuccess)
        return -(1000 + (int)hipGetLastError());
    hipLaunchKernelGGL(k_tail_bwd, dim3((unsigned)blocks), dim3(THREADS), sizeof(TailLds), (hipStream_t)stream, dlogits, dvalues, *WT, *S, *G,
                       (__bf16 *)d_o, dx_cls, M, p_drop, seed, seed_state);
    return done();
}

extern "C" int g2048_dweight_t(const g2048_dw_job *jobs, int n_jobs, int64_t ld, int64_t m, int slices, void *stream) {
    if (!jobs || n_jobs <= 0 || n_jobs > G2048_DW_MAX_JOBS || slices <= 0 || m <= 0 || m > ld || (ld & 7) || m % (16 * (int64_t)slices))
        return G2048_EINVAL;
    DwTable T;
    T.n_jobs = n_jobs;
    T.slices = slices;
    T.ld = ld;
    T.m_per_slice = m / slices;
    int items = 0;
    for (int j = 0; j < n_jobs; ++j) {
        const g2048_dw_job &J = jobs[j];
        if (mis16(J.dyT) || mis16(J.xT) || mis16(J.dw) || J.N <= 0 || J.K <= 0 || (J.N & 31) || (J.K & 31) || ((uintptr_t)J.db & 3))
            return G2048_EINVAL;
        T.jobs[j] = J;
        T.first_item[j] = items;
        items += (J.N / 32) * (J.K / 32 + (J.db ? 1 : 0)) * slices;
    }
    for (int j = n_jobs; j <= G2048_DW_MAX_JOBS; ++j) T.first_item[j] = items;
    const int per_block = THREADS / 64;
    hipLaunchKernelGGL(k_dweight_t, dim3((unsigned)((items + per_block - 1) / per_block)), dim3(THREADS), 0, (hipStream_t)stream, T);
    return done();
}
