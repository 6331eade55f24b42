// The second half of a full encoder layer of the PPO update as ONE kernel on gfx950:
//   x_mid = x + dropout(out_proj(a));  h2 = LayerNorm2(x_mid);  u = dropout(relu(linear1(h2)));
//   x_out = x_mid + dropout(linear2(u));  h_next = LayerNorm_next(x_out)
// (reference: nn.TransformerEncoderLayer(norm_first=True), src/ppo/transformer_encoder.py:138-148: the part of _sa_block after the
// attention itself, the residual adds, _ff_block, and norm1 of the following layer).  Round 2 ran this as five launches per
// layer - out_proj GEMM, add + LayerNorm, linear1 + ReLU + dropout, linear2 GEMM, add + LayerNorm: 138 us and 460 MB of HBM traffic
// for [34 816, 256] activations whose pre-activations and branch outputs each crossed HBM twice.  Here every token row is read
// once (a: 0.5 KB, x: 1 KB) and what the backward needs is written once (x_mid, h2, u, x_out, h_next, statistics: 3.6 KB).
//
// Decomposition (the CLS tail's, csrc/g2048_tail.hip, with five token blocks per workgroup): a workgroup of 4 waves owns 160
// tokens; the waves split every Linear's OUTPUT features, tokens sit on lanes (five blocks of 32), the f32 residual rows of the
// wave's 64 features x 160 tokens stay in 160 accumulator registers from the first load to the last store; bf16 activations
// between Linears live in LDS (row-major, padded rows); weights are fragment-packed bf16 shadows streamed from L2 through a
// register ring (one unit = 16 fragments = one 32-row tile over 256 inputs), each fragment multiplied against all five token
// blocks.  34 816 tokens = 218 workgroups: one round on 256 CUs.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <type_traits>

#include "../../include/g2048.h"
#include "g2048_mfma.h"

namespace {

using namespace g2048_mfma;

constexpr int D = 256, FF = 1024, THREADS = 256, FC = 128, NB = 5, TBK = 32 * NB;
constexpr int SA = 2 * D + 16, SU = 2 * FC + 16;  // LDS row strides in bytes
constexpr int N_UNITS = 2 + 2 * (FF / FC);
// two ring slots, one unit ahead: a unit is multiplied against five token blocks (80 MFMAs, >= 2 560 cycles), which covers an L2 round
// trip; the third slot's 64 registers are what the five-block accumulators need
constexpr int SLOTS = 2;

// the layer-norm kernels' dropout decision (csrc/g2048_layernorm.hip keep_elem): sites 1 and 3 use it unchanged so that
// g2048_add_ln_bwd can recompute the same masks from (seed, element index)
struct DropLN {
    uint32_t s0, s1, thr;
    float inv_keep;
    __device__ __forceinline__ float apply(float v, uint64_t idx) const {
        return thr ? (keep_elem(s0, s1, thr, idx) ? v * inv_keep : 0.0f) : v;
    }
};
__device__ __forceinline__ DropLN make_drop_ln(uint64_t seed, const uint64_t *seed_state, float p_drop) {
    uint32_t s0 = (uint32_t)seed, s1 = (uint32_t)(seed >> 32);
    if (seed_state) {
        const uint64_t s = *seed_state;
        s0 ^= (uint32_t)s * 0x9E3779B1u;
        s1 += (uint32_t)(s >> 32) * 0x85EBCA77u + (uint32_t)s;
    }
    return DropLN{s0, s1, (uint32_t)(p_drop * 16777216.0f), 1.0f / (1.0f - p_drop)};
}

struct BlockLds {
    char xa[TBK * SA];      // bf16 [160][256]: a, then h2, then h_next
    char u[TBK * SU];       // bf16 [160][128]: one chunk of the hidden activation
    float bias[D + FF + D]; // bo | b1 | b2
    float ln[4][D];         // gamma2 | beta2 | gamma_next | beta_next
    float part[4][TBK][2];  // LayerNorm partial sums of the four waves (64 features each) per token
};
static_assert(sizeof(BlockLds) <= 160 * 1024, "LDS budget");
constexpr int BO_BO = 0, BO_B1 = D, BO_B2 = D + FF;

// mean / rstd of every token of the tile from the four waves' 64-feature partials, two passes like k_add_ln_fwd
// (sum, then sum of squared deviations).  R[j][b]: this wave's tiles 2w + j, token block b.  Every lane ends with the statistics of
// ITS token of every block.
__device__ __forceinline__ void tile_stats(BlockLds &L, const f32x16 R[2][NB], float mean[NB], float rstd[NB], float eps, int w, int r,
                                           int h) {
    for (int b = 0; b < NB; ++b) {
        float s = 0.f;
        for (int j = 0; j < 2; ++j)
            for (int i = 0; i < 16; ++i) s += R[j][b][i];
        s += __shfl_xor(s, 32);
        if (h == 0) L.part[w][32 * b + r][0] = s;
    }
    lds_barrier();
    for (int b = 0; b < NB; ++b)
        mean[b] = (L.part[0][32 * b + r][0] + L.part[1][32 * b + r][0] + L.part[2][32 * b + r][0] + L.part[3][32 * b + r][0]) * (1.0f / D);
    for (int b = 0; b < NB; ++b) {
        float s = 0.f;
        for (int j = 0; j < 2; ++j)
            for (int i = 0; i < 16; ++i) {
                const float d = R[j][b][i] - mean[b];
                s = __builtin_fmaf(d, d, s);
            }
        s += __shfl_xor(s, 32);
        if (h == 0) L.part[w][32 * b + r][1] = s;
    }
    lds_barrier();
    for (int b = 0; b < NB; ++b)
        rstd[b] = rsqrtf((L.part[0][32 * b + r][1] + L.part[1][32 * b + r][1] + L.part[2][32 * b + r][1] + L.part[3][32 * b + r][1]) *
                             (1.0f / D) + eps);
    lds_barrier();  // the partials are rewritten by the next call
}

// copy the bf16 tile in LDS (rows of `row_bytes`, stride `stride`) to global rows of `ld` elements starting at column col0:
// 16 bytes per thread, whole rows contiguous
__device__ __forceinline__ void store_rows(const char *buf, int stride, int row_bytes, __bf16 *__restrict__ dst, int64_t ld, int col0,
                                           int64_t m0, int64_t M, int tid) {
    const int per_row = row_bytes / 16;
    for (int e = tid; e < TBK * per_row; e += THREADS) {
        const int row = e / per_row, c = e - row * per_row;
        if (m0 + row < M)
            *reinterpret_cast<uint4 *>(dst + (m0 + row) * ld + col0 + 8 * c) = *reinterpret_cast<const uint4 *>(buf + row * stride + 16 * c);
    }
}

__global__ void __launch_bounds__(THREADS, 1)
k_block_fwd(const __bf16 *__restrict__ a, const float *__restrict__ x, g2048_block_weights W, g2048_block_saved S, int64_t M, float eps2,
            float eps_n, float p_drop, uint64_t seed1, uint64_t seed2, uint64_t seed3, const uint64_t *seed_state) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    BlockLds &L = *reinterpret_cast<BlockLds *>(smem);
    const int tid = threadIdx.x, lane = tid & 63, r = lane & 31, h = lane >> 5;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int64_t m0 = (int64_t)blockIdx.x * TBK;
    const DropLN d1 = make_drop_ln(seed1, seed_state, p_drop), d3 = make_drop_ln(seed3, seed_state, p_drop);
    const Drop d2 = make_drop(seed2, seed_state, p_drop);
    uint16_t *const bits_wg = reinterpret_cast<uint16_t *>(S.bits) + (int64_t)blockIdx.x * (FF / 32) * NB * 64;

    Ring R_;
    auto issue = [&](auto ic) __attribute__((always_inline)) {
        constexpr int i = decltype(ic)::value;
        if constexpr (i < N_UNITS) {
            constexpr int slot = i % SLOTS;
            if constexpr (i < 2) {
                fetch_unit<slot>(R_, unit_ptr(W.wo, D, 32 * (2 * w + i), 0, lane), NEXT_8_STEPS);
            } else {
                constexpr int c = (i - 2) / 2;
                if constexpr ((i - 2) % 2 == 0) fetch_unit<slot>(R_, unit_ptr(W.w1, D, 32 * (4 * c + w), 0, lane), NEXT_8_STEPS);
                else fetch_unit<slot>(R_, unit_ptr(W.w2, FF, 32 * (2 * w), FC * c, lane), next_row_tile(FF));
            }
        }
        sched_fence();
    };

    // ---- prologue: the tile's loads first (in-order return: see csrc/g2048_tail.hip), then the ring, then the LDS writes
    f32x16 R[2][NB];  // the residual rows of this wave's 64 features: x -> x_mid -> (+ linear2) -> x_out
    float in_b[6], in_ln[4];
    {
        const float *src[3] = {W.bo, W.b1, W.b2};
        const int len[3] = {D, FF, D};
        int n = 0;
#pragma unroll
        for (int k = 0; k < 3; ++k)
#pragma unroll
            for (int i = 0; i < len[k] / THREADS; ++i) in_b[n++] = src[k][tid + i * THREADS];
        in_ln[0] = W.ln2_g[tid]; in_ln[1] = W.ln2_b[tid]; in_ln[2] = W.lnn_g[tid]; in_ln[3] = W.lnn_b[tid];
    }
    constexpr int A_PASSES = TBK * 32 / THREADS;  // 20 passes of 8 rows
    uint4 in_a[A_PASSES / 2];
    auto load_a = [&](int p0) __attribute__((always_inline)) {
#pragma unroll
        for (int p = 0; p < A_PASSES / 2; ++p) {
            const int row = 8 * (p0 + p) + (tid >> 5), ch = tid & 31;
            in_a[p] = make_uint4(0u, 0u, 0u, 0u);
            if (m0 + row < M) in_a[p] = *reinterpret_cast<const uint4 *>(a + (m0 + row) * D + 8 * ch);
        }
    };
    auto put_a = [&](int p0) __attribute__((always_inline)) {
#pragma unroll
        for (int p = 0; p < A_PASSES / 2; ++p)
            *reinterpret_cast<uint4 *>(L.xa + (8 * (p0 + p) + (tid >> 5)) * SA + 16 * (tid & 31)) = in_a[p];
    };
    load_a(0);
    sched_fence();
    put_a(0);
    load_a(A_PASSES / 2);
    sched_fence();
    issue(std::integral_constant<int, 0>{});
    put_a(A_PASSES / 2);
    {
        const int off[3] = {BO_BO, BO_B1, BO_B2};
        const int len[3] = {D, FF, D};
        int n = 0;
#pragma unroll
        for (int k = 0; k < 3; ++k)
#pragma unroll
            for (int i = 0; i < len[k] / THREADS; ++i) L.bias[off[k] + tid + i * THREADS] = in_b[n++];
        for (int k = 0; k < 4; ++k) L.ln[k][tid] = in_ln[k];
    }
    lds_barrier();

    // ---- out_proj + dropout + residual: x_mid = x + dropout(bf16(Wo a + bo)); both of this wave's tiles per token block, so that the
    //      block's operand fragments are read once
    static_for<0, 2>([&](auto jc) __attribute__((always_inline)) {
        constexpr int j = decltype(jc)::value;
        issue(std::integral_constant<int, j + 1>{});
        static_for<0, NB>([&](auto bc) __attribute__((always_inline)) {
            constexpr int b = decltype(bc)::value;
            const int64_t m = m0 + 32 * b + r;
            f32x4 xb[4];  // this tile's 16 residual values of the lane's row: in flight behind the block's MFMAs
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                xb[g] = f32x4{0.f, 0.f, 0.f, 0.f};
                if (m < M) xb[g] = *reinterpret_cast<const f32x4 *>(x + m * D + 32 * (2 * w + j) + 8 * g + 4 * h);
            }
            f32x16 acc = bias_tile(L.bias + BO_BO, 32 * (2 * w + j), h);
            bf16x8 xf[8];
            load_frags<8>(L.xa + 32 * b * SA, SA, 0, xf, r, h);
            acc = mm8<j % SLOTS, 0>(R_, xf, acc);
            load_frags<8>(L.xa + 32 * b * SA, SA, 128, xf, r, h);
            acc = mm8<j % SLOTS, 1>(R_, xf, acc);
            for (int g = 0; g < 4; ++g) {
                const int f0 = 32 * (2 * w + j) + 8 * g + 4 * h;
                f32x4 v = xb[g];
                for (int q = 0; q < 4; ++q) {
                    v[q] += d1.apply((float)(__bf16)acc[4 * g + q], (uint64_t)m * D + f0 + q);
                    R[j][b][4 * g + q] = v[q];
                }
                if (m < M) *reinterpret_cast<f32x4 *>(S.x_mid + m * D + f0) = v;  // (the backward's LayerNorm needs x_mid)
            }
        });
    });

    // ---- LayerNorm2 -> h2 (LDS xa: every wave is done with `a` after the barriers inside tile_stats)
    {
        float mean[NB], rstd[NB];
        tile_stats(L, R, mean, rstd, eps2, w, r, h);
        for (int b = 0; b < NB; ++b) {
            const bool ok = m0 + 32 * b + r < M;
            if (w == 0 && h == 0 && ok) {
                S.mean2[m0 + 32 * b + r] = mean[b];
                S.rstd2[m0 + 32 * b + r] = rstd[b];
            }
            for (int j = 0; j < 2; ++j)
                for (int g = 0; g < 4; ++g) {
                    const int f0 = 32 * (2 * w + j) + 8 * g + 4 * h;
                    const f32x4 gm = *reinterpret_cast<const f32x4 *>(&L.ln[0][f0]), bt = *reinterpret_cast<const f32x4 *>(&L.ln[1][f0]);
                    float v[4];
                    for (int q = 0; q < 4; ++q) v[q] = ok ? (R[j][b][4 * g + q] - mean[b]) * rstd[b] * gm[q] + bt[q] : 0.f;
                    put4(L.xa, SA, 32 * b + r, f0, v);
                }
        }
    }
    lds_barrier();
    store_rows(L.xa, SA, 2 * D, (__bf16 *)S.h2, D, 0, m0, M, tid);

    // ---- feed-forward in chunks of 128 hidden units: u = dropout(relu(W1 h2 + b1)) -> LDS + HBM, R += W2[:, chunk] u
    static_for<0, FF / FC>([&](auto cc) __attribute__((always_inline)) {
        constexpr int c = decltype(cc)::value, u1 = 2 + 2 * c, u2 = 3 + 2 * c;
        const int ht = 4 * c + w;  // hidden tile of this wave
        issue(std::integral_constant<int, u1 + 1>{});  // one unit ahead of the one being multiplied, at all times
#pragma unroll
        for (int b = 0; b < NB; ++b) {
            f32x16 z = bias_tile(L.bias + BO_B1, 32 * ht, h);
            for (int half = 0; half < 2; ++half) {
                bf16x8 xf[8];
                load_frags<8>(L.xa + 32 * b * SA, SA, 128 * half, xf, r, h);
                z = half == 0 ? mm8<u1 % SLOTS, 0>(R_, xf, z) : mm8<u1 % SLOTS, 1>(R_, xf, z);
            }
            const bool ok = m0 + 32 * b + r < M;
            uint32_t bits = 0;
            for (int g = 0; g < 4; ++g) {
                float v[4];
                for (int q = 0; q < 4; ++q) v[q] = ok ? fmaxf(z[4 * g + q], 0.f) : 0.f;
                d2.apply4(v, (uint64_t)(m0 + 32 * b + r) * FF + 32 * ht + 8 * g + 4 * h);
                for (int q = 0; q < 4; ++q) bits |= (uint32_t)((float)(__bf16)v[q] != 0.f) << (4 * g + q);
                put4(L.u, SU, 32 * b + r, 32 * w + 8 * g + 4 * h, v);
            }
            bits_wg[(ht * NB + b) * 64 + lane] = (uint16_t)bits;
        }
        lds_barrier();  // chunk c of every wave visible
        store_rows(L.u, SU, 2 * FC, (__bf16 *)S.u, FF, FC * c, m0, M, tid);
        issue(std::integral_constant<int, u2 + 1>{});
#pragma unroll
        for (int b = 0; b < NB; ++b) {
            bf16x8 uf[8];
            load_frags<8>(L.u + 32 * b * SU, SU, 0, uf, r, h);
            R[0][b] = mm8<u2 % SLOTS, 0>(R_, uf, R[0][b]);
            R[1][b] = mm8<u2 % SLOTS, 1>(R_, uf, R[1][b]);
        }
        lds_barrier();  // the chunk buffer is rewritten by the next chunk
    });

    // ---- x_out = x_mid + dropout(bf16(linear2 output)): the branch is what the accumulators gained over x_mid (re-read from
    //      L2) plus the bias; the difference is exact to one ulp of x_mid, three decimal orders below the bf16 rounding that follows
    static_for<0, NB>([&](auto bc) __attribute__((always_inline)) {
        constexpr int b = decltype(bc)::value;
        const int64_t m = m0 + 32 * b + r;
        f32x4 xb[8];
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                xb[4 * j + g] = f32x4{0.f, 0.f, 0.f, 0.f};
                if (m < M) xb[4 * j + g] = *reinterpret_cast<const f32x4 *>(S.x_mid + m * D + 32 * (2 * w + j) + 8 * g + 4 * h);
            }
        for (int j = 0; j < 2; ++j)
            for (int g = 0; g < 4; ++g) {
                const int f0 = 32 * (2 * w + j) + 8 * g + 4 * h;
                const f32x4 xm = xb[4 * j + g], b2 = *reinterpret_cast<const f32x4 *>(L.bias + BO_B2 + f0);
                f32x4 v;
                for (int q = 0; q < 4; ++q) {
                    const float f = (float)(__bf16)((R[j][b][4 * g + q] - xm[q]) + b2[q]);
                    v[q] = xm[q] + d3.apply(f, (uint64_t)m * D + f0 + q);
                    R[j][b][4 * g + q] = v[q];
                }
                if (m < M) *reinterpret_cast<f32x4 *>(S.x_out + m * D + f0) = v;
            }
    });

    // ---- the next LayerNorm -> h_next
    {
        float mean[NB], rstd[NB];
        tile_stats(L, R, mean, rstd, eps_n, w, r, h);
        for (int b = 0; b < NB; ++b) {
            const bool ok = m0 + 32 * b + r < M;
            if (w == 0 && h == 0 && ok) {
                S.mean_n[m0 + 32 * b + r] = mean[b];
                S.rstd_n[m0 + 32 * b + r] = rstd[b];
            }
            for (int j = 0; j < 2; ++j)
                for (int g = 0; g < 4; ++g) {
                    const int f0 = 32 * (2 * w + j) + 8 * g + 4 * h;
                    const f32x4 gm = *reinterpret_cast<const f32x4 *>(&L.ln[2][f0]), bt = *reinterpret_cast<const f32x4 *>(&L.ln[3][f0]);
                    float v[4];
                    for (int q = 0; q < 4; ++q) v[q] = ok ? (R[j][b][4 * g + q] - mean[b]) * rstd[b] * gm[q] + bt[q] : 0.f;
                    put4(L.xa, SA, 32 * b + r, f0, v);
                }
        }
    }
    lds_barrier();
    store_rows(L.xa, SA, 2 * D, (__bf16 *)S.h_next, D, 0, m0, M, tid);
}

inline int done() {
    const hipError_t e = hipGetLastError();
    return e == hipSuccess ? 0 : -(1000 + (int)e);
}
inline bool mis16(const void *p) { return !p || ((uintptr_t)p & 15); }

}  // namespace

extern "C" int64_t g2048_block_bits_bytes(int64_t M) {
    return M <= 0 ? 0 : ((M + TBK - 1) / TBK) * (int64_t)(FF / 32) * NB * 64 * (int64_t)sizeof(uint16_t);
}

extern "C" int g2048_block_fwd(const void *a, const float *x, const g2048_block_weights *W, const g2048_block_saved *S, int64_t M,
                               float eps2, float eps_next, float p_drop, uint64_t seed1, uint64_t seed2, uint64_t seed3,
                               const uint64_t *seed_state, void *stream) {
    if (!W || !S || M <= 0 || !(p_drop >= 0.f && p_drop < 1.f)) return G2048_EINVAL;
    const void *ptrs[] = {a, x, W->wo, W->w1, W->w2, W->bo, W->b1, W->b2, W->ln2_g, W->ln2_b, W->lnn_g, W->lnn_b,
                          S->x_mid, S->h2, S->u, S->bits, S->x_out, S->h_next};
    for (const void *p : ptrs)
        if (mis16(p)) return G2048_EINVAL;
    if (!S->mean2 || !S->rstd2 || !S->mean_n || !S->rstd_n) return G2048_EINVAL;
    if (hipFuncSetAttribute(reinterpret_cast<const void *>(k_block_fwd), hipFuncAttributeMaxDynamicSharedMemorySize,
                            (int)sizeof(BlockLds)) != hipSuccess)
        return -(1000 + (int)hipGetLastError());
    hipLaunchKernelGGL(k_block_fwd, dim3((unsigned)((M + TBK - 1) / TBK)), dim3(THREADS), sizeof(BlockLds), (hipStream_t)stream,
                       (const __bf16 *)a, x, *W, *S, M, eps2, eps_next, p_drop, seed1, seed2, seed3, seed_state);
    return done();
}
