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

#include <type_traits>

#include "../../include/g2048.h"
#include "g2048_mfma.h"

namespace {

using namespace g2048_mfma;

constexpr int D = 256, FF = 1024, HID = 512, TB = 32, THREADS = 256, FC = 128;
constexpr int S256 = 2 * 256 + 16, S512 = 2 * 512 + 16, S128 = 2 * 128 + 16;  // LDS row strides in bytes (+16: bank spread)
constexpr int XM_S = S512 / 4;                                                  // the f32 [32][256] tile uses the same rows
constexpr int MT_FFN = 0, MT_A1 = 32, MT_A2 = 48, MT_C1 = 64, MT_C2 = 80, N_MASK_TILES = G2048_TAIL_MASK_TILES;
static_assert(N_MASK_TILES == 96, "mask tile table");

// the transposed copy the weight-gradient kernel reads, X^T[f0 + f][m0 + row] = buf[row][f] for f < NFEAT, row < 32, stored
// fragment-packed with `steps` = ld / 16 k-steps per row tile: 16-byte stores of 8 rows each, consecutive threads = consecutive
// slots.  Call between two barriers; NFEAT and f0 multiples of 32, m0 a multiple of 32.  32-bit offsets (buffers < 2^31 elements).
template <int NFEAT>
__device__ __forceinline__ void lds_to_T(const char *buf, int stride, __bf16 *__restrict__ dstT, int f0, int steps, int step0, int tid) {
    static_assert(NFEAT % 64 == 0 && (NFEAT & (NFEAT - 1)) == 0, "feature count");
#pragma unroll
    for (int e = tid; e < NFEAT * 4; e += THREADS) {
        const int f = e & (NFEAT - 1), g = e / NFEAT, F = f0 + f;
        const char *src = buf + 8 * g * stride + 2 * f;
        uint32_t wv[4];
#pragma unroll
        for (int q = 0; q < 4; ++q)
            wv[q] = (uint32_t)*reinterpret_cast<const uint16_t *>(src + (2 * q) * stride) |
                    ((uint32_t)*reinterpret_cast<const uint16_t *>(src + (2 * q + 1) * stride) << 16);
        const uint32_t off = ((((uint32_t)(F >> 5) * steps + step0 + (g >> 1)) * 2 + (g & 1)) * 32 + (F & 31)) * 8;
        *reinterpret_cast<uint4 *>(dstT + off) = make_uint4(wv[0], wv[1], wv[2], wv[3]);
    }
}

struct TailLds {
    char xa[TB * S512];     // bf16 activations, rows of up to 512
    char xb[TB * S512];
    char xc[TB * S512];     // first the f32 [32][256] residual tile (same row stride), later bf16 activations
    char u[2][TB * S128];   // feed-forward hidden chunk (128 units), double-buffered
    float red[4][2][D];     // backward: LayerNorm gradient partials of the four waves
    float dl[TB][8];        // backward: d logits (4) and d value of the tile's rows
    // staged once at kernel start, so that no tile waits for a dependent global load in front of its MFMAs (28 bias tiles forward,
    // 48 mask words backward at ~1 us of exposed L2 latency each)
    union {
        float bias[D + FF + D + 4 * HID];            // forward: bo | b1 | b2 | ab1 | ab2 | cb1 | cb2
        uint16_t masks[G2048_TAIL_MASK_TILES * 64];  // backward: this workgroup's ReLU / dropout bit words
    };
    __bf16 w3[5 * HID];     // a3 (4 rows) | c3 (1 row)
};
constexpr int BO_BO = 0, BO_B1 = D, BO_B2 = D + FF, BO_AB1 = 2 * D + FF, BO_AB2 = BO_AB1 + HID, BO_CB1 = BO_AB2 + HID, BO_CB2 = BO_CB1 + HID;
static_assert(sizeof(TailLds) <= 160 * 1024, "LDS budget");

// ---------------------------------------------------------------------------------------------------------------------
// weight stream: every wave walks a fixed list of UNITS, one unit = 16 operand fragments (16 KB per wave) = one 32-row
// weight tile over 256 inputs, or two 32-row tiles over 128 inputs.  Unit i is fetched into ring slot i % RING while unit
// i - DIST is being multiplied, across phase boundaries and barriers (weights depend on nothing): without this every tile
// exposed one L2 round trip (~2 us) before its 0.2 us of MFMAs and the kernels took 158 / 133 us instead of ~30.
// ---------------------------------------------------------------------------------------------------------------------
constexpr int N_UNITS = 42;

// ---------------------------------------------------------------------------------------------------------------------
// forward
// ---------------------------------------------------------------------------------------------------------------------
// units of wave w: 0-1 out_proj tiles 2w, 2w+1 | 2+2c linear1 tile 4c+w, 3+2c linear2 tiles 2w, 2w+1 over chunk c (c < 8) |
// 18-21 actor L1 tiles 4w+t, 22-29 actor L2 tile 4w+t halves 0/1 | 30-41 the same for the critic
__global__ void __launch_bounds__(THREADS, 1)
k_tail_fwd(const __bf16 *__restrict__ o, const float *__restrict__ x_cls, int64_t x_rs, g2048_tail_weights W, g2048_tail_saved S,
           float *__restrict__ logits, float *__restrict__ values, int64_t M, float eps, float p_drop, uint64_t seed,
           const uint64_t *seed_state) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    TailLds &L = *reinterpret_cast<TailLds *>(smem);
    const int tid = threadIdx.x, lane = tid & 63, r = lane & 31, h = lane >> 5;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int64_t m0 = (int64_t)blockIdx.x * TB, ld = S.ld;
    const int steps_ld = (int)(ld >> 4), step_m0 = (int)(m0 >> 4);
    const bool valid = m0 + r < M;  // this lane's row in the accumulator layout
    const Drop drop = make_drop(seed, seed_state, p_drop);
    float *const xm = reinterpret_cast<float *>(L.xc);
    uint16_t *const masks_wg = reinterpret_cast<uint16_t *>(S.masks) + (int64_t)blockIdx.x * N_MASK_TILES * 64;
    // gridDim.y == 2: two workgroups per 32 boards, one per head.  Both walk the shared part (out_proj .. features: 1.2 MB of weights),
    // workgroup 0 saves what the backward needs of it and runs the actor, workgroup 1 runs the critic (0.8 MB each): 2.0 instead of
    // 2.8 MB through one CU's memory path, which is what the kernel's time is made of (73 -> 52 us).  gridDim.y == 1: both heads here.
    const int role = gridDim.y == 2 ? (int)blockIdx.y : -1;
    const bool save = role <= 0;

    Ring R;
    auto issue = [&](auto ic) __attribute__((always_inline)) {
        constexpr int i = decltype(ic)::value;
        if constexpr (i < N_UNITS) {
            constexpr int slot = i % RING;
            if constexpr (i < 2) {
                fetch_unit<slot>(R, unit_ptr(W.wo, D, 32 * (2 * w + i), 0, lane), NEXT_8_STEPS);
            } else if constexpr (i < 18) {
                constexpr int c = (i - 2) / 2;
                if constexpr ((i - 2) % 2 == 0) fetch_unit<slot>(R, unit_ptr(W.w1, D, 32 * (4 * c + w), 0, lane), NEXT_8_STEPS);
                else fetch_unit<slot>(R, unit_ptr(W.w2, FF, 32 * (2 * w), FC * c, lane), next_row_tile(FF));
            } else if (!(role == 0 && i >= 30)) {  // (units 30.. are the critic's: the other workgroup's when the heads are split)
                constexpr int j = (i - 18) % 12;
                // the critic's workgroup walks the critic's units right behind the shared part: unit 18 + j of its list = unit 30 + j
                const bool critic = i >= 30 || role == 1;
                const void *l1 = critic ? W.c1 : W.a1, *l2 = critic ? W.c2 : W.a2;
                if constexpr (j < 4) fetch_unit<slot>(R, unit_ptr(l1, D, 32 * (4 * w + j), 0, lane), NEXT_8_STEPS);
                else fetch_unit<slot>(R, unit_ptr(l2, HID, 32 * (4 * w + (j - 4) / 2), 256 * ((j - 4) % 2), lane), NEXT_8_STEPS);
            }
        }
        sched_fence();
    };
    // ---- prologue.  Vector-memory loads return in order: a load consumed right after the ring's fetches would wait for all of
    // them, and a loop of load -> LDS write pairs is one exposed L2 round trip per iteration (35 of them in the first version
    // of this prologue).  So: every load of the prologue first, into registers; then the ring's first two units; then the LDS
    // writes, which wait for nothing younger than themselves.
    uint4 in_o[4];
    float4 in_x[8];
    float in_b[14];
    bf16x8 in_w3[2];
    {
        const float *src[7] = {W.bo, W.b1, W.b2, W.ab1, W.ab2, W.cb1, W.cb2};
        const int len[7] = {D, FF, D, HID, HID, HID, HID};
        int n = 0;
#pragma unroll
        for (int k = 0; k < 7; ++k)
#pragma unroll
            for (int i = 0; i < len[k] / THREADS; ++i) in_b[n++] = src[k][tid + i * THREADS];
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int e = tid + i * THREADS;  // 320 vectors of 8: a3 rows 0..3, then c3
            in_w3[i] = e < 4 * HID / 8 ? reinterpret_cast<const bf16x8 *>(W.a3)[e]
                                       : reinterpret_cast<const bf16x8 *>(W.c3)[e < 5 * HID / 8 ? e - 4 * HID / 8 : 0];
        }
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            const int row = 8 * p + (tid >> 5), ch = tid & 31;
            in_o[p] = make_uint4(0u, 0u, 0u, 0u);
            if (m0 + row < M) in_o[p] = *reinterpret_cast<const uint4 *>(o + (m0 + row) * D + 8 * ch);
        }
#pragma unroll
        for (int p = 0; p < 8; ++p) {
            const int row = 4 * p + (tid >> 6), c4 = tid & 63;
            in_x[p] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (m0 + row < M) in_x[p] = *reinterpret_cast<const float4 *>(x_cls + (m0 + row) * x_rs + 4 * c4);
        }
    }
    sched_fence();
    issue(std::integral_constant<int, 0>{});
    issue(std::integral_constant<int, 1>{});
    {
        const int off[7] = {BO_BO, BO_B1, BO_B2, BO_AB1, BO_AB2, BO_CB1, BO_CB2};
        const int len[7] = {D, FF, D, HID, HID, HID, HID};
        int n = 0;
#pragma unroll
        for (int k = 0; k < 7; ++k)
#pragma unroll
            for (int i = 0; i < len[k] / THREADS; ++i) L.bias[off[k] + tid + i * THREADS] = in_b[n++];
#pragma unroll
        for (int i = 0; i < 2; ++i)
            if (tid + i * THREADS < 5 * HID / 8) reinterpret_cast<bf16x8 *>(L.w3)[tid + i * THREADS] = in_w3[i];
        // the tile's inputs: attention output rows -> xa (bf16), residual CLS rows -> xm (f32); rows past M are zero
#pragma unroll
        for (int p = 0; p < 4; ++p) *reinterpret_cast<uint4 *>(L.xa + (8 * p + (tid >> 5)) * S256 + 16 * (tid & 31)) = in_o[p];
#pragma unroll
        for (int p = 0; p < 8; ++p) *reinterpret_cast<float4 *>(xm + (4 * p + (tid >> 6)) * XM_S + 4 * (tid & 63)) = in_x[p];
    }
    lds_barrier();
    if (save) lds_to_T<D>(L.xa, S256, (__bf16 *)S.oT, 0, steps_ld, step_m0, tid);

    // ---- out_proj, dropout, residual add: x_mid = x + dropout(bf16(Wo o + bo))
    {
        bf16x8 xf[16];
        load_frags<16>(L.xa, S256, 0, xf, r, h);
        const Drop d1 = drop.site(1);
        static_for<0, 2>([&](auto jc) __attribute__((always_inline)) {
            constexpr int j = decltype(jc)::value;
            const int mt = 2 * w + j;
            issue(std::integral_constant<int, j + DIST>{});
            const f32x16 acc = mm16<j % RING>(R, xf, bias_tile(L.bias + BO_BO, 32 * mt, h));
            for (int g = 0; g < 4; ++g) {
                const int f0 = 32 * mt + 8 * g + 4 * h;
                f32x4 x = *reinterpret_cast<const f32x4 *>(xm + r * XM_S + f0);
                float av[4];
                for (int q = 0; q < 4; ++q) av[q] = (float)(__bf16)acc[4 * g + q];
                d1.apply4(av, (uint64_t)(m0 + r) * D + f0);
                for (int q = 0; q < 4; ++q) x[q] += av[q];
                *reinterpret_cast<f32x4 *>(xm + r * XM_S + f0) = x;
            }
        });
    }
    lds_barrier();
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
            if (ok && save) {
                reinterpret_cast<float4 *>(S.x_mid + (m0 + row) * D)[lane] = v;
                if (lane == 0) {
                    S.mean[m0 + row] = mean;
                    S.rstd[m0 + row] = rstd;
                }
            }
        }
    }
    lds_barrier();
    if (save) lds_to_T<D>(L.xb, S256, (__bf16 *)S.h2T, 0, steps_ld, step_m0, tid);

    // ---- feed-forward: u = dropout(relu(W1 h2 + b1)) in chunks of 128 hidden units, f = W2 u + b2 accumulated per chunk
    {
        bf16x8 xf[16];
        load_frags<16>(L.xb, S256, 0, xf, r, h);
        const Drop d2 = drop.site(2), d3 = drop.site(3);
        f32x16 acc2[2];
        for (int j = 0; j < 2; ++j) acc2[j] = bias_tile(L.bias + BO_B2, 32 * (2 * w + j), h);
        static_for<0, FF / FC>([&](auto cc) __attribute__((always_inline)) {
            constexpr int c = decltype(cc)::value, u1 = 2 + 2 * c, u2 = 3 + 2 * c;
            const int ht = 4 * c + w;  // hidden tile of this wave
            char *ub = L.u[c & 1];
            issue(std::integral_constant<int, u1 + DIST>{});
            const f32x16 z = mm16<u1 % RING>(R, xf, bias_tile(L.bias + BO_B1, 32 * ht, h));
            uint32_t bits = 0;
            for (int g = 0; g < 4; ++g) {
                float v[4];
                for (int q = 0; q < 4; ++q) v[q] = valid ? fmaxf(z[4 * g + q], 0.f) : 0.f;
                d2.apply4(v, (uint64_t)(m0 + r) * FF + 32 * ht + 8 * g + 4 * h);
                for (int q = 0; q < 4; ++q) bits |= (uint32_t)((float)(__bf16)v[q] != 0.f) << (4 * g + q);
                put4(ub, S128, r, 32 * w + 8 * g + 4 * h, v);
            }
            if (save) masks_wg[(MT_FFN + ht) * 64 + lane] = (uint16_t)bits;
            lds_barrier();  // chunk c of every wave visible
            if (save) lds_to_T<FC>(ub, S128, (__bf16 *)S.uT, FC * c, steps_ld, step_m0, tid);
            bf16x8 uf[8];
            load_frags<8>(ub, S128, 0, uf, r, h);
            issue(std::integral_constant<int, u2 + DIST>{});
            acc2[0] = mm8<u2 % RING, 0>(R, uf, acc2[0]);
            acc2[1] = mm8<u2 % RING, 1>(R, uf, acc2[1]);
            // a wave reaches the writes of chunk c + 2 (same buffer) only after the barrier of chunk c + 1, which every wave
            // passes after these reads
        });
        // ---- features = bf16(x_mid + dropout(bf16(f))) -> xa
        for (int j = 0; j < 2; ++j) {
            const int mt = 2 * w + j;
            for (int g = 0; g < 4; ++g) {
                const int f0 = 32 * mt + 8 * g + 4 * h;
                const f32x4 x = *reinterpret_cast<const f32x4 *>(xm + r * XM_S + f0);
                float v[4];
                for (int q = 0; q < 4; ++q) v[q] = (float)(__bf16)acc2[j][4 * g + q];
                d3.apply4(v, (uint64_t)(m0 + r) * D + f0);
                for (int q = 0; q < 4; ++q) v[q] = valid ? x[q] + v[q] : 0.f;
                put4(L.xa, S256, r, f0, v);
            }
        }
    }
    lds_barrier();
    if (save) lds_to_T<D>(L.xa, S256, (__bf16 *)S.featsT, 0, steps_ld, step_m0, tid);
    lds_barrier();  // xc (the f32 residual tile) is dead from here on: the heads write bf16 rows into it

    // ---- heads: Linear(256 -> 512) + ReLU, Linear(512 -> 512) + ReLU, Linear(512 -> n_out, no bias) on the features in xa
    auto head = [&](auto u0c, auto nout_c, const float *b1, const float *b2, const __bf16 *w3, int mt1, int mt2, __bf16 *h1T, __bf16 *h2T,
                    float *out) __attribute__((always_inline)) {
        constexpr int U0 = decltype(u0c)::value, N_OUT = decltype(nout_c)::value;
        {
            bf16x8 ff[16];
            load_frags<16>(L.xa, S256, 0, ff, r, h);
            static_for<0, 4>([&](auto tc) __attribute__((always_inline)) {
                constexpr int t = decltype(tc)::value, u = U0 + t;
                const int mt = 4 * w + t;
                issue(std::integral_constant<int, u + DIST>{});
                const f32x16 acc = mm16<u % RING>(R, ff, bias_tile(b1, 32 * mt, h));
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
            });
        }
        lds_barrier();
        lds_to_T<HID>(L.xb, S512, h1T, 0, steps_ld, step_m0, tid);
        {
            static_for<0, 4>([&](auto tc) __attribute__((always_inline)) {
                constexpr int t = decltype(tc)::value, u = U0 + 4 + 2 * t;
                const int mt = 4 * w + t;
                bf16x8 af[16];  // (re-read per tile: 128 registers for all 32 fragments would push the weight ring into scratch)
                issue(std::integral_constant<int, u + DIST>{});
                load_frags<16>(L.xb, S512, 0, af, r, h);
                f32x16 acc = mm16<u % RING>(R, af, bias_tile(b2, 32 * mt, h));
                issue(std::integral_constant<int, u + 1 + DIST>{});
                load_frags<16>(L.xb, S512, 256, af, r, h);
                acc = mm16<(u + 1) % RING>(R, af, acc);
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
            });
        }
        lds_barrier();
        lds_to_T<HID>(L.xc, S512, h2T, 0, steps_ld, step_m0, tid);
        {   // the output layer (4 logits / 1 value) on the vector ALU: 8 threads per row, 64 inputs each
            const int row = tid >> 3, part = tid & 7;
            float s[N_OUT];
            for (int oo = 0; oo < N_OUT; ++oo) s[oo] = 0.f;
            for (int c = 0; c < 8; ++c) {
                const bf16x8 x = *reinterpret_cast<const bf16x8 *>(L.xc + row * S512 + 2 * (64 * part + 8 * c));
                for (int oo = 0; oo < N_OUT; ++oo) {
                    const bf16x8 wv = *reinterpret_cast<const bf16x8 *>(w3 + oo * HID + 64 * part + 8 * c);
                    for (int j = 0; j < 8; ++j) s[oo] = __builtin_fmaf((float)x[j], (float)wv[j], s[oo]);
                }
            }
            for (int oo = 0; oo < N_OUT; ++oo) {
                s[oo] += __shfl_xor(s[oo], 1);
                s[oo] += __shfl_xor(s[oo], 2);
                s[oo] += __shfl_xor(s[oo], 4);
            }
            if (part == 0 && m0 + row < M)
                for (int oo = 0; oo < N_OUT; ++oo) out[(m0 + row) * N_OUT + oo] = s[oo];
        }
        lds_barrier();  // xb / xc are free again
    };
    if (role != 1)
        head(std::integral_constant<int, 18>{}, std::integral_constant<int, 4>{}, L.bias + BO_AB1, L.bias + BO_AB2, L.w3, MT_A1, MT_A2,
             (__bf16 *)S.a1T, (__bf16 *)S.a2T, logits);
    if (role != 0)
        head(std::integral_constant<int, 30>{}, std::integral_constant<int, 1>{}, L.bias + BO_CB1, L.bias + BO_CB2, L.w3 + 4 * HID, MT_C1, MT_C2,
             (__bf16 *)S.c1T, (__bf16 *)S.c2T, values);
}

// ---------------------------------------------------------------------------------------------------------------------
// backward
// ---------------------------------------------------------------------------------------------------------------------
// units of wave w: 0-7 actor L2^T tile 4w+t halves 0/1, 8-11 actor L1^T tile 2w+j halves 0/1 | 12-23 the same for the critic |
// 24+2c linear2^T tile 4c+w, 25+2c linear1^T tiles 2w, 2w+1 over chunk c (c < 8) | 40-41 out_proj^T tiles 2w, 2w+1
__global__ void __launch_bounds__(THREADS, 1)
k_tail_bwd(const float *__restrict__ dlogits, const float *__restrict__ dvalues, g2048_tail_weights_t WT, g2048_tail_saved S,
           g2048_tail_grads G, __bf16 *__restrict__ d_o, float *__restrict__ dx_cls, int64_t M, float p_drop, uint64_t seed,
           const uint64_t *seed_state) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    TailLds &L = *reinterpret_cast<TailLds *>(smem);
    const int tid = threadIdx.x, lane = tid & 63, r = lane & 31, h = lane >> 5;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int64_t m0 = (int64_t)blockIdx.x * TB, ld = S.ld;
    const int steps_ld = (int)(ld >> 4), step_m0 = (int)(m0 >> 4);
    const Drop drop = make_drop(seed, seed_state, p_drop);
    float *const xm = reinterpret_cast<float *>(L.xc);
    const uint16_t *const masks_wg = reinterpret_cast<const uint16_t *>(S.masks) + (int64_t)blockIdx.x * N_MASK_TILES * 64;

    Ring R;
    auto issue = [&](auto ic) __attribute__((always_inline)) {
        constexpr int i = decltype(ic)::value;
        if constexpr (i < N_UNITS) {
            constexpr int slot = i % RING;
            if constexpr (i < 24) {
                constexpr int j = i % 12;
                const void *l1 = i < 12 ? WT.a1T : WT.c1T, *l2 = i < 12 ? WT.a2T : WT.c2T;
                if constexpr (j < 8) fetch_unit<slot>(R, unit_ptr(l2, HID, 32 * (4 * w + j / 2), 256 * (j % 2), lane), NEXT_8_STEPS);
                else fetch_unit<slot>(R, unit_ptr(l1, HID, 32 * (2 * w + (j - 8) / 2), 256 * ((j - 8) % 2), lane), NEXT_8_STEPS);
            } else if constexpr (i < 40) {
                constexpr int c = (i - 24) / 2;
                if constexpr ((i - 24) % 2 == 0) fetch_unit<slot>(R, unit_ptr(WT.w2T, D, 32 * (4 * c + w), 0, lane), NEXT_8_STEPS);
                else fetch_unit<slot>(R, unit_ptr(WT.w1T, FF, 32 * (2 * w), FC * c, lane), next_row_tile(FF));
            } else {
                fetch_unit<slot>(R, unit_ptr(WT.woT, D, 32 * (2 * w + (i - 40)), 0, lane), NEXT_8_STEPS);
            }
        }
        sched_fence();
    };
    // ---- prologue: all of its loads first, then the ring's first two units, then the LDS writes (see k_tail_fwd)
    uint4 in_m[3];
    bf16x8 in_w3[2];
    float4 in_dl = make_float4(0.f, 0.f, 0.f, 0.f);
    float in_dv = 0.f;
    {
#pragma unroll
        for (int i = 0; i < 3; ++i) in_m[i] = reinterpret_cast<const uint4 *>(masks_wg)[tid + i * THREADS];  // 96 * 64 * 2 B = 768 vectors
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int e = tid + i * THREADS;
            in_w3[i] = e < 4 * HID / 8 ? reinterpret_cast<const bf16x8 *>(WT.a3)[e]
                                       : reinterpret_cast<const bf16x8 *>(WT.c3)[e < 5 * HID / 8 ? e - 4 * HID / 8 : 0];
        }
        if (tid < 32 && m0 + tid < M) {
            in_dl = reinterpret_cast<const float4 *>(dlogits)[m0 + tid];
            in_dv = dvalues[m0 + tid];
        }
    }
    sched_fence();
    issue(std::integral_constant<int, 0>{});
    issue(std::integral_constant<int, 1>{});
#pragma unroll
    for (int i = 0; i < 3; ++i) reinterpret_cast<uint4 *>(L.masks)[tid + i * THREADS] = in_m[i];
#pragma unroll
    for (int i = 0; i < 2; ++i)
        if (tid + i * THREADS < 5 * HID / 8) reinterpret_cast<bf16x8 *>(L.w3)[tid + i * THREADS] = in_w3[i];
    if (tid < 32) {
        L.dl[tid][0] = in_dl.x; L.dl[tid][1] = in_dl.y; L.dl[tid][2] = in_dl.z; L.dl[tid][3] = in_dl.w;
        L.dl[tid][4] = in_dv;
    }
    lds_barrier();

    // ---- both heads -> d features (f32, this wave's two tiles)
    f32x16 dfeat[2] = {zero_tile(), zero_tile()};
    auto head = [&](auto u0c, auto nout_c, const __bf16 *w3, int dl_col0, int mt1, int mt2, __bf16 *d3T, __bf16 *d2T, __bf16 *d1T)
                    __attribute__((always_inline)) {
        constexpr int U0 = decltype(u0c)::value, N_OUT = decltype(nout_c)::value;
        // d a2 = (W3^T d out) where a2 > 0 -> xa; d out^T (bf16; rows N_OUT.. of the 32-row buffer stay zero) for the weight gradient
        if (tid < 32 * N_OUT) {
            const int oo = tid / 32, row = tid % 32;
            d3T[packed_off(oo, m0 + row, ld)] = (__bf16)L.dl[row][dl_col0 + oo];
        }
        for (int t = 0; t < 4; ++t) {
            const int mt = 4 * w + t;
            const uint32_t bits = L.masks[(mt2 + mt) * 64 + lane];
            float dlr[N_OUT];
            for (int oo = 0; oo < N_OUT; ++oo) dlr[oo] = (float)(__bf16)L.dl[r][dl_col0 + oo];
            for (int g = 0; g < 4; ++g) {
                const int f0 = 32 * mt + 8 * g + 4 * h;
                float v[4] = {0.f, 0.f, 0.f, 0.f};
                for (int oo = 0; oo < N_OUT; ++oo) {
                    const bf16x4 wv = *reinterpret_cast<const bf16x4 *>(w3 + oo * HID + f0);
                    for (int q = 0; q < 4; ++q) v[q] = __builtin_fmaf((float)wv[q], dlr[oo], v[q]);
                }
                for (int q = 0; q < 4; ++q) v[q] = ((bits >> (4 * g + q)) & 1u) ? v[q] : 0.f;
                put4(L.xa, S512, r, f0, v);
            }
        }
        lds_barrier();
        lds_to_T<HID>(L.xa, S512, d2T, 0, steps_ld, step_m0, tid);
        {   // d a1 = (W2^T d a2) where a1 > 0 -> xb
            static_for<0, 4>([&](auto tc) __attribute__((always_inline)) {
                constexpr int t = decltype(tc)::value, u = U0 + 2 * t;
                const int mt = 4 * w + t;
                bf16x8 af[16];
                issue(std::integral_constant<int, u + DIST>{});
                load_frags<16>(L.xa, S512, 0, af, r, h);
                f32x16 acc = mm16<u % RING>(R, af, zero_tile());
                issue(std::integral_constant<int, u + 1 + DIST>{});
                load_frags<16>(L.xa, S512, 256, af, r, h);
                acc = mm16<(u + 1) % RING>(R, af, acc);
                const uint32_t bits = L.masks[(mt1 + mt) * 64 + lane];
                for (int g = 0; g < 4; ++g) {
                    float v[4];
                    for (int q = 0; q < 4; ++q) v[q] = ((bits >> (4 * g + q)) & 1u) ? acc[4 * g + q] : 0.f;
                    put4(L.xb, S512, r, 32 * mt + 8 * g + 4 * h, v);
                }
            });
        }
        lds_barrier();
        lds_to_T<HID>(L.xb, S512, d1T, 0, steps_ld, step_m0, tid);
        {   // d features += W1^T d a1
            static_for<0, 2>([&](auto jc) __attribute__((always_inline)) {
                constexpr int j = decltype(jc)::value, u = U0 + 8 + 2 * j;
                bf16x8 af[16];
                issue(std::integral_constant<int, u + DIST>{});
                load_frags<16>(L.xb, S512, 0, af, r, h);
                dfeat[j] = mm16<u % RING>(R, af, dfeat[j]);
                issue(std::integral_constant<int, u + 1 + DIST>{});
                load_frags<16>(L.xb, S512, 256, af, r, h);
                dfeat[j] = mm16<(u + 1) % RING>(R, af, dfeat[j]);
            });
        }
        lds_barrier();  // xa / xb are free again (every wave holds its fragments in registers)
    };
    head(std::integral_constant<int, 0>{}, std::integral_constant<int, 4>{}, L.w3, 0, MT_A1, MT_A2, (__bf16 *)G.dlT, (__bf16 *)G.da2T,
         (__bf16 *)G.da1T);
    head(std::integral_constant<int, 12>{}, std::integral_constant<int, 1>{}, L.w3 + 4 * HID, 4, MT_C1, MT_C2, (__bf16 *)G.dvT,
         (__bf16 *)G.dc2T, (__bf16 *)G.dc1T);

    // ---- features = bf16(x_mid + dropout(f)): g = bf16(d features) flows into the residual (-> xm, f32) and, masked, into f (-> xa)
    {
        const Drop d3 = drop.site(3);
        for (int j = 0; j < 2; ++j) {
            const int mt = 2 * w + j;
            for (int g = 0; g < 4; ++g) {
                const int f0 = 32 * mt + 8 * g + 4 * h;
                f32x4 gx;
                float v[4];
                for (int q = 0; q < 4; ++q) v[q] = gx[q] = (float)(__bf16)dfeat[j][4 * g + q];
                d3.apply4(v, (uint64_t)(m0 + r) * D + f0);
                *reinterpret_cast<f32x4 *>(xm + r * XM_S + f0) = gx;
                put4(L.xa, S256, r, f0, v);
            }
        }
    }
    lds_barrier();
    lds_to_T<D>(L.xa, S256, (__bf16 *)G.df2T, 0, steps_ld, step_m0, tid);

    // ---- feed-forward backward, in the forward's chunks: dz = (W2^T df) * [u != 0] / keep;  d h2 += W1^T dz
    {
        bf16x8 xf[16];
        load_frags<16>(L.xa, S256, 0, xf, r, h);
        f32x16 dh[2] = {zero_tile(), zero_tile()};
        static_for<0, FF / FC>([&](auto cc) __attribute__((always_inline)) {
            constexpr int c = decltype(cc)::value, u1 = 24 + 2 * c, u2 = 25 + 2 * c;
            const int ht = 4 * c + w;
            char *ub = L.u[c & 1];
            issue(std::integral_constant<int, u1 + DIST>{});
            const f32x16 du = mm16<u1 % RING>(R, xf, zero_tile());
            const uint32_t bits = L.masks[(MT_FFN + ht) * 64 + lane];
            for (int g = 0; g < 4; ++g) {
                float v[4];
                for (int q = 0; q < 4; ++q) v[q] = ((bits >> (4 * g + q)) & 1u) ? du[4 * g + q] * drop.inv_keep : 0.f;
                put4(ub, S128, r, 32 * w + 8 * g + 4 * h, v);
            }
            lds_barrier();
            lds_to_T<FC>(ub, S128, (__bf16 *)G.dzT, FC * c, steps_ld, step_m0, tid);
            bf16x8 uf[8];
            load_frags<8>(ub, S128, 0, uf, r, h);
            issue(std::integral_constant<int, u2 + DIST>{});
            dh[0] = mm8<u2 % RING, 0>(R, uf, dh[0]);
            dh[1] = mm8<u2 % RING, 1>(R, uf, dh[1]);
        });
        // d h2 (bf16, as the unfused path hands it to the LayerNorm backward) -> xb
        for (int j = 0; j < 2; ++j)
            for (int g = 0; g < 4; ++g) {
                float v[4];
                for (int q = 0; q < 4; ++q) v[q] = dh[j][4 * g + q];
                put4(L.xb, S256, r, 32 * (2 * w + j) + 8 * g + 4 * h, v);
            }
    }
    lds_barrier();

    // ---- LayerNorm backward + residual, one wavefront per row (the arithmetic of k_add_ln_bwd):
    //      dx = g + dLN(d h2);  d(out_proj output) = dropout-masked dx (bf16) -> xa;  gamma / beta partials of this workgroup
    {
        const Drop d1 = drop.site(1);
        const float4 gm = reinterpret_cast<const float4 *>(WT.ln_g)[lane];
        const float gg[4] = {gm.x, gm.y, gm.z, gm.w};
        float dg[4] = {0, 0, 0, 0}, db[4] = {0, 0, 0, 0};
        // the eight rows of this wave: saved residual rows and statistics fetched up front (eight independent loads in flight
        // instead of one exposed round trip per row)
        float4 vrow[8];
        float mrow[8], rrow[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int row = 8 * w + i;
            const int64_t m = m0 + row < M ? m0 + row : M - 1;  // clamped: loads stay in bounds; rows past M carry zero gradients
            vrow[i] = reinterpret_cast<const float4 *>(S.x_mid + m * D)[lane];
            mrow[i] = S.mean[m];
            rrow[i] = S.rstd[m];
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int row = 8 * w + i;
            const bool ok = m0 + row < M;
            const float4 v = vrow[i];
            const float mean = mrow[i], rstd = rrow[i];
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
            d1.apply4(ov, (uint64_t)(m0 + row) * D + 4 * lane);
            for (int q = 0; q < 4; ++q) ov[q] = ok ? ov[q] : 0.f;
            put4(L.xa, S256, row, 4 * lane, ov);
        }
        for (int q = 0; q < 4; ++q) {
            L.red[w][0][4 * lane + q] = dg[q];
            L.red[w][1][4 * lane + q] = db[q];
        }
    }
    lds_barrier();
    for (int c = tid; c < 2 * D; c += THREADS) {
        const int which = c / D, col = c - which * D;
        G.ln_partial[(int64_t)blockIdx.x * 2 * D + c] = L.red[0][which][col] + L.red[1][which][col] + L.red[2][which][col] + L.red[3][which][col];
    }
    lds_to_T<D>(L.xa, S256, (__bf16 *)G.daoT, 0, steps_ld, step_m0, tid);

    // ---- d o = Wo^T d(out_proj output), bf16 rows for the attention backward
    {
        bf16x8 xf[16];
        load_frags<16>(L.xa, S256, 0, xf, r, h);
        static_for<0, 2>([&](auto jc) __attribute__((always_inline)) {
            constexpr int j = decltype(jc)::value, u = 40 + j;
            const int mt = 2 * w + j;
            const f32x16 acc = mm16<u % RING>(R, xf, zero_tile());
            if (m0 + r < M)
                for (int g = 0; g < 4; ++g) {
                    bf16x4 pk;
                    for (int q = 0; q < 4; ++q) pk[q] = (__bf16)acc[4 * g + q];
                    *reinterpret_cast<bf16x4 *>(d_o + (m0 + r) * D + 32 * mt + 8 * g + 4 * h) = pk;
                }
        });
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// weight gradients from transposed operands: dW[n][k] = sum_m dY^T[n][m] X^T[k][m], bias gradient = row sums of dY^T
// ---------------------------------------------------------------------------------------------------------------------
struct DwTable {
    g2048_dw_job jobs[G2048_DW_MAX_JOBS];
    int first_item[G2048_DW_MAX_JOBS + 1];  // prefix sums of the items of one slice: N-blocks x (K / 64 + has_bias)
    int n_jobs, slices;
    int64_t ld, m_per_slice;
};

// One wavefront per (slice of the row axis, job, block of <= 64 x 64 outputs): operands are fragment-packed, so every operand
// load is one contiguous KB, and a 64 x 64 block needs 2 + 2 fragments per 4 MFMAs (a 32 x 32 tile per wave read 2 per MFMA:
// 360 MB out of L2 for 5.6 GFLOP).  Slice = blockIdx % slices: workgroups are dealt round-robin over the 8 XCDs, so with 8
// slices every XCD's L2 only ever sees its own eighth of the row axis of all operands (30 MB in total instead of 8 x 30 MB
// over the fabric).  f32 partial blocks are stored as they are; the slices are summed by g2048_reduce_jobs.
// k-block K / 64 of a job with a bias is the bias block: B = ones, column 0 of the result.
__global__ void __launch_bounds__(THREADS)
k_dweight_t(DwTable T) {
    const int lane = threadIdx.x & 63, r = lane & 31, h = lane >> 5;
    const int slice = blockIdx.x % T.slices;
    const int item = (blockIdx.x / T.slices) * (THREADS / 64) + (threadIdx.x >> 6);
    if (item >= T.first_item[T.n_jobs]) return;
    int j = 0;
    while (item >= T.first_item[j + 1]) ++j;
    const g2048_dw_job J = T.jobs[j];
    const int kb_n = J.K / 64 + (J.db ? 1 : 0);
    const int e = item - T.first_item[j];
    const int nb = e / kb_n, kb = e - nb * kb_n;
    const bool bias = kb == J.K / 64;
    const int nt0 = 2 * nb, nt_cnt = (32 * (nt0 + 1) < J.N) ? 2 : 1;  // (N = 32: one row tile)
    const int64_t steps_total = T.ld >> 4, s0 = (int64_t)slice * (T.m_per_slice >> 4);
    const __bf16 *pa0 = (const __bf16 *)J.dyT + ((int64_t)nt0 * steps_total + s0) * 512 + lane * 8;
    const __bf16 *pa1 = pa0 + (nt_cnt == 2 ? steps_total * 512 : 0);
    const __bf16 *pb0 = bias ? pa0 : (const __bf16 *)J.xT + ((int64_t)(2 * kb) * steps_total + s0) * 512 + lane * 8;
    const __bf16 *pb1 = bias ? pa0 : pb0 + steps_total * 512;
    bf16x8 ones;
    for (int q = 0; q < 8; ++q) ones[q] = (__bf16)1.0f;
    f32x16 c00 = zero_tile(), c01 = zero_tile(), c10 = zero_tile(), c11 = zero_tile();
    const int steps = (int)(T.m_per_slice >> 4), groups = steps / 4;
    struct Frag { bf16x8 a0[4], a1[4], b0[4], b1[4]; };
    auto load = [&](Frag &f, int g) __attribute__((always_inline)) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int64_t o = (int64_t)(4 * g + u) * 512;
            f.a0[u] = *reinterpret_cast<const bf16x8 *>(pa0 + o);
            f.a1[u] = *reinterpret_cast<const bf16x8 *>(pa1 + o);
            f.b0[u] = bias ? ones : *reinterpret_cast<const bf16x8 *>(pb0 + o);
            f.b1[u] = bias ? ones : *reinterpret_cast<const bf16x8 *>(pb1 + o);
        }
    };
    auto mul = [&](const Frag &f) __attribute__((always_inline)) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            c00 = mfma(f.a0[u], f.b0[u], c00);
            c01 = mfma(f.a0[u], f.b1[u], c01);
            c10 = mfma(f.a1[u], f.b0[u], c10);
            c11 = mfma(f.a1[u], f.b1[u], c11);
        }
    };
    Frag f0, f1;
    if (groups > 0) load(f0, 0);
    for (int g = 0; g < groups; g += 2) {
        if (g + 1 < groups) load(f1, g + 1);
        mul(f0);
        if (g + 2 < groups) load(f0, g + 2);
        if (g + 1 < groups) mul(f1);
    }
    for (int st = 4 * groups; st < steps; ++st) {
        const int64_t o = (int64_t)st * 512;
        const bf16x8 a0 = *reinterpret_cast<const bf16x8 *>(pa0 + o), a1 = *reinterpret_cast<const bf16x8 *>(pa1 + o);
        const bf16x8 b0 = bias ? ones : *reinterpret_cast<const bf16x8 *>(pb0 + o), b1 = bias ? ones : *reinterpret_cast<const bf16x8 *>(pb1 + o);
        c00 = mfma(a0, b0, c00);
        c01 = mfma(a0, b1, c01);
        c10 = mfma(a1, b0, c10);
        c11 = mfma(a1, b1, c11);
    }
    if (bias) {
        if (r == 0)
            for (int i = 0; i < 16; ++i) {
                J.db[(int64_t)slice * J.N + 32 * nt0 + rowof(i, h)] = c00[i];
                if (nt_cnt == 2) J.db[(int64_t)slice * J.N + 32 * (nt0 + 1) + rowof(i, h)] = c10[i];
            }
    } else {
        float *out = J.dw + ((int64_t)slice * J.N + 32 * nt0) * J.K + 64 * kb + r;
        for (int i = 0; i < 16; ++i) {
            float *o = out + (int64_t)rowof(i, h) * J.K;
            o[0] = c00[i];
            o[32] = c01[i];
            if (nt_cnt == 2) {
                o[(int64_t)32 * J.K] = c10[i];
                o[(int64_t)32 * J.K + 32] = c11[i];
            }
        }
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
    hipLaunchKernelGGL(k_tail_fwd, dim3((unsigned)blocks, 2), dim3(THREADS), sizeof(TailLds), (hipStream_t)stream, (const __bf16 *)o, x_cls,
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
    if (!jobs || n_jobs <= 0 || n_jobs > G2048_DW_MAX_JOBS || slices <= 0 || m <= 0 || m > ld || (ld & 15) || m % (16 * (int64_t)slices))
        return G2048_EINVAL;
    DwTable T;
    T.n_jobs = n_jobs;
    T.slices = slices;
    T.ld = ld;
    T.m_per_slice = m / slices;
    int items = 0;
    for (int j = 0; j < n_jobs; ++j) {
        const g2048_dw_job &J = jobs[j];
        if (mis16(J.dyT) || mis16(J.xT) || mis16(J.dw) || J.N <= 0 || J.K <= 0 || (J.N & 31) || (J.K & 63) || ((uintptr_t)J.db & 3))
            return G2048_EINVAL;
        T.jobs[j] = J;
        T.first_item[j] = items;
        items += ((J.N + 63) / 64) * (J.K / 64 + (J.db ? 1 : 0));
    }
    for (int j = n_jobs; j <= G2048_DW_MAX_JOBS; ++j) T.first_item[j] = items;
    const int per_block = THREADS / 64;
    hipLaunchKernelGGL(k_dweight_t, dim3((unsigned)(((items + per_block - 1) / per_block) * slices)), dim3(THREADS), 0, (hipStream_t)stream,
                       T);
    return done();
}
