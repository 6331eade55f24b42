// Linear with a 256-wide output FUSED with the row-wise kernel behind it, for the PPO update's minibatch-sized activations on gfx950:
//
//   forward   x_new = x + dropout(u W^T + b);  h = bf16(LayerNorm(x_new))          (out_proj / linear2 + g2048_add_ln_fwd)
//   backward  g_h = dy Wt^T;  dx = g_x + dLayerNorm(g_h);  da = dropout'(dx)        (the input-gradient GEMM of linear1 / in_proj +
//                                                                                   g2048_add_ln_bwd of the LayerNorm in front of it)
//
// Reference: the closing Linear of each sub-layer of nn.TransformerEncoderLayer(norm_first=True) (src/ppo/transformer_encoder.py:
// 138-148) followed by `x = x + dropout(.)` and the next sub-layer's LayerNorm, and their autograd.  Unfused, the [T][256] bf16 tensor
// between the GEMM and the row kernel is written and read back (2 x 17.8 MB at 34 816 tokens, four sites per encoder layer), and the
// three K > 256 products (linear2 forward, the input gradients of linear1 and in_proj) ran on hipBLASLt's 256 x 160 tiles at 3.0-3.3 TB/s
// of their operand bytes.  N = 256 means a whole output row fits one workgroup, so the row kernel can run on the GEMM's tile in LDS.
//
// Decomposition (one workgroup of 8 waves per CU):
//   * a workgroup owns `tpw` <= 160 consecutive tokens (136 = 8 boards at minibatch 2048: 256 workgroups, one round) and ALL 256 output
//     features; wave w owns features 32 w .. 32 w + 31 for every token: up to five 32 x 32 accumulator tiles (80 registers);
//   * computed transposed like every GEMM of this library, Y^T[n][token] = W[n][k] X^T[k][token] with v_mfma_f32_32x32x16_bf16: the
//     weight is the A operand, read from its FRAGMENT-PACKED bf16 shadow (include/g2048.h: one contiguous KB per wave-instruction,
//     L2-resident: 128-512 KB shared by all workgroups) straight into registers through a ring of four fragments (fragment g + 4 is
//     requested right behind the MFMAs that last read fragment g);
//   * X streams through LDS in K-chunks of 128: [160 tokens][256 B], 16-byte pieces XOR-swizzled by row so that a B fragment is one
//     conflict-free ds_read_b128; chunk c is multiplied from one LDS stage while chunk c + 1 is written to the other and chunks c + 2,
//     c + 3 are in flight in registers (global_load -> ds_write, all of it visible to the compiler's wait-count pass: no LDS-DMA here,
//     because the weight fragments are register loads in the same in-order queue and an inline-assembly DMA beside them would make every
//     compiler-generated wait a vmcnt(0)).  The K-loop is a template over K / 128 and fully unrolled - every prefetch decision is a
//     compile-time one (a run-time `if` around a load or an MFMA costs a vmcnt(0) in front of every MFMA) - with the chunk's addresses
//     derived from a run-time counter, so that the unrolled code does not keep 40 precomputed addresses in registers;
//   * one barrier per chunk (LDS-only: __syncthreads() would drain the prefetch);
//   * epilogue: the output tile goes to LDS as bf16 rows (the rounding of the unfused Linear's output), then every wave walks rows of
//     it exactly as g2048_add_ln_fwd / _bwd do (one wavefront per token row, 4 features per lane, statistics by wave reduction, the
//     same dropout hash on the same element index), with the residual rows' loads issued ten rows ahead.
// Bound: HBM.  Per token the forward reads 2 K (u) + 1024 (x) and writes 1024 (x_new) + 512 (h) bytes; the MFMA work of a chunk
// (80 MFMAs per SIMD = 2 560 cycles) is about what the chunk's 35 KB take to arrive at a CU's share of 6 TB/s.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

#include <type_traits>

#include "../../include/g2048.h"
#include "g2048_mfma.h"

namespace {

using namespace g2048_mfma;

constexpr int RG_N = 256, RG_KC = 128, RG_THREADS = 512, RG_WAVES = 8;
// NB = 32-token MFMA blocks per tile (3: tiles of <= 96 tokens, 5: <= 160).  Per NB: one K-chunk of the X tile in LDS (NB x 8 KB), 16-byte
// pieces per thread and chunk (NB), rows per wave in the row pass (4 NB), rows of a wave whose residual loads are in flight together
template <int NB> struct RgShape {
    static constexpr int TT = 32 * NB, STAGE = TT * RG_KC * 2, PIECES = TT * (RG_KC / 8) / RG_THREADS, RPW = TT / RG_WAVES, BATCH = RPW / 2;
    static constexpr int LDS = 2 * STAGE + RG_WAVES * 3 * RG_N * 4 + RG_N * 4;  // X stages / output tile, backward partials, bias
    static_assert(PIECES * RG_THREADS == TT * (RG_KC / 8) && RPW % 2 == 0 && BATCH >= 1, "tile shape");
};

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
enum { RG_FWD = 0, RG_BWD = 1 };

struct RowGemmArgs {
    const __bf16 *x;  // [T][K] GEMM input (u forward, dy backward), leading dimension ldx
    int64_t ldx;
    const __bf16 *w;  // fragment-packed [256][K] (or K columns of a wider packed matrix: w_tile_stride)
    int64_t w_tile_stride;  // elements between the 32-row tiles of the packed weight: (K / 16) * 512 for a dense [256][K]
    const uint16_t *gh_extra;  // backward: optional bf16 [T / extra_period][256] added to g_h on the rows tok % extra_period == 0
    int extra_period;
    const float *bias;  // forward: the Linear's bias (f32 [256]) or null
    int64_t T;
    int K, tpw;
    // row pass
    const float *res;   // forward: x (residual stream), backward: x_norm (the tensor that was normalised); row stride res_rs
    int64_t res_rs;
    const float *g_x;   // backward: gradient of the residual stream (may be null), row r belongs to token r * g_x_period
    int g_x_period;
    const float *gamma, *beta;
    float *mean, *rstd;  // forward: written; backward: read
    float *out_f32;      // forward: x_new; backward: dx
    uint16_t *out_bf16;  // forward: h; backward: da (may be null)
    float *partial;      // backward: [gridDim.x][3][256]
    float eps, inv_keep;
    uint32_t thr, s0, s1;
    const uint64_t *seed_state;
};

__device__ __forceinline__ float bf2f(uint32_t hi16) { return __uint_as_float(hi16 << 16); }
__device__ __forceinline__ uint32_t f2bf(float f) {
    const __bf16 b = (__bf16)f;
    return *reinterpret_cast<const uint16_t *>(&b);
}

// Sum over the 64 lanes, the same value in every lane: four DPP adds inside the 16-lane rows (quad permutes, half-row mirror, row
// mirror), then the four row sums through scalar registers.  `__shfl_xor` compiles to ds_bpermute_b32: six LDS-crossbar round trips per
// reduction (~100 cycles of latency each) where this is ~12 short vector instructions.
template <int CTRL>
__device__ __forceinline__ float dpp_step(float v) {
    return v + __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, true));
}
__device__ __forceinline__ float wave_sum_dpp(float v) {
    v = dpp_step<0xB1>(v);   // quad_perm [1, 0, 3, 2]
    v = dpp_step<0x4E>(v);   // quad_perm [2, 3, 0, 1]
    v = dpp_step<0x141>(v);  // row_half_mirror
    v = dpp_step<0x140>(v);  // row_mirror: every lane of a 16-lane row holds the row's sum
    const int iv = __builtin_bit_cast(int, v);
    const float r0 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(iv, 0)), r1 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(iv, 16));
    const float r2 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(iv, 32)), r3 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(iv, 48));
    return (r0 + r1) + (r2 + r3);
}

template <int MODE, int NB, int NCH>  // NCH = K / 128: the K-loop is unrolled over its chunks (every prefetch decision at compile time)
__global__ void __launch_bounds__(RG_THREADS, 2)
k_rowgemm(RowGemmArgs A) {
    constexpr int RG_NB = NB, RG_TT = RgShape<NB>::TT, RG_STAGE = RgShape<NB>::STAGE, RG_PIECES = RgShape<NB>::PIECES;
    constexpr int RG_RPW = RgShape<NB>::RPW, RG_BATCH = RgShape<NB>::BATCH;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, r = lane & 31, h = lane >> 5;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    float *const red = reinterpret_cast<float *>(smem + 2 * RG_STAGE);             // [8][3][256] (backward)
    float *const bias_l = reinterpret_cast<float *>(smem + 2 * RG_STAGE + RG_WAVES * 3 * RG_N * 4);
    uint32_t s0 = A.s0, s1 = A.s1;
    if (A.seed_state) {  // the same mixing as g2048_layernorm.hip
        const uint64_t s = *A.seed_state;
        s0 ^= (uint32_t)s * 0x9E3779B1u;
        s1 += (uint32_t)(s >> 32) * 0x85EBCA77u + (uint32_t)s;
    }
    const int64_t n_tiles = (A.T + A.tpw - 1) / A.tpw;
    if (tid < RG_N) bias_l[tid] = (MODE == RG_FWD && A.bias) ? A.bias[tid] : 0.f;

    for (int64_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
        const int64_t tok0 = tile * A.tpw;
        const int n_valid = (int)((A.T - tok0 < A.tpw) ? A.T - tok0 : A.tpw);  // tokens of this tile
        // ---- X pieces of this thread: piece e = tid + 512 j -> tile row (tid >> 4) + 32 j, 16-byte piece tid & 15 (rows past the tile
        // read its last row); in the stage the piece sits at pc ^ (row & 15), and row & 15 does not depend on j
        uint32_t goff[RG_PIECES];
        const int row0 = tid >> 4, pc = tid & 15;
#pragma unroll
        for (int j = 0; j < RG_PIECES; ++j) {
            const int row = row0 + 32 * j;
            goff[j] = (uint32_t)((int64_t)(row < n_valid ? row : n_valid - 1) * A.ldx * 2) + 16u * (uint32_t)pc;
        }
        const uint32_t loff0 = (uint32_t)(row0 * (RG_KC * 2) + ((pc ^ (row0 & 15)) << 4));
        const char *const xg = reinterpret_cast<const char *>(A.x + tok0 * A.ldx);
        // chunk c of the X tile travels global -> G[c % 3] -> LDS stage c & 1: THREE chunks ahead in registers, two stages in LDS.  The
        // K-loop is bound by the latency of its HBM reads (2.3 us per chunk = 15 GB/s per CU with two chunks in flight, round 4's first
        // form): the bytes in flight are what buys bandwidth, and the registers for the third chunk come from a four-fragment weight
        // ring (instead of eight) and three B fragments ahead (instead of five).  (Indexed with compile-time constants only, and a
        // NATIVE vector type: HIP's uint4 struct is copied by memcpy across address spaces, which kept the array in scratch memory - and
        // a scratch store of a register that a global load is still filling waits for the load right behind its issue.)
        u32x4 G[3][RG_PIECES];
        // (kc: the current chunk, advanced at run time and hidden from constant folding - with
        // every chunk's addresses known at compile time the compiler computes all 40 of them up front and keeps them in registers)
        unsigned kc = 0;  // the chunk the K loop is at
#define RG_GLOAD(P, ahead)                                                                                                \
    _Pragma("unroll") for (int j = 0; j < RG_PIECES; ++j) G[P][j] =                                                       \
        *reinterpret_cast<const u32x4 *>(xg + goff[j] + (size_t)(kc + (ahead)) * (RG_KC * 2))
#define RG_LSTORE(P, ST)                                                                                                  \
    _Pragma("unroll") for (int j = 0; j < RG_PIECES; ++j)                                                                 \
        *reinterpret_cast<u32x4 *>(smem + (ST) * RG_STAGE + loff0 + j * (32 * RG_KC * 2)) = G[P][j]
        // ---- weight fragments of this wave: row tile w, k-step g = 8 c + ks at (w tile stride + g 512 + 8 lane) elements.  A ring of
        // WR fragments: fragment g + WR is fetched right behind the MFMAs that were the last to read fragment g (WR k-steps = ~0.6 us
        // ahead of its first use: an L2 hit)
        constexpr int WR = 4, NG = 8 * NCH;
        const __bf16 *const wp = A.w + (size_t)w * A.w_tile_stride + lane * 8;
        bf16x8 W[WR];
        // B fragment of k-step ks, block b: row 32 b + r of the stage, piece (2 ks + h) ^ (r & 15) = 2 ks ^ (h ^ (r & 15))
        const uint32_t brow = (uint32_t)(r * (RG_KC * 2)), bx = (uint32_t)((h ^ (r & 15)) << 4);

        f32x16 acc[RG_NB];
        RG_GLOAD(0, 0);
        if constexpr (NCH > 1) { RG_GLOAD(1, 1); }
        if constexpr (NCH > 2) { RG_GLOAD(2, 2); }
#pragma unroll
        for (int g = 0; g < WR; ++g) W[g] = *reinterpret_cast<const bf16x8 *>(wp + (size_t)g * 512);
        lds_barrier();  // every wave is done with the previous tile's LDS (and bias_l is visible)
#pragma unroll
        for (int b = 0; b < RG_NB; ++b) acc[b] = bias_tile(bias_l, 32 * w, h);  // the bias enters through the accumulators (backward: zeros)
        RG_LSTORE(0, 0);
        lds_barrier();

        // ---- epilogue state, declared in front of the K-loop (its first loads are issued inside the last chunk): wave w takes rows w, w + 8, ... of the tile; lane l holds features
        // 4 l .. 4 l + 3 of a row.  The residual rows of the first batch are requested under the last chunk's MFMAs (the registers of
        // the X prefetch are free by then), the rest right after the output tile is staged.  Rows past the tile are computed on the
        // tile's last row and not stored - no control flow inside a batch.
        const bool norm = A.gamma != nullptr;
        const float4 gm = norm ? reinterpret_cast<const float4 *>(A.gamma)[lane] : make_float4(0.f, 0.f, 0.f, 0.f);
        auto row_tok = [&](int row) -> int64_t { return tok0 + (row < n_valid ? row : n_valid - 1); };
        // PIPE (the 96-token shape): loads of batch k + 1 in flight under the arithmetic of batch k, the first batch requested under the
        // last K-chunk.  The 160-token shape has no registers for that next to its five accumulator tiles (every pipelined form spilled
        // 40-280 registers): it loads a batch, works on it, loads the next.
        constexpr bool PIPE = NB <= 3;
        constexpr int BB = RG_BATCH / 2, NBB = RG_RPW / BB;  // backward: rows per batch (two 16-byte loads per row), batches
        static_assert(RG_RPW % BB == 0, "backward batches");
        float4 fv[MODE == RG_FWD ? RG_RPW : 1];
        float4 bx0[2][MODE == RG_BWD ? BB : 1], bg0[2][MODE == RG_BWD ? BB : 1];
        float bmu[2][MODE == RG_BWD ? BB : 1], brs[2][MODE == RG_BWD ? BB : 1];
        auto bwd_load = [&](int k, auto parity) {
            constexpr int P = decltype(parity)::value;
#pragma unroll
            for (int j = 0; j < BB; ++j) {
                const int64_t tok = row_tok(w + RG_WAVES * (k * BB + j));
                bx0[P][j] = reinterpret_cast<const float4 *>(A.res + tok * A.res_rs)[lane];
                bmu[P][j] = A.mean[tok];
                brs[P][j] = A.rstd[tok];
                bg0[P][j] = make_float4(0.f, 0.f, 0.f, 0.f);
                if (A.g_x) {
                    if (A.g_x_period == 1) bg0[P][j] = reinterpret_cast<const float4 *>(A.g_x + tok * RG_N)[lane];
                    else if (tok % A.g_x_period == 0) bg0[P][j] = reinterpret_cast<const float4 *>(A.g_x + (tok / A.g_x_period) * RG_N)[lane];
                }
            }
        };
        auto fwd_load = [&](auto lo_c, auto hi_c) {
#pragma unroll
            for (int j = decltype(lo_c)::value; j < decltype(hi_c)::value; ++j)
                fv[j] = reinterpret_cast<const float4 *>(A.res + row_tok(w + RG_WAVES * j) * A.res_rs)[lane];
        };
        using I0 = std::integral_constant<int, 0>;
        using IB = std::integral_constant<int, RG_BATCH>;
        using IR = std::integral_constant<int, RG_RPW>;
        // one K-chunk c (a compile-time constant): before its MFMAs the loads of chunk c + 3 (into the registers chunk c came through),
        // after every k-step's MFMAs the weight fragment WR k-steps ahead; after all of them chunk c + 1 goes from its registers into
        // the other stage.  Everything conditional is `if constexpr`, and the MFMAs are unconditional (a partial tile multiplies its
        // clamped rows too): with run-time conditions around the loads or the MFMAs the compiler's wait-count pass loses track at every
        // join and puts `s_waitcnt vmcnt(0)` in front of every MFMA, i.e. the whole prefetch waits for memory 40 times per chunk (first
        // version of this kernel: 4.3 us per chunk).
#define RG_FRAG(st, s_) \
    *reinterpret_cast<const bf16x8 *>((st) + brow + ((32u * ((s_) / RG_NB)) ^ bx) + ((s_) % RG_NB) * (32 * RG_KC * 2))
#define RG_CHUNK(c)                                                                                                        \
    if constexpr ((c) < NCH) {                                                                                             \
        if constexpr ((c) == NCH - 1 && PIPE) { /* the first batch of residual rows, under the last chunk's MFMAs */       \
            if (MODE == RG_FWD) fwd_load(I0{}, IB{});                                                                      \
            else bwd_load(0, I0{});                                                                                        \
        }                                                                                                                  \
        if constexpr ((c) + 3 < NCH) { RG_GLOAD((c) % 3, 3); }                                                             \
        const char *const st = smem + ((c) & 1) * RG_STAGE;                                                                \
        constexpr int PRE = 3, NS = 8 * RG_NB; /* step s: k-step s / NB, block s % NB; B fragments three MFMAs ahead */    \
        bf16x8 q[PRE];                                                                                                     \
        _Pragma("unroll") for (int s_ = 0; s_ < PRE; ++s_) q[s_] = RG_FRAG(st, s_);                                        \
        _Pragma("unroll") for (int s_ = 0; s_ < NS; ++s_) {                                                                \
            __builtin_amdgcn_sched_barrier(0);                                                                             \
            const int ks = s_ / RG_NB, b = s_ % RG_NB, g = 8 * (c) + ks;                                                   \
            acc[b] = mfma(W[g % WR], q[s_ % PRE], acc[b]);                                                                 \
            if (s_ + PRE < NS) q[s_ % PRE] = RG_FRAG(st, s_ + PRE);                                                        \
            if (b == RG_NB - 1 && g + WR < NG) W[g % WR] = *reinterpret_cast<const bf16x8 *>(wp + (size_t)(kc * 8 + ks + WR) * 512); \
        }                                                                                                                  \
        __builtin_amdgcn_sched_barrier(0);                                                                                 \
        if constexpr ((c) + 1 < NCH) { RG_LSTORE(((c) + 1) % 3, ((c) + 1) & 1); }                                          \
        kc += 1;                                                                                                           \
        asm volatile("" : "+s"(kc));                                                                                       \
        lds_barrier();                                                                                                     \
    }
        RG_CHUNK(0) RG_CHUNK(1) RG_CHUNK(2) RG_CHUNK(3) RG_CHUNK(4) RG_CHUNK(5) RG_CHUNK(6) RG_CHUNK(7)
#undef RG_CHUNK
#undef RG_FRAG
#undef RG_GLOAD
#undef RG_LSTORE

        // output tile -> LDS as bf16 rows of 512 bytes, 16-byte chunk ch of row t at ch ^ (t & 31) (every wave is past the last chunk's
        // barrier: both stages are free)
#pragma unroll
        for (int b = 0; b < RG_NB; ++b) {
            const int trow = 32 * b + r;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const uint32_t lo = f2bf(acc[b][4 * g + 0]) | (f2bf(acc[b][4 * g + 1]) << 16);
                const uint32_t hi = f2bf(acc[b][4 * g + 2]) | (f2bf(acc[b][4 * g + 3]) << 16);
                *reinterpret_cast<uint2 *>(smem + trow * 512 + (((4 * w + g) ^ (trow & 31)) << 4) + 8 * h) = make_uint2(lo, hi);
            }
        }
        lds_barrier();
        if (MODE == RG_FWD) {
            if (PIPE) fwd_load(IB{}, IR{});
            else fwd_load(I0{}, IR{});
        }
        auto tile_row = [&](int row) -> uint2 {
            return *reinterpret_cast<const uint2 *>(smem + row * 512 + ((((lane >> 1) ^ (row & 31))) << 4) + 8 * (lane & 1));
        };
        // Rows are taken in batches, the arithmetic of a batch's rows side by side and their wave reductions INTERLEAVED (independent
        // chains: a single row's two dependent reductions are hundreds of cycles of pure latency, and with eight waves per CU nothing
        // else hides them).
        if (MODE == RG_FWD) {
            const float4 bt = norm ? reinterpret_cast<const float4 *>(A.beta)[lane] : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
            for (int j0 = 0; j0 < RG_RPW; j0 += RG_BATCH) {
                if (w + RG_WAVES * j0 >= n_valid) break;  // (uniform)
                float4 v[RG_BATCH];
                float s[RG_BATCH];
#pragma unroll
                for (int j = 0; j < RG_BATCH; ++j) {
                    const int row = w + RG_WAVES * (j0 + j), rowc = row < n_valid ? row : n_valid - 1;
                    const int64_t tok = tok0 + rowc;
                    const uint2 ab = tile_row(rowc);
                    float av[4] = {bf2f(ab.x & 0xFFFFu), bf2f(ab.x >> 16), bf2f(ab.y & 0xFFFFu), bf2f(ab.y >> 16)};
                    if (A.thr) {
                        const uint64_t base = (uint64_t)tok * RG_N + 4 * lane;
#pragma unroll
                        for (int q = 0; q < 4; ++q) av[q] = keep_elem(s0, s1, A.thr, base + q) ? av[q] * A.inv_keep : 0.0f;
                    }
                    v[j] = fv[j0 + j];
                    v[j].x += av[0]; v[j].y += av[1]; v[j].z += av[2]; v[j].w += av[3];
                    if (row < n_valid) reinterpret_cast<float4 *>(A.out_f32 + tok * RG_N)[lane] = v[j];
                    s[j] = v[j].x + v[j].y + v[j].z + v[j].w;
                }
#pragma unroll
                for (int j = 0; j < RG_BATCH; ++j) s[j] = wave_sum_dpp(s[j]);  // (independent chains: the scheduler interleaves them)
                float mean[RG_BATCH];
#pragma unroll
                for (int j = 0; j < RG_BATCH; ++j) {
                    mean[j] = s[j] * (1.0f / RG_N);
                    v[j].x -= mean[j]; v[j].y -= mean[j]; v[j].z -= mean[j]; v[j].w -= mean[j];
                    s[j] = v[j].x * v[j].x + v[j].y * v[j].y + v[j].z * v[j].z + v[j].w * v[j].w;
                }
#pragma unroll
                for (int j = 0; j < RG_BATCH; ++j) s[j] = wave_sum_dpp(s[j]);
#pragma unroll
                for (int j = 0; j < RG_BATCH; ++j) {
                    const int row = w + RG_WAVES * (j0 + j);
                    const int64_t tok = tok0 + row;
                    const float rstd = rsqrtf(s[j] * (1.0f / RG_N) + A.eps);
                    const uint32_t lo = f2bf(v[j].x * rstd * gm.x + bt.x) | (f2bf(v[j].y * rstd * gm.y + bt.y) << 16);
                    const uint32_t hi = f2bf(v[j].z * rstd * gm.z + bt.z) | (f2bf(v[j].w * rstd * gm.w + bt.w) << 16);
                    if (row < n_valid) {
                        reinterpret_cast<uint2 *>(A.out_bf16 + tok * RG_N)[lane] = make_uint2(lo, hi);
                        if (lane == 0) {
                            A.mean[tok] = mean[j];
                            A.rstd[tok] = rstd;
                        }
                    }
                }
            }
        } else {
            float dg[4] = {0, 0, 0, 0}, db[4] = {0, 0, 0, 0}, dsum[4] = {0, 0, 0, 0};
            const float gg[4] = {gm.x, gm.y, gm.z, gm.w};
            // one batch (parity P of the load registers): the NEXT batch's loads are issued before this one's arithmetic
            auto bwd_batch = [&](int k, auto parity) {
                constexpr int P = decltype(parity)::value;
                if (PIPE && k + 1 < NBB && w + RG_WAVES * (k + 1) * BB < n_valid) bwd_load(k + 1, std::integral_constant<int, P ^ 1>{});
                float xh[BB][4], dxh[BB][4], c1[BB], c2[BB];
#pragma unroll
                for (int j = 0; j < BB; ++j) {
                    const int row = w + RG_WAVES * (k * BB + j), rowc = row < n_valid ? row : n_valid - 1;
                    const float ok = row < n_valid ? 1.f : 0.f;  // a clamped duplicate row adds nothing to the column sums
                    const uint2 gb = tile_row(rowc);
                    float gh[4] = {bf2f(gb.x & 0xFFFFu), bf2f(gb.x >> 16), bf2f(gb.y & 0xFFFFu), bf2f(gb.y >> 16)};
                    if (A.gh_extra) {  // (uniform) e.g. the CLS rows' share of the last layer's query projection
                        const int64_t tk = tok0 + rowc;
                        if (tk % A.extra_period == 0) {
                            const uint2 eb = reinterpret_cast<const uint2 *>(A.gh_extra + (tk / A.extra_period) * RG_N)[lane];
                            // (the unfused path adds in bf16: g_h = bf16(g_h + extra))
                            gh[0] = bf2f(f2bf(gh[0] + bf2f(eb.x & 0xFFFFu))); gh[1] = bf2f(f2bf(gh[1] + bf2f(eb.x >> 16)));
                            gh[2] = bf2f(f2bf(gh[2] + bf2f(eb.y & 0xFFFFu))); gh[3] = bf2f(f2bf(gh[3] + bf2f(eb.y >> 16)));
                        }
                    }
                    const float xs[4] = {bx0[P][j].x, bx0[P][j].y, bx0[P][j].z, bx0[P][j].w};
                    c1[j] = c2[j] = 0.f;
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        xh[j][q] = (xs[q] - bmu[P][j]) * brs[P][j];
                        dxh[j][q] = gh[q] * gg[q];
                        c1[j] += dxh[j][q];
                        c2[j] += dxh[j][q] * xh[j][q];
                        dg[q] += ok * (gh[q] * xh[j][q]);
                        db[q] += ok * gh[q];
                    }
                }
#pragma unroll
                for (int j = 0; j < BB; ++j) {
                    c1[j] = wave_sum_dpp(c1[j]);
                    c2[j] = wave_sum_dpp(c2[j]);
                }
#pragma unroll
                for (int j = 0; j < BB; ++j) {
                    const int row = w + RG_WAVES * (k * BB + j);
                    const bool ok = row < n_valid;
                    const int64_t tok = tok0 + (ok ? row : n_valid - 1);
                    float o[4] = {bg0[P][j].x, bg0[P][j].y, bg0[P][j].z, bg0[P][j].w};
                    const float m1 = c1[j] * (1.0f / RG_N), m2 = c2[j] * (1.0f / RG_N);
#pragma unroll
                    for (int q = 0; q < 4; ++q) o[q] += brs[P][j] * (dxh[j][q] - m1 - xh[j][q] * m2);
                    if (ok) reinterpret_cast<float4 *>(A.out_f32 + tok * RG_N)[lane] = make_float4(o[0], o[1], o[2], o[3]);
                    if (A.out_bf16) {
                        if (A.thr) {
                            const uint64_t base = (uint64_t)tok * RG_N + 4 * lane;
#pragma unroll
                            for (int q = 0; q < 4; ++q) o[q] = keep_elem(s0, s1, A.thr, base + q) ? o[q] * A.inv_keep : 0.0f;
                        }
                        uint32_t bq[4];
#pragma unroll
                        for (int q = 0; q < 4; ++q) {
                            bq[q] = f2bf(o[q]);
                            dsum[q] += ok ? bf2f(bq[q]) : 0.f;  // what at::sum over the bf16 tensor would add
                        }
                        if (ok) reinterpret_cast<uint2 *>(A.out_bf16 + tok * RG_N)[lane] = make_uint2(bq[0] | (bq[1] << 16), bq[2] | (bq[3] << 16));
                    }
                }
            };
            if (PIPE) {
#pragma unroll
                for (int k = 0; k < NBB; k += 2) {
                    if (w + RG_WAVES * k * BB < n_valid) bwd_batch(k, std::integral_constant<int, 0>{});
                    if (k + 1 < NBB && w + RG_WAVES * (k + 1) * BB < n_valid) bwd_batch(k + 1, std::integral_constant<int, 1>{});
                }
            } else {
#pragma unroll 1
                for (int k = 0; k < NBB; ++k) {
                    if (w + RG_WAVES * k * BB >= n_valid) break;
                    bwd_load(k, std::integral_constant<int, 0>{});
                    bwd_batch(k, std::integral_constant<int, 0>{});
                }
            }
            // this workgroup's column sums (one partial row per workgroup AND tile: a workgroup that walks several tiles adds them up)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                red[(w * 3 + 0) * RG_N + 4 * lane + q] = dg[q];
                red[(w * 3 + 1) * RG_N + 4 * lane + q] = db[q];
                red[(w * 3 + 2) * RG_N + 4 * lane + q] = dsum[q];
            }
            lds_barrier();
            for (int c = tid; c < 3 * RG_N; c += RG_THREADS) {
                float s = 0.f;
#pragma unroll
                for (int ww = 0; ww < RG_WAVES; ++ww) s += red[ww * 3 * RG_N + c];
                float *dst = A.partial + (int64_t)blockIdx.x * 3 * RG_N + c;
                *dst = (tile == (int64_t)blockIdx.x) ? s : *dst + s;
            }
        }
        // (the next tile's first barrier separates these LDS reads from its writes)
    }
}

inline int rg_done() {
    const hipError_t e = hipGetLastError();
    return e == hipSuccess ? 0 : -(1000 + (int)e);
}

// Tile size: ONE tile per workgroup when T allows it (136 tokens = 8 boards at T = 34 816: 256 workgroups, one round on 256 CUs), because
// every tile streams the whole weight through its CU once - 512 KB from L2 at K = 1024 - and five token blocks per weight fragment
// amortise that better than three.  Measured (linear2 + add + LayerNorm at T = 34 816, cold operands): one 136-token tile per workgroup
// 45.9 us, two 68-token tiles (96-token shape) 55.2 us - the hope that the second tile's reads would overlap the first tile's stores did
// not survive the doubled weight traffic.  G2048_RG_TWO_TILES=1 selects the two-tile policy (the A/B switch of that measurement).
inline int rg_tpw(int64_t T, int *nb) {
    const char *e = getenv("G2048_RG_TWO_TILES");  // (read per call: the library keeps no latched state)
    const bool two = e && e[0] == '1';
    int64_t t = (T + 255) / 256;
    if (two) t = (T + 511) / 512;
    if (t <= 96 && (two || t <= 48)) {  // small T: the 96-token shape wastes fewer MFMA rows
        *nb = 3;
        return (int)(t < 32 ? 32 : t);
    }
    *nb = 5;
    return (int)(t > 160 ? 160 : t);
}

template <int MODE, int NB, int NCH>
int rg_launch3(RowGemmArgs &A, unsigned grid, hipStream_t stream) {
    const void *fn = reinterpret_cast<const void *>(k_rowgemm<MODE, NB, NCH>);
    constexpr int lds = RgShape<NB>::LDS;
    if (hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, lds) != hipSuccess) return -(1000 + (int)hipGetLastError());
    hipLaunchKernelGGL((k_rowgemm<MODE, NB, NCH>), dim3(grid), dim3(RG_THREADS), lds, stream, A);
    return rg_done();
}
template <int MODE, int NB>
int rg_launch2(RowGemmArgs &A, unsigned grid, hipStream_t stream) {
    switch (A.K / RG_KC) {  // the K-loop is unrolled over its chunks: K = 256, 512, 768, 1024
        case 2: return rg_launch3<MODE, NB, 2>(A, grid, stream);
        case 4: return rg_launch3<MODE, NB, 4>(A, grid, stream);
        case 6: return rg_launch3<MODE, NB, 6>(A, grid, stream);
        case 8: return rg_launch3<MODE, NB, 8>(A, grid, stream);
        default: return G2048_EINVAL;
    }
}
template <int MODE>
int rg_launch(RowGemmArgs &A, hipStream_t stream) {
    int nb = 5;
    A.tpw = rg_tpw(A.T, &nb);
    const int64_t tiles = (A.T + A.tpw - 1) / A.tpw;
    const unsigned grid = (unsigned)(tiles < 256 ? tiles : 256);
    return nb == 3 ? rg_launch2<MODE, 3>(A, grid, stream) : rg_launch2<MODE, 5>(A, grid, stream);
}

inline bool rg_gemm_ok(const void *x, int64_t ldx, const void *w, int64_t T, int K) {
    return x && w && T > 0 && K >= 256 && K % 256 == 0 && K <= 1024 && ldx >= K && !(ldx & 7) && !(((uintptr_t)x | (uintptr_t)w) & 15) &&
           (int64_t)160 * ldx * 2 < (1ll << 31);
}

}  // namespace

extern "C" int64_t g2048_linear_add_ln_bwd_partial_rows(int64_t T) {
    if (T <= 0) return 0;
    int nb;
    const int tpw = rg_tpw(T, &nb);
    const int64_t tiles = (T + tpw - 1) / tpw;
    return tiles < 256 ? tiles : 256;
}

extern "C" int g2048_linear_add_ln_fwd(const void *u, int64_t ldu, const void *w_packed, const float *bias, int K, const float *x,
                                       int64_t x_row_stride, const float *gamma, const float *beta, float *x_new, void *h, float *mean,
                                       float *rstd, int64_t T, float eps, float p_drop, uint64_t seed, const uint64_t *seed_state,
                                       void *stream) {
    if (!rg_gemm_ok(u, ldu, w_packed, T, K) || !x || !gamma || !beta || !x_new || !h || !mean || !rstd || !(p_drop >= 0.f && p_drop < 1.f) ||
        (x_row_stride & 3) || (((uintptr_t)x | (uintptr_t)gamma | (uintptr_t)beta | (uintptr_t)x_new | (uintptr_t)bias) & 15) ||
        ((uintptr_t)h & 7))
        return G2048_EINVAL;
    RowGemmArgs A{};
    A.x = (const __bf16 *)u; A.ldx = ldu; A.w = (const __bf16 *)w_packed; A.bias = bias; A.T = T; A.K = K;
    A.w_tile_stride = (int64_t)(K >> 4) * 512; A.extra_period = 1;
    A.res = x; A.res_rs = x_row_stride; A.gamma = gamma; A.beta = beta; A.mean = mean; A.rstd = rstd; A.out_f32 = x_new;
    A.out_bf16 = (uint16_t *)h; A.eps = eps; A.inv_keep = 1.0f / (1.0f - p_drop); A.thr = (uint32_t)(p_drop * 16777216.0f);
    A.s0 = (uint32_t)seed; A.s1 = (uint32_t)(seed >> 32); A.seed_state = seed_state; A.g_x_period = 1;
    return rg_launch<RG_FWD>(A, (hipStream_t)stream);
}

extern "C" int g2048_linear_add_ln_bwd(const void *dy, int64_t lddy, const void *wt_packed, int64_t wt_tile_stride, int K,
                                       const float *x_norm, int64_t x_row_stride, const float *g_x, int g_x_period, const void *g_h_extra,
                                       int extra_period, const float *mean, const float *rstd, const float *gamma, float *dx, void *da,
                                       float *partial, int64_t T, float p_drop, uint64_t seed, const uint64_t *seed_state, void *stream) {
    if (wt_tile_stride == 0) wt_tile_stride = (int64_t)(K >> 4) * 512;
    if (wt_tile_stride < (int64_t)(K >> 4) * 512 || (wt_tile_stride & 7) || (g_h_extra && extra_period < 1) || ((uintptr_t)g_h_extra & 7))
        return G2048_EINVAL;
    if (!rg_gemm_ok(dy, lddy, wt_packed, T, K) || !x_norm || !mean || !rstd || !gamma || !dx || !partial || g_x_period < 1 ||
        !(p_drop >= 0.f && p_drop < 1.f) || (x_row_stride & 3) ||
        (((uintptr_t)x_norm | (uintptr_t)g_x | (uintptr_t)gamma | (uintptr_t)dx) & 15) || ((uintptr_t)da & 7) || ((uintptr_t)partial & 3))
        return G2048_EINVAL;
    RowGemmArgs A{};
    A.x = (const __bf16 *)dy; A.ldx = lddy; A.w = (const __bf16 *)wt_packed; A.T = T; A.K = K;
    A.w_tile_stride = wt_tile_stride; A.gh_extra = (const uint16_t *)g_h_extra; A.extra_period = extra_period < 1 ? 1 : extra_period;
    A.res = x_norm; A.res_rs = x_row_stride; A.g_x = g_x; A.g_x_period = g_x_period; A.gamma = gamma;
    A.mean = const_cast<float *>(mean); A.rstd = const_cast<float *>(rstd); A.out_f32 = dx; A.out_bf16 = (uint16_t *)da;
    A.partial = partial; A.inv_keep = 1.0f / (1.0f - p_drop); A.thr = da ? (uint32_t)(p_drop * 16777216.0f) : 0u;
    A.s0 = (uint32_t)seed; A.s1 = (uint32_t)(seed >> 32); A.seed_state = seed_state;
    return rg_launch<RG_BWD>(A, (hipStream_t)stream);
}
