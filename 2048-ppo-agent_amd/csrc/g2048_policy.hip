// Fused Transformer-encoder forward for policy inference on gfx950 (bf16 MFMA, f32 accumulate).
//
// Replaces, for rollouts, the PyTorch path of PPOAgent.features() (reference src/ppo/ppo_agent.py:103-106,
// src/ppo/transformer_encoder.py:150-190: embedding + 2-D positional code + CLS token + L pre-norm encoder
// layers, "cls" reduction) at the reference's default shape: d_model 256, 8 heads of 32, feed-forward 1024,
// 17 tokens per board.  One workgroup (4 waves, one per SIMD) carries 7 boards = 119 tokens (+9 pad) through ALL
// layers: the f32 residual stream never leaves registers, the only activations that touch LDS are one head's K and
// V^T tiles, weights stream L2 -> LDS by LDS-DMA, and nothing but the 16-byte boards is read from / the 1 KiB CLS
// feature rows written to HBM.
//
// Orientation: every GEMM is computed transposed, Y^T[out_feature][token] = W[out][in] . X^T[in][token], with
// v_mfma_f32_32x32x16_bf16: the A operand is a weight tile (row-major [out][in], nn.Linear's layout), the B operand
// holds tokens on lanes.  Each wave owns 32 tokens; its residual R^T[256][32] is 8 accumulator tiles (128 registers).
// LayerNorm is then a per-lane reduction over registers (+1 exchange with lane^32), and an accumulator tile is
// directly the B operand of the next product (LN -> QKV, LN -> FFN1, relu(FFN1) -> FFN2, Q^T -> scores, P^T -> P.V,
// O^T -> out-proj) with the k-order permutation the hardware layout implies (frag_from_acc).
//
// Attention (per head): Q^T and O^T never leave registers.  Every wave computes K^T (tokens on lanes) and V (operands
// swapped: tokens on accumulator rows) of its 32 tokens and writes them as 16-byte runs into the workgroup's K
// [128 keys][32] and V^T [32][128 keys] tiles.  Boards (17 tokens) straddle waves, so a wave scores its 32 queries
// against a 64-key window of those tiles that covers every board it touches; keys of other boards are masked by one
// extra MFMA step instead of vector instructions: a one-hot code of the board index, scaled by 16, appended to the
// reduction dimension adds 256 to every same-board score, which after the max-subtraction leaves exp2(-65) = 0 for
// every other key.
//
// Software pipeline: while the vector ALU runs head h's softmax the matrix pipe runs head h+1's Q/K/V projections;
// while it packs feed-forward chunk c (ReLU, bf16) the matrix pipe runs chunk c+1's first GEMM.  Weight tiles for the
// next head / chunk are always in flight behind the current one (two barriers per head, one per feed-forward chunk).
//
// Parameter folding (done by the caller when it packs the blobs, see include/g2048.h): LayerNorm's affine is folded
// into the following Linear (W' = W diag(gamma), b' = b + W beta), the key bias is dropped (it shifts all scores of a
// query equally) and the value bias moves into the out-proj bias (softmax rows sum to 1).
//
// Numerics mirror torch.autocast(bf16): GEMM inputs rounded to bf16, f32 accumulation, f32 residual stream,
// f32 LayerNorm / softmax statistics.
//
// Two-kernel form (when the caller passes a workspace).  Only the CLS row of the LAST layer is read by the "cls"
// reduction.  k_encoder_main<HEAD> runs layers 0..L-2, then only LayerNorm + the K/V projections of layer L-1, and
// parks K, V (bf16, 17 KB per board) and the CLS residual row in HBM.  k_encoder_tail batches the CLS tokens of 128
// boards per workgroup (one board per lane) through the rest of layer L-1: Q projection, attention of that single query
// over its board's 17 keys in-lane (K/V from HBM), out-proj, feed-forward.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <type_traits>

#include "../../include/g2048.h"

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef short s16x2 __attribute__((ext_vector_type(2)));

constexpr int D = 256, NH = 8, HD = 32, FF = 1024, SEQ = 17, NBOARD = 7, NTOK = 128, THREADS = 256;
constexpr int W_LAYER = 3 * D * D + D * D + FF * D + D * FF;        // bf16 elements per layer
constexpr int P_LAYER = D + D + 3 * D + D + D + D + FF + D;         // f32 elements per layer
// offsets inside the per-layer blobs
constexpr int WO_QKV = 0, WO_O = 3 * D * D, WO_1 = WO_O + D * D, WO_2 = WO_1 + FF * D;
constexpr int PO_LN1G = 0, PO_LN1B = D, PO_BQKV = 2 * D, PO_BO = 5 * D, PO_LN2G = 6 * D, PO_LN2B = 7 * D,
              PO_B1 = 8 * D, PO_B2 = 8 * D + FF;
constexpr int FFC = 64;               // feed-forward hidden units per pipeline stage
constexpr int TILE = 32 * D * 2;      // bytes of a [32][256] (= [256][32]) bf16 tile: 16 KiB
constexpr float SM_SCALE_LOG2E = 0.17677669529663687f * 1.4426950408889634f;  // 1/sqrt(32) * log2(e)

__device__ __forceinline__ int rowof(int reg, int h) { return (reg & 3) + 8 * (reg >> 2) + 4 * h; }

// swizzles: CPR = 16-byte chunks per row; chunk q of row r lives at q ^ swz(r), so that the 16 lanes of a
// ds_read_b128 lane group (16 rows, distinct mod 16) always hit 16 different 16-byte bank slots
template <int CPR> __device__ __forceinline__ int swz(int r);
template <> __device__ __forceinline__ int swz<32>(int r) { return r & 15; }
template <> __device__ __forceinline__ int swz<16>(int r) { return r & 15; }
template <> __device__ __forceinline__ int swz<8>(int r) { return (r >> 1) & 7; }
template <> __device__ __forceinline__ int swz<4>(int r) { return (r >> 2) & 3; }

// Per-lane byte offsets of this lane's operand fragments inside a swizzled tile, computed once per kernel so that
// every operand read is `ds_read_b128 base_vgpr + immediate` with no address arithmetic in the MFMA loops:
// the row of M-tile m / output tile j only adds a multiple of 32 rows (a constant), and the XOR swizzle depends on
// (row mod 32, chunk mod 16) only.
struct LaneOff {
    int a256[8];  // [32m + r][32 chunks]: chunk 2ks + h, ks mod 8   (+ 256 B for ks >= 8, + 16 KiB per M-tile)
    int a64[4];   // [32j + r][8 chunks]:  chunk 2ks + h             (+ 4 KiB per j)
    int a32[2];   // [32j + r][4 chunks]:  chunk 2ks + h             (+ 2 KiB per j)
    // per-lane SOURCE byte offsets of the LDS-DMA instructions (see dma_tile)
    unsigned s256[2], s64, s32;
};
__device__ __forceinline__ void lane_offsets(LaneOff &o, int r, int h, int w, int lane) {
    for (int k = 0; k < 8; ++k) o.a256[k] = r * 512 + (((2 * k + h) ^ swz<32>(r)) * 16);
    for (int k = 0; k < 4; ++k) o.a64[k] = r * 128 + (((2 * k + h) ^ swz<8>(r)) * 16);
    for (int k = 0; k < 2; ++k) o.a32[k] = r * 64 + (((2 * k + h) ^ swz<4>(r)) * 16);
    // DMA: wave-instruction t of wave w fills LDS chunks [(4t + w) * 64, +64); lane i supplies physical chunk
    // P = (4t + w) * 64 + i = (row, p) and must fetch logical chunk q = p ^ swz(row) of that row.
    for (int par = 0; par < 2; ++par) {  // [rows][32]: row = 8t + 2w + (i >> 5); the swizzle sees t only through t & 1
        const int x = 2 * w + (lane >> 5), row15 = (8 * par + x) & 15;
        o.s256[par] = (unsigned)(x * D * 2 + (((lane & 31) ^ row15) * 16));
    }
    {  // [256][8] slice of linear2.weight (row stride FF): row = 32t + 8w + (i >> 3)
        const int x = 8 * w + (lane >> 3);
        o.s64 = (unsigned)(x * FF * 2 + (((lane & 7) ^ swz<8>(x)) * 16));
    }
    {  // [256][4] slice of out_proj.weight (row stride D): row = 64t + 16w + (i >> 2)
        const int x = 16 * w + (lane >> 2);
        o.s32 = (unsigned)(x * D * 2 + (((lane & 3) ^ swz<4>(x)) * 16));
    }
}

// Operand fragment (8 consecutive bf16 = chunk 2ks + h of row 32*mt + r) from a swizzled weight tile: one
// ds_read_b128 at lane offset + constant.
// When the OTHER operand comes out of an accumulator (frag_from_acc) its element j of lane-half h is
// k = 16ks + 8(j>>2) + 4h + (j&3); tiles read against it are stored with the columns of every group of 16 in the
// order KPERM = [0 1 2 3 8 9 10 11 4 5 6 7 12 13 14 15] (the host packs the weights that way, the kernel writes K and
// V^T that way), so the same contiguous read delivers exactly those k.
template <int CPR>
__device__ __forceinline__ bf16x8 load_w(const char *tile, const LaneOff &o, int mt, int ks) {
    if (CPR == 32) return *reinterpret_cast<const bf16x8 *>(tile + o.a256[ks & 7] + (ks >> 3) * 256 + mt * 32 * 512);
    if (CPR == 8) return *reinterpret_cast<const bf16x8 *>(tile + o.a64[ks] + mt * 32 * 128);
    return *reinterpret_cast<const bf16x8 *>(tile + o.a32[ks] + mt * 32 * 64);
}
// same from a padded kernel-written tile (tail kernel)
__device__ __forceinline__ bf16x8 load_p(const char *row, int ks, int h) {
    return *reinterpret_cast<const bf16x8 *>(row + 2 * (16 * ks + 8 * h));
}

__device__ __forceinline__ f32x16 mfma(bf16x8 a, bf16x8 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
}

// An accumulator tile X[row][col=lane] as the B operand of a product that sums over X's rows: registers
// 8s..8s+7 are the fragment of k-step s, holding rows 16s + 8(j>>2) + 4h + (j&3).
__device__ __forceinline__ void frag_from_acc(const f32x16 &x, bf16x8 out[2]) {
    for (int s = 0; s < 2; ++s)
        for (int j = 0; j < 8; ++j) out[s][j] = (__bf16)x[8 * s + j];
}
// the same with ReLU: max(x, 0) on the packed bf16 pairs as 16-bit integers (negative floats are negative integers,
// -0.0 included), one v_pk_max_i16 per two elements
__device__ __forceinline__ void relu_frag_from_acc(const f32x16 &x, bf16x8 out[2]) {
    frag_from_acc(x, out);
    for (int s = 0; s < 2; ++s) {
        s16x2 *p = reinterpret_cast<s16x2 *>(&out[s]);
        const s16x2 zero = {0, 0};
        for (int q = 0; q < 4; ++q) p[q] = __builtin_elementwise_max(p[q], zero);
    }
}

// acc[i] = b[rowof(i, h)] for a 32-row tile: the bias enters through the accumulator's initial value (4 broadcast
// ds_read_b128) instead of 16 vector adds after the MFMA chain
__device__ __forceinline__ f32x16 bias_tile(const float *b32, int h) {
    f32x16 a;
    for (int g = 0; g < 4; ++g) {
        const f32x4 v = *reinterpret_cast<const f32x4 *>(b32 + 8 * g + 4 * h);
        for (int q = 0; q < 4; ++q) a[4 * g + q] = v[q];
    }
    return a;
}

// acc += W_tile[32 rows of M-tile mt][16*NK] . B  (SWAP: acc += B^T . W_tile^T, i.e. tokens on the accumulator's rows).
// The operand reads are ordinary loads here; pipe_mfma() after a group of these calls (one basic block) tells the
// scheduler to keep DEPTH reads in flight ahead of a back-to-back MFMA chain: with one wave per SIMD nothing else
// hides the ~100-cycle LDS latency, and left alone the compiler emits read -> wait -> MFMA per k-step.
template <int CPR, int NK, bool SWAP = false>
__device__ __forceinline__ f32x16 gemm_tile(const char *tile, const LaneOff &o, int mt, const bf16x8 *b, f32x16 acc) {
    bf16x8 a[NK];
    for (int ks = 0; ks < NK; ++ks) a[ks] = load_w<CPR>(tile, o, mt, ks);
    for (int ks = 0; ks < NK; ++ks) acc = SWAP ? mfma(b[ks], a[ks], acc) : mfma(a[ks], b[ks], acc);
    return acc;
}
// Scheduling recipe for one region (the code between two sched_fence() / barriers): NM MFMAs fed by NM operand reads
// (+ EXTRA reads issued up front: bias tiles), DEPTH reads in flight ahead of a back-to-back MFMA chain, and behind each
// of the first NV MFMAs VALU_PER plain vector instructions and TRANS_PER transcendental ones (the softmax / packing work
// that runs in the matrix pipe's shadow).  Group masks: 0x008 MFMA, 0x100 DS read, 0x002 VALU, 0x400 TRANS.
template <int NM, int EXTRA = 0, int NV = 0, int VALU_PER = 0, int TRANS_PER = 0, int DEPTH = 8>
__device__ __forceinline__ void pipe_mfma() {
    __builtin_amdgcn_sched_group_barrier(0x100, DEPTH + EXTRA, 0);
#pragma unroll
    for (int i = 0; i < NM; ++i) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        if (i < NM - DEPTH) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
        if (i < NV && VALU_PER) __builtin_amdgcn_sched_group_barrier(0x002, VALU_PER, 0);
        if (i < NV && TRANS_PER) __builtin_amdgcn_sched_group_barrier(0x400, TRANS_PER, 0);
    }
}
// compile-time loop: f(integral_constant<int, I>) for I = I0..N-1, every index a constant (register arrays stay registers,
// `if constexpr` on the step number prunes the body per step)
template <int I, int N, class F>
__device__ __forceinline__ void static_for(F &&f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        static_for<I + 1, N>(f);
    }
}
// nothing is scheduled across this point
__device__ __forceinline__ void sched_fence() { __builtin_amdgcn_sched_barrier(0); }

// Diagnostic build only (-DG2048_STAMPS, tools/stamps_encoder.py; never compiled into the product library): cycle stamps
// at phase boundaries, summed per phase in scalar registers and written to a table of their own at kernel exit.
#ifdef G2048_STAMPS
constexpr int N_STAMPS = 24;
__device__ unsigned long long g_stamps[N_STAMPS];
struct Stamps {
    unsigned long long last, acc[N_STAMPS];
    __device__ __forceinline__ void start() {
        for (int i = 0; i < N_STAMPS; ++i) acc[i] = 0;
        last = now();
    }
    __device__ __forceinline__ static unsigned long long now() {
        unsigned long long t;
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
        __builtin_amdgcn_sched_barrier(0);
        return t;
    }
    __device__ __forceinline__ void mark(int k) {
        const unsigned long long t = now();
        acc[k] += t - last;
        last = t;
    }
    __device__ __forceinline__ void flush(int lane, int w) {
        if (lane == 0 && w == 0 && blockIdx.x % 64 == 0)
            for (int i = 0; i < N_STAMPS; ++i) atomicAdd(&g_stamps[i], acc[i]);
    }
};
#define STAMP(k) stamps.mark(k)
#else
#define STAMP(k)
#endif

// LDS-DMA of a [rows][CPR*8] bf16 tile (global row stride ld elements) into a swizzled LDS image.  Asynchronous:
// complete for this wave after s_waitcnt vmcnt(..), for the other waves after the following barrier.
// `buffer_load_dwordx4 ... offen lds`: the weight blob is one buffer resource (SGPRs), the tile base + per-instruction
// step is the scalar offset, the per-lane part (LaneOff) the vector offset: no vector address arithmetic per fetch.
// rows * CPR / 256 wave-instructions per wave: 4 per 16 KiB.
typedef __amdgpu_buffer_rsrc_t rsrc_t;
template <int CPR>
__device__ __forceinline__ void dma_tile(char *dst, rsrc_t blob, unsigned src_byte, int ld, int rows, const LaneOff &o, int w) {
    const int n_inst = rows * CPR / (64 * 4);       // wave-instructions per wave
    const int rows_per_inst = 4 * 64 / CPR;         // rows covered by one instruction of all four waves
    for (int t = 0; t < n_inst; ++t) {
        const unsigned lane_off = CPR == 32 ? o.s256[t & 1] : (CPR == 8 ? o.s64 : o.s32);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(blob, (__attribute__((address_space(3))) void *)(dst + (t * 4 + w) * 1024), 16,
                                                 (int)lane_off, (int)(src_byte + (unsigned)(t * rows_per_inst * ld * 2)), 0, 0);
    }
}
// one wave-instruction (piece t) of dma_tile
template <int CPR>
__device__ __forceinline__ void dma_inst(char *dst, rsrc_t blob, unsigned src_byte, int ld, int t, const LaneOff &o, int w) {
    const int rows_per_inst = 4 * 64 / CPR;
    const unsigned lane_off = CPR == 32 ? o.s256[t & 1] : (CPR == 8 ? o.s64 : o.s32);
    __builtin_amdgcn_raw_ptr_buffer_load_lds(blob, (__attribute__((address_space(3))) void *)(dst + (t * 4 + w) * 1024), 16,
                                             (int)lane_off, (int)(src_byte + (unsigned)(t * rows_per_inst * ld * 2)), 0, 0);
}
// all but the N youngest vector-memory operations of this wave have completed (they retire in issue order)
template <int N> __device__ __forceinline__ void dma_wait() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
__device__ __forceinline__ void dma_wait_all() { dma_wait<0>(); }

__device__ __forceinline__ void stage_f32(float *dst, const float *src, int n, int tid) {
    for (int i = tid; i < n; i += THREADS) dst[i] = src[i];
}

// Park an MFMA operand fragment in the accumulator half of the register file (MFMA reads A/B operands from either half).
// With one wave per SIMD the 256 architectural VGPRs are the scarce half: the normalised activations (64 registers) and
// the masking constants would otherwise crowd out the operand prefetch buffers of the MFMA chains.
__device__ __forceinline__ void park_in_agpr(bf16x8 &x) { asm volatile("" : "+a"(x)); }

// lanes l and l^32 hold the two halves of a token's reduction: one v_permlane32_swap gives both lanes both halves
__device__ __forceinline__ float xhalf_sum(float x) {
    const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(x), false, false);
    return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}
__device__ __forceinline__ float xhalf_max(float x) {
    const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(x), false, false);
    return fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
}

// Normalisation over the 256 features of each token (lane = token; this lane holds rows 4h + ... of every tile, the
// partner lane^32 the others) -> bf16 B-operand fragments for a K = 256 product (k-step 2j + s).  The affine part
// lives in the following Linear's packed weights.  One pass over the registers for both moments.
__device__ __forceinline__ void layer_norm(const f32x16 r[8], bf16x8 out[16]) {
    // four independent packed accumulators per moment (v_pk_add_f32 / v_pk_fma_f32): one dependent chain of 128 adds + 128 fmas
    // per token was ~2 k cycles of pure latency per LayerNorm with a single wave on the SIMD (7 LayerNorms per tile)
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    f32x2 s2[4], q2[4];
    for (int k = 0; k < 4; ++k) s2[k] = q2[k] = f32x2{0.f, 0.f};
    for (int j = 0; j < 8; ++j)
        for (int i = 0; i < 16; i += 2) {
            const f32x2 v = {r[j][i], r[j][i + 1]};
            const int k = (i >> 1) & 3;
            s2[k] += v;
            q2[k] = __builtin_elementwise_fma(v, v, q2[k]);
        }
    const f32x2 st = (s2[0] + s2[1]) + (s2[2] + s2[3]), qt = (q2[0] + q2[1]) + (q2[2] + q2[3]);
    float s = st[0] + st[1], ss = qt[0] + qt[1];
    s = xhalf_sum(s);
    ss = xhalf_sum(ss);
    const float mean = s * (1.0f / D);
    const float var = fmaxf(__builtin_fmaf(-mean, mean, ss * (1.0f / D)), 0.0f);
    const float rstd = rsqrtf(var + 1e-5f), shift = -mean * rstd;
    for (int j = 0; j < 8; ++j)
        for (int i = 0; i < 16; ++i) out[2 * j + (i >> 3)][i & 7] = (__bf16)__builtin_fmaf(r[j][i], rstd, shift);
    for (int k = 0; k < 16; ++k) park_in_agpr(out[k]);
}

constexpr int MODE_FULL = 0, MODE_HEAD = 1;
// HBM workspace of the two-kernel form, per board: K and V of the last layer as [head][key][lane half][16] bf16 (the 16
// values a lane half holds for one (token, head) are contiguous), then the CLS residual row (256 f32).
constexpr int64_t KV_ELEMS = (int64_t)NH * SEQ * HD;  // per board, per K or V

// ---------------------------------------------------------------------------------------------------------------------
// main kernel
// ---------------------------------------------------------------------------------------------------------------------
// LDS map (bytes).  Attention block: in_proj tiles of two heads, the out-proj slice of one, one head's K and V^T.
// Feed-forward block: linear1 / linear2 slices of two 64-unit chunks each, over the same 128 KiB.
constexpr int L_WQKV = 0;                 // [2][3 TILE]: q, k, v tile of head (h & 1)
constexpr int L_WO = 6 * TILE;            // [256][32] slice of out_proj.weight
constexpr int L_K = 7 * TILE;             // K   [128 keys][32 d]   bf16, 64-byte rows, chunks swizzled (CPR 4)
constexpr int L_VT = 7 * TILE + TILE / 2; // V^T [32 d][128 keys]   bf16, 256-byte rows, chunks swizzled (CPR 16)
constexpr int L_W1 = 0;                   // [2][2 TILE]: [64][256] slice of linear1.weight of chunk (c & 1)
constexpr int L_W2 = 4 * TILE;            // [2][2 TILE]: [256][64] slice of linear2.weight of chunk (c & 1)
struct LdsMain {
    char w[8 * TILE];
    float bq[D], bo[D], b1[FF], b2[D];
};
static_assert(sizeof(LdsMain) <= 160 * 1024, "LDS budget");

template <int MODE>
__global__ void __launch_bounds__(THREADS, 1)
k_encoder_main(const uint8_t *__restrict__ boards, const float *__restrict__ table, const float *__restrict__ cls,
               const __bf16 *__restrict__ wblob, const float *__restrict__ pblob, int n_layers,
               float *__restrict__ features, int64_t B, __bf16 *__restrict__ ws_k, __bf16 *__restrict__ ws_v,
               float *__restrict__ ws_r) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    LdsMain &L = *reinterpret_cast<LdsMain *>(smem);
    const int tid = threadIdx.x, lane = tid & 63, r = lane & 31, h = lane >> 5;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int64_t board0 = (int64_t)blockIdx.x * NBOARD;
    LaneOff lo;
    lane_offsets(lo, r, h, w, lane);

    // ---- this lane's token
    const int tok = 32 * w + r;                     // 0..127 inside the tile
    const int tb = tok / SEQ;                       // board in tile (7 = the 9 pad tokens)
    const int tc = tok - tb * SEQ;                  // position (0 = CLS)
    const bool tok_valid = tb < NBOARD && board0 + tb < B;

    // ---- attention window of this wave: 64 key slots starting at kb cover every board its 32 queries belong to
    const int kb = w == 0 ? 0 : (w == 1 ? 16 : (w == 2 ? 48 : 64));
    int k_rd[2], vt_rd[4], k_wr[2], vt_wr[2];
    for (int ks = 0; ks < 2; ++ks) k_rd[ks] = L_K + kb * 64 + lo.a32[ks];                 // + 32 rows * 64 B per key tile
    for (int ks = 0; ks < 4; ++ks) vt_rd[ks] = L_VT + r * 256 + (((kb / 8 + 2 * ks + h) ^ swz<16>(r)) * 16);
    for (int s = 0; s < 2; ++s) {
        k_wr[s] = L_K + tok * 64 + (((2 * s + h) ^ swz<4>(tok)) * 16);                    // lane = token, regs = d
        vt_wr[s] = L_VT + r * 256 + (((4 * w + 2 * s + h) ^ swz<16>(r)) * 16);            // lane = d, regs = tokens
    }
    // board one-hot codes (x16) for the masking MFMA step: element j of lane half h is reduction index 8h + j
    bf16x8 kmask[2], qmask;
    {
        const int qb = tb;
        for (int j = 0; j < 8; ++j) qmask[j] = (__bf16)((qb == 8 * h + j) ? 16.0f : 0.0f);
        for (int tl = 0; tl < 2; ++tl) {
            const int kbd = (kb + 32 * tl + r) / SEQ;
            for (int j = 0; j < 8; ++j) kmask[tl][j] = (__bf16)((kbd == 8 * h + j) ? 16.0f : 0.0f);
        }
        park_in_agpr(qmask);
        park_in_agpr(kmask[0]);
        park_in_agpr(kmask[1]);
    }

    // ---- embedding + positional code + CLS: R^T[f][tok]
    f32x16 R[8];
    {
        const float *src = cls;
        if (tok_valid && tc != 0) {
            int e = boards[(board0 + tb) * 16 + (tc - 1)];
            e = e > 30 ? 30 : e;  // the table has 31 rows per cell
            src = table + ((size_t)(tc - 1) * 31 + e) * D;
        }
        const float keep = tok_valid ? 1.0f : 0.0f;
        for (int j = 0; j < 8; ++j)
            for (int g = 0; g < 4; ++g) {
                const f32x4 v = *reinterpret_cast<const f32x4 *>(src + 32 * j + 8 * g + 4 * h);
                for (int q = 0; q < 4; ++q) R[j][4 * g + q] = keep * v[q];
            }
    }

    const int full_layers = MODE == MODE_HEAD ? n_layers - 1 : n_layers;
    char *const lds = L.w;
    // The weight blob as a buffer resource; tile addresses below are byte offsets into it (< 2^31: 1.5 MB per layer).
    // Fetches past its end (the "next layer" prefetch of the last layer) are dropped by the hardware's range check.
    const rsrc_t blob = __builtin_amdgcn_make_buffer_rsrc(const_cast<__bf16 *>(wblob), 0, n_layers * (W_LAYER * 2), 0x00020000);
    auto wqkv = [&](int hd) -> char * { return lds + L_WQKV + (hd & 1) * 3 * TILE; };
    // Weight stream, one wave-instruction (1 KiB per wave, 4 KiB per workgroup) at a time so that the fetches can be dealt
    // out between MFMA steps: 64 of them per head / feed-forward chunk keep the CU's vector-memory path busy for ~1200
    // cycles, which issued back to back stalls all four waves (lw = byte offset of the layer's weights).
    auto dma_qkv1 = [&](unsigned lw, int hd, int n) __attribute__((always_inline)) {  // n = 0..11: tile n/4 of {q, k, v}, piece n%4
        dma_inst<32>(wqkv(hd) + (n / 4) * TILE, blob, lw + 2u * (unsigned)(WO_QKV + ((n / 4) * D + HD * hd) * D), D, n % 4, lo, w);
    };
    auto dma_wo1 = [&](unsigned lw, int hd, int n) __attribute__((always_inline)) {   // n = 0..3
        dma_inst<4>(lds + L_WO, blob, lw + 2u * (unsigned)(WO_O + HD * hd), D, n, lo, w);
    };
    auto dma_w11 = [&](unsigned lw, int c, int n) __attribute__((always_inline)) {    // n = 0..7
        dma_inst<32>(lds + L_W1 + (c & 1) * 2 * TILE, blob, lw + 2u * (unsigned)(WO_1 + FFC * c * D), D, n, lo, w);
    };
    auto dma_w21 = [&](unsigned lw, int c, int n) __attribute__((always_inline)) {    // n = 0..7
        dma_inst<8>(lds + L_W2 + (c & 1) * 2 * TILE, blob, lw + 2u * (unsigned)(WO_2 + FFC * c), FF, n, lo, w);
    };

#ifdef G2048_STAMPS
    Stamps stamps;
    stamps.start();
#endif
    // prologue: in_proj tiles of heads 0 and 1 of layer 0 (HEAD mode with a single layer: its K/V part reads them)
    for (int n = 0; n < 12; ++n) dma_qkv1(0u, 0, n);
    for (int n = 0; n < 12; ++n) dma_qkv1(0u, 1, n);

    constexpr int PD = 8;  // operand reads in flight ahead of every MFMA chain
    bf16x8 xn[16];
#pragma nounroll
    for (int layer = 0; layer < full_layers; ++layer) {
        const unsigned lw = (unsigned)layer * (unsigned)(W_LAYER * 2);
        const float *P = pblob + (size_t)layer * P_LAYER;

        // ================= attention block =================
        // on entry: in_proj tiles of heads 0 and 1 are in flight (issued by the prologue / the previous layer's tail)
        // the layer's biases: loaded first, written to LDS after the LayerNorm (their latency hides behind it)
        float stg[7];
        stg[0] = P[PO_BQKV + tid];
        stg[1] = P[PO_BO + tid];
        stg[2] = P[PO_B2 + tid];
        for (int q = 0; q < 4; ++q) stg[3 + q] = P[PO_B1 + 256 * q + tid];
        layer_norm(R, xn);
        L.bq[tid] = stg[0];
        L.bo[tid] = stg[1];
        L.b2[tid] = stg[2];
        for (int q = 0; q < 4; ++q) L.b1[256 * q + tid] = stg[3 + q];
        dma_wait_all();
        __syncthreads();  // tiles of heads 0/1 and the biases visible
        STAMP(0);

        // ---- explicit software pipeline.  Step i = MFMA i of a head's q | k | v projection chains (K = 256: 16 steps each),
        // the operand read PD steps ahead, every third step one weight fetch, and a slice of the previous head's softmax for
        // the vector ALU; sched_fence() ends every step, so each MFMA is followed by its few vector instructions instead
        // of the compiler's all-MFMAs-then-all-VALU order (one wave per SIMD: only this wave's own instructions can fill
        // the matrix pipe's shadow).  Q^T gets its bias through the accumulator; K^T has none (folded away); V is
        // computed with swapped operands, tokens on the accumulator's rows, so that V^T is written as 16-byte runs.
        f32x16 qa, ka, va;          // projections of the NEXT head (accumulators, packed during the out-proj steps)
        bf16x8 qf[2], kf[2], vf[2];
        auto project = [&](int hd, auto softmax_tag, auto fetch, f32x16 &s0, f32x16 &s1, bf16x8 *pf, float &sum) __attribute__((always_inline)) {
            constexpr bool SOFTMAX = decltype(softmax_tag)::value;
            const char *t = wqkv(hd);
            bf16x8 a[PD];
            qa = bias_tile(L.bq + HD * hd, h);
            for (int i = 0; i < 16; ++i) ka[i] = va[i] = 0.f;
            static_for<0, PD>([&](auto ic) __attribute__((always_inline)) {
                constexpr int i = decltype(ic)::value;
                a[i] = load_w<32>(t + (i / 16) * TILE, lo, 0, i % 16);
            });
            float m = 0.f, mc = 0.f;
            sched_fence();
            static_for<0, 48>([&](auto ic) __attribute__((always_inline)) {
                constexpr int i = decltype(ic)::value;
                constexpr int ks = i % 16;
                if constexpr (i < 16) qa = mfma(a[i % PD], xn[ks], qa);
                else if constexpr (i < 32) ka = mfma(a[i % PD], xn[ks], ka);
                else va = mfma(xn[ks], a[i % PD], va);
                if constexpr (i + PD < 48) a[i % PD] = load_w<32>(t + ((i + PD) / 16) * TILE, lo, 0, (i + PD) % 16);
                if constexpr (i % 3 == 0) fetch(std::integral_constant<int, i / 3>{});
                if constexpr (SOFTMAX) {
                    if constexpr (i < 8) {  // running maximum, 4 scores per step
                        const float x = fmaxf(fmaxf(s0[2 * i], s0[2 * i + 1]), fmaxf(s1[2 * i], s1[2 * i + 1]));
                        m = i == 0 ? x : fmaxf(m, x);
                    } else if constexpr (i == 8) {
                        m = xhalf_max(m);
                        mc = -m * SM_SCALE_LOG2E;
                        sum = 0.f;
                    } else if constexpr (i < 41) {  // one probability per step: p = exp2((s - max) / sqrt(32) * log2 e)
                        constexpr int e = i - 9;
                        if constexpr (e < 16) {
                            s0[e] = __builtin_amdgcn_exp2f(__builtin_fmaf(s0[e], SM_SCALE_LOG2E, mc));
                            sum += s0[e];
                        } else {
                            s1[e - 16] = __builtin_amdgcn_exp2f(__builtin_fmaf(s1[e - 16], SM_SCALE_LOG2E, mc));
                            sum += s1[e - 16];
                        }
                    } else if constexpr (i == 41) {
                        sum = xhalf_sum(sum);
                    } else if constexpr (i < 46) {  // bf16 operand fragments of P^T, one k-step (16 keys) per step
                        constexpr int k = i - 42;
                        for (int j = 0; j < 8; ++j) {
                            if constexpr (k < 2) pf[k][j] = (__bf16)s0[8 * (k & 1) + j];
                            else pf[k][j] = (__bf16)s1[8 * (k & 1) + j];
                        }
                    }
                }
                sched_fence();
            });
        };
        // pack piece n = 0..11 of the projections: half a k-step (4 values) of q | k | v per piece
        auto pack_qkv = [&](auto nc) __attribute__((always_inline)) {
            constexpr int n = decltype(nc)::value, which = n / 4, s = (n / 2) & 1, half = n & 1;
            for (int j = 0; j < 4; ++j) {
                if constexpr (which == 0) qf[s][4 * half + j] = (__bf16)qa[8 * s + 4 * half + j];
                else if constexpr (which == 1) kf[s][4 * half + j] = (__bf16)ka[8 * s + 4 * half + j];
                else vf[s][4 * half + j] = (__bf16)va[8 * s + 4 * half + j];
            }
        };
        auto store_kv = [&]() __attribute__((always_inline)) {
            for (int s = 0; s < 2; ++s) {
                *reinterpret_cast<bf16x8 *>(lds + k_wr[s]) = kf[s];
                *reinterpret_cast<bf16x8 *>(lds + vt_wr[s]) = vf[s];
            }
        };
        {
            f32x16 d0, d1;
            float dsum;
            project(0, std::false_type{}, [](auto) {}, d0, d1, nullptr, dsum);
            static_for<0, 12>(pack_qkv);
        }
        store_kv();
        __syncthreads();  // K / V^T of head 0 visible; head 0's in_proj tiles are free
        STAMP(1);

        // one head: scores + softmax + P.V for head hd (Q^T in qf, K / V^T in LDS), the next head's projections in the
        // shadow of the softmax, then out-proj with the packing of those projections in ITS shadow.  Fetched meanwhile:
        // this head's out-proj slice (waited for at barrier 1) and, NEXT = 0: the in_proj tiles of head hd+2; 1 (hd = 6):
        // linear1 chunk 0; 2 (hd = 7, the last head): nothing more, the feed-forward prologue fetches the rest.
        auto head = [&](int hd, auto next_tag) __attribute__((always_inline)) {
            constexpr int NEXT = decltype(next_tag)::value;
            // this head's K fragments first: their LDS latency hides behind the first fetches
            bf16x8 kr[4];
            for (int ks = 0; ks < 2; ++ks) {
                kr[2 * ks] = *reinterpret_cast<const bf16x8 *>(lds + k_rd[ks]);
                kr[2 * ks + 1] = *reinterpret_cast<const bf16x8 *>(lds + k_rd[ks] + 32 * 64);
            }
            sched_fence();
            auto fetch = [&](auto nc) __attribute__((always_inline)) {  // n = 0..15: this iteration's 16 fetches in issue order
                constexpr int n = decltype(nc)::value;
                if constexpr (n < 4) dma_wo1(lw, hd, n);
                else if constexpr (NEXT == 0) dma_qkv1(lw, hd + 2, n - 4);
                else if constexpr (NEXT == 1 && n < 12) dma_w11(lw, 0, n - 4);
            };
            if constexpr (NEXT == 2) static_for<0, 4>(fetch);
            sched_fence();
            STAMP(2);
            // S^T[key][query] = K Q^T for the two 32-key tiles of the window, + 256 on same-board pairs
            f32x16 s0 = {0}, s1 = {0};
            for (int ks = 0; ks < 2; ++ks) {
                s0 = mfma(kr[2 * ks], qf[ks], s0);
                s1 = mfma(kr[2 * ks + 1], qf[ks], s1);
            }
            s0 = mfma(kmask[0], qmask, s0);
            s1 = mfma(kmask[1], qmask, s1);
            sched_fence();
            STAMP(3);
            // softmax over the 64 key slots: in-lane, one exchange with lane^32 each for the maximum and the sum;
            // probabilities stay unnormalised (<= 1), O is scaled by 1/sum afterwards
            bf16x8 pf[4];
            float sum;
            if constexpr (NEXT != 2) {
                project(hd + 1, std::true_type{}, fetch, s0, s1, pf, sum);
            } else {
                float m = fmaxf(s0[0], s1[0]);
                for (int i = 1; i < 16; ++i) m = fmaxf(m, fmaxf(s0[i], s1[i]));
                m = xhalf_max(m);
                const float mc = -m * SM_SCALE_LOG2E;
                sum = 0.f;
                for (int i = 0; i < 16; ++i) {
                    s0[i] = __builtin_amdgcn_exp2f(__builtin_fmaf(s0[i], SM_SCALE_LOG2E, mc));
                    s1[i] = __builtin_amdgcn_exp2f(__builtin_fmaf(s1[i], SM_SCALE_LOG2E, mc));
                    sum += s0[i] + s1[i];
                }
                sum = xhalf_sum(sum);
                frag_from_acc(s0, pf);
                frag_from_acc(s1, pf + 2);
            }
            sched_fence();
            STAMP(4);
            // O^T[d][query] = V^T P^T over the window
            f32x16 o = {0};
            for (int ks = 0; ks < 4; ++ks) o = mfma(*reinterpret_cast<const bf16x8 *>(lds + vt_rd[ks]), pf[ks], o);
            const float inv = __builtin_amdgcn_rcpf(sum);
            for (int i = 0; i < 16; ++i) o[i] *= inv;
            bf16x8 of[2];
            frag_from_acc(o, of);
            STAMP(5);
            // barrier 1: every wave is done with this head's K / V^T and the next head's in_proj tiles; the out-proj
            // slice has landed (only the younger fetches may still be in flight)
            if constexpr (NEXT == 0) dma_wait<12>();
            if constexpr (NEXT == 1) dma_wait<8>();
            if constexpr (NEXT == 2) dma_wait<0>();
            // raw barrier: __syncthreads() would drain the fetches that are meant to stay in flight (vmcnt(0))
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            sched_fence();
            STAMP(6);
            // R^T += Wo[:, head] . O^T: 16 MFMAs, operand reads PD ahead, the next head's q / k / v packed in their shadow
            {
                const char *wo = lds + L_WO;
                bf16x8 a[PD];
                static_for<0, PD>([&](auto ic) __attribute__((always_inline)) {
                    constexpr int i = decltype(ic)::value;
                    a[i] = load_w<4>(wo, lo, i / 2, i % 2);
                });
                sched_fence();
                static_for<0, 16>([&](auto ic) __attribute__((always_inline)) {
                    constexpr int i = decltype(ic)::value;
                    R[i / 2] = mfma(a[i % PD], of[i % 2], R[i / 2]);
                    if constexpr (i + PD < 16) a[i % PD] = load_w<4>(wo, lo, (i + PD) / 2, (i + PD) % 2);
                    if constexpr (NEXT != 2 && i < 12) pack_qkv(ic);
                    if constexpr (NEXT != 2 && i == 12) store_kv();
                    sched_fence();
                });
            }
            // barrier 2: next head's K / V^T visible, everything fetched in this iteration has landed
            STAMP(7);
            dma_wait_all();
            STAMP(8);
            __syncthreads();
            STAMP(9);
        };
#pragma nounroll
        for (int hd = 0; hd < NH - 2; ++hd) head(hd, std::integral_constant<int, 0>{});
        head(NH - 2, std::integral_constant<int, 1>{});
        head(NH - 1, std::integral_constant<int, 2>{});

        // ================= feed-forward block =================
        // on entry: linear1 chunk 0 is in LDS (fetched under head 6); chunk 1 and linear2 chunk 0 are fetched here, dealt
        // out over the bias add and the first linear1 chain
        for (int j = 0; j < 8; ++j) {
            dma_w11(lw, 1, j);
            const f32x16 b = bias_tile(L.bo + 32 * j, h);
            for (int i = 0; i < 16; ++i) R[j][i] += b[i];
        }
        layer_norm(R, xn);
        STAMP(10);
        // One feed-forward chunk = one 64-step chain: h(c+1) = linear1 chunk c+1 (32 steps, bias through the accumulator)
        // with the packing of chunk c's activations (ReLU, bf16) sliced into its first 16 steps, then R^T += linear2 chunk
        // c (32 steps); the operand reads run PD ahead across the seam, every fourth step issues one weight fetch.
        // Accumulators ping-pong between (hA0, hA1) and (hB0, hB1) so that packing never waits for a register.
        f32x16 hA0, hA1, hB0, hB1;
        bf16x8 hf[4];
        auto pack_h = [&](auto nc, const f32x16 &g0, const f32x16 &g1) __attribute__((always_inline)) {  // piece n = 0..15: 2 values
            constexpr int n = decltype(nc)::value, k = n / 4, q = n % 4;
            for (int j = 0; j < 2; ++j) {
                if constexpr (k < 2) hf[k][2 * q + j] = (__bf16)g0[8 * (k & 1) + 2 * q + j];
                else hf[k][2 * q + j] = (__bf16)g1[8 * (k & 1) + 2 * q + j];
            }
            if constexpr (q == 3) {  // ReLU on the packed pairs as 16-bit integers
                s16x2 *p2 = reinterpret_cast<s16x2 *>(&hf[k]);
                const s16x2 zero = {0, 0};
                for (int u = 0; u < 4; ++u) p2[u] = __builtin_elementwise_max(p2[u], zero);
            }
        };
        // WITH1: run linear1 of chunk c1 into (n0, n1); WITH2: pack (g0, g1) = h(c2) and run linear2 of chunk c2
        auto ffn_chain = [&](auto with1_tag, auto with2_tag, int c1, int c2, f32x16 &n0, f32x16 &n1, const f32x16 &g0, const f32x16 &g1,
                             auto fetch) __attribute__((always_inline)) {
            constexpr bool WITH1 = decltype(with1_tag)::value, WITH2 = decltype(with2_tag)::value;
            constexpr int N1 = WITH1 ? 32 : 0, NS = N1 + (WITH2 ? 32 : 0);
            const char *t1 = lds + L_W1 + (c1 & 1) * 2 * TILE, *t2 = lds + L_W2 + (c2 & 1) * 2 * TILE;
            if constexpr (WITH1) {
                n0 = bias_tile(L.b1 + FFC * c1, h);
                n1 = bias_tile(L.b1 + FFC * c1 + 32, h);
            }
            bf16x8 a[PD];
            auto rd = [&](auto sc) __attribute__((always_inline)) {  // operand of step s
                constexpr int s = decltype(sc)::value;
                if constexpr (s < N1) a[s % PD] = load_w<32>(t1, lo, s / 16, s % 16);
                else a[s % PD] = load_w<8>(t2, lo, (s - N1) / 4, (s - N1) % 4);
            };
            static_for<0, PD>(rd);
            if constexpr (!WITH1 && WITH2) static_for<0, 16>([&](auto nc) __attribute__((always_inline)) { pack_h(nc, g0, g1); });
            sched_fence();
            static_for<0, NS>([&](auto ic) __attribute__((always_inline)) {
                constexpr int i = decltype(ic)::value;
                if constexpr (i < N1) {
                    if constexpr (i < 16) n0 = mfma(a[i % PD], xn[i % 16], n0);
                    else n1 = mfma(a[i % PD], xn[i % 16], n1);
                } else {
                    R[(i - N1) / 4] = mfma(a[i % PD], hf[(i - N1) % 4], R[(i - N1) / 4]);
                }
                if constexpr (i + PD < NS) rd(std::integral_constant<int, i + PD>{});
                // 16 fetches per chain, early enough to have landed at its end: every third step of 64, every second of 32
                if constexpr (NS == 64 ? (i % 3 == 0 && i < 48) : i % 2 == 0) fetch(std::integral_constant<int, NS == 64 ? i / 3 : i / 2>{});
                if constexpr (WITH1 && WITH2 && i < 16) pack_h(ic, g0, g1);
                sched_fence();
            });
        };
        {   // h(0): linear1 chunk 0 alone, with the 8 fetches of linear2 chunk 0
            f32x16 dummy0, dummy1;
            ffn_chain(std::true_type{}, std::false_type{}, 0, 0, hA0, hA1, dummy0, dummy1, [&](auto nc) __attribute__((always_inline)) {
                constexpr int n = decltype(nc)::value;
                if constexpr (n < 8) dma_w21(lw, 0, n);
            });
        }
        dma_wait_all();
        __syncthreads();  // linear1 chunk 1 / linear2 chunk 0 landed; every wave has read linear1 chunk 0
        STAMP(11);
        // chunk c: LAST = 0: plain (fetches linear1 chunk c+2 and linear2 chunk c+1); 1: c = 14 (linear2 chunk 15 and the
        // next layer's q / k tiles of head 0); 2: c = 15 (no linear1; the next layer's v tile of head 0 and head 1's tiles)
        auto chunk = [&](int c, auto last_tag, f32x16 &n0, f32x16 &n1, const f32x16 &g0, const f32x16 &g1) __attribute__((always_inline)) {
            constexpr int LAST = decltype(last_tag)::value;
            auto fetch = [&](auto nc) __attribute__((always_inline)) {
                constexpr int n = decltype(nc)::value;
                if constexpr (LAST == 0) {
                    if constexpr (n < 8) dma_w11(lw, c + 2, n);
                    else dma_w21(lw, c + 1, n - 8);
                } else if constexpr (LAST == 1) {
                    if constexpr (n < 8) dma_w21(lw, c + 1, n);
                    else dma_qkv1(lw + W_LAYER * 2, 0, n - 8);        // q, k tiles of head 0
                } else {
                    if constexpr (n < 4) dma_qkv1(lw + W_LAYER * 2, 0, 8 + n);  // v tile of head 0
                    else dma_qkv1(lw + W_LAYER * 2, 1, n - 4);
                }
            };
            STAMP(12);
            if constexpr (LAST != 2) ffn_chain(std::true_type{}, std::true_type{}, c + 1, c, n0, n1, g0, g1, fetch);
            else ffn_chain(std::false_type{}, std::true_type{}, 0, c, n0, n1, g0, g1, fetch);
            STAMP(14);
            if constexpr (LAST == 2)  // linear2's bias, read before the barrier behind which the next layer restages it
                for (int j = 0; j < 8; ++j) {
                    const f32x16 b = bias_tile(L.b2 + 32 * j, h);
                    for (int i = 0; i < 16; ++i) R[j][i] += b[i];
                }
            dma_wait_all();
            STAMP(15);
            __syncthreads();  // fetched tiles landed; this chunk's tiles are free
            STAMP(16);
        };
#pragma nounroll
        for (int c = 0; c < FF / FFC - 2; c += 2) {
            chunk(c, std::integral_constant<int, 0>{}, hB0, hB1, hA0, hA1);
            chunk(c + 1, std::integral_constant<int, 0>{}, hA0, hA1, hB0, hB1);
        }
        chunk(FF / FFC - 2, std::integral_constant<int, 1>{}, hB0, hB1, hA0, hA1);
        chunk(FF / FFC - 1, std::integral_constant<int, 2>{}, hA0, hA1, hB0, hB1);
    }

    STAMP(17);
    if (MODE == MODE_HEAD) {
        // ---- last layer, K/V only: LayerNorm, K^T and V^T of every head -> HBM, CLS residual row -> HBM.
        // On entry the in_proj tiles of heads 0 and 1 of this layer are in LDS or in flight (previous layer's tail / prologue).
        const unsigned lwl = (unsigned)(n_layers - 1) * (unsigned)(W_LAYER * 2);
        if (tok_valid && tc == 0) {
            float *dst = ws_r + (board0 + tb) * D;
            for (int j = 0; j < 8; ++j)
                for (int g = 0; g < 4; ++g) {
                    f32x4 v;
                    for (int q = 0; q < 4; ++q) v[q] = R[j][4 * g + q];
                    *reinterpret_cast<f32x4 *>(dst + 32 * j + 8 * g + 4 * h) = v;
                }
        }
        layer_norm(R, xn);
        dma_wait_all();
        __syncthreads();
        // K^T and V^T of head hd from the k / v tiles in buffer (hd & 1): 32 steps.  FETCH: every fourth step one piece of head
        // hd+1's k / v tiles goes into the other buffer (free since the barrier that ended iteration hd-1); heads 0 and 1
        // are already there.
        bf16x8 kst[2], vst[2];  // head hd-1's K^T / V^T, stored one iteration late: the wait at the end of an iteration then
                                // covers stores issued a whole chain earlier instead of stalling on HBM write latency
        auto store_head = [&](int hd) __attribute__((always_inline)) {
            if (tok_valid) {
                const size_t off = (size_t)(board0 + tb) * KV_ELEMS + ((size_t)hd * SEQ + tc) * HD + 16 * h;
                *reinterpret_cast<bf16x8 *>(ws_k + off) = kst[0];
                *reinterpret_cast<bf16x8 *>(ws_k + off + 8) = kst[1];
                *reinterpret_cast<bf16x8 *>(ws_v + off) = vst[0];
                *reinterpret_cast<bf16x8 *>(ws_v + off + 8) = vst[1];
            }
        };
        auto kv_head = [&](int hd, auto fetch_tag, auto store_tag) __attribute__((always_inline)) {
            constexpr bool FETCH = decltype(fetch_tag)::value, STORE = decltype(store_tag)::value;
            if constexpr (STORE) store_head(hd - 1);
            const char *t = wqkv(hd);
            f32x16 ka2 = {0}, va2 = {0};
            bf16x8 a[PD];
            static_for<0, PD>([&](auto ic) __attribute__((always_inline)) {
                constexpr int i = decltype(ic)::value;
                a[i] = load_w<32>(t + (1 + i / 16) * TILE, lo, 0, i % 16);
            });
            sched_fence();
            static_for<0, 32>([&](auto ic) __attribute__((always_inline)) {
                constexpr int i = decltype(ic)::value;
                if constexpr (i < 16) ka2 = mfma(a[i % PD], xn[i % 16], ka2);
                else va2 = mfma(a[i % PD], xn[i % 16], va2);
                if constexpr (i + PD < 32) a[i % PD] = load_w<32>(t + (1 + (i + PD) / 16) * TILE, lo, 0, (i + PD) % 16);
                if constexpr (FETCH && i % 3 == 0 && i < 24) dma_qkv1(lwl, hd + 1, 4 + i / 3);
                if constexpr (i >= 20 && i < 22)
                    for (int j = 0; j < 8; ++j) kst[i - 20][j] = (__bf16)ka2[8 * (i - 20) + j];
                sched_fence();
            });
            frag_from_acc(va2, vst);
            dma_wait_all();
            __syncthreads();  // head hd's tiles are free, head hd+1's have landed
        };
        kv_head(0, std::false_type{}, std::false_type{});
#pragma nounroll
        for (int hd = 1; hd < NH - 1; ++hd) kv_head(hd, std::true_type{}, std::true_type{});
        kv_head(NH - 1, std::false_type{}, std::true_type{});
        store_head(NH - 1);
        STAMP(18);
#ifdef G2048_STAMPS
        stamps.flush(lane, w);
#endif
        return;
    }

    // ---- CLS rows out
    if (tok_valid && tc == 0) {
        float *dst = features + (board0 + tb) * D;
        for (int j = 0; j < 8; ++j)
            for (int g = 0; g < 4; ++g) {
                f32x4 v;
                for (int q = 0; q < 4; ++q) v[q] = R[j][4 * g + q];
                *reinterpret_cast<f32x4 *>(dst + 32 * j + 8 * g + 4 * h) = v;
            }
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// tail kernel: the CLS token of 128 boards per workgroup (one board per lane) through the rest of the last layer
// ---------------------------------------------------------------------------------------------------------------------
constexpr int ST32 = 2 * 32 + 16;     // padded row of the kernel-written O tile
struct LdsTail {
    char w[6 * TILE];                 // wq @0, wo @3 TILE / 4 TILE (alternating heads); feed-forward: w1[c&1] @0 / 2 TILE, w2[0] @4 TILE
    char w2b[2 * TILE];               // w2[1]
    char o[NTOK * ST32];
    float bias[FF];
};
static_assert(sizeof(LdsTail) <= 160 * 1024, "LDS budget");

__global__ void __launch_bounds__(THREADS, 1)
k_encoder_tail(const __bf16 *__restrict__ wblob, const float *__restrict__ pblob, int n_layers, float *__restrict__ features,
               int64_t B, const __bf16 *__restrict__ ws_k, const __bf16 *__restrict__ ws_v, const float *__restrict__ ws_r) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    LdsTail &L = *reinterpret_cast<LdsTail *>(smem);
    const int tid = threadIdx.x, lane = tid & 63, r = lane & 31, h = lane >> 5;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int64_t board0 = (int64_t)blockIdx.x * NTOK;
    char *const wq = L.w;
    LaneOff lo;
    lane_offsets(lo, r, h, w, lane);
    const int tok = 32 * w + r;
    const bool tok_valid = board0 + tok < B;
    const int64_t my_board = tok_valid ? board0 + tok : (B - 1);  // clamped: loads stay in bounds, stores are guarded

    f32x16 R[8];
    {
        const float *src = ws_r + my_board * D;
        for (int j = 0; j < 8; ++j)
            for (int g = 0; g < 4; ++g) {
                const f32x4 v = *reinterpret_cast<const f32x4 *>(src + 32 * j + 8 * g + 4 * h);
                for (int q = 0; q < 4; ++q) R[j][4 * g + q] = v[q];
            }
    }
    const rsrc_t blob = __builtin_amdgcn_make_buffer_rsrc(const_cast<__bf16 *>(wblob), 0, n_layers * (W_LAYER * 2), 0x00020000);
    const unsigned lw = (unsigned)(n_layers - 1) * (unsigned)(W_LAYER * 2);
    const float *P = pblob + (size_t)(n_layers - 1) * P_LAYER;
    auto dma_q = [&](int hd) { dma_tile<32>(wq, blob, lw + 2u * (unsigned)(WO_QKV + HD * hd * D), D, HD, lo, w); };
    auto wo_of = [&](int hd) -> char * { return L.w + (3 + (hd & 1)) * TILE; };  // double-buffered
    auto dma_wo = [&](int hd) { dma_tile<4>(wo_of(hd), blob, lw + 2u * (unsigned)(WO_O + HD * hd), D, D, lo, w); };

    // ================= attention block =================
    dma_q(0);
    dma_wo(0);
    stage_f32(L.bias, P + PO_BQKV, D, tid);
    bf16x8 xn[16];
    layer_norm(R, xn);
    dma_wait_all();
    __syncthreads();
#pragma nounroll
    for (int hd = 0; hd < NH; ++hd) {
        // Q^T [32 d][32 boards] of the CLS tokens; the single query of a board attends to that board's 17 keys
        // in-lane: this lane half holds 16 of the 32 head dims (d = rowof(i, h)), exactly the 16 contiguous
        // values the main kernel parked per (board, head, key, half)
        f32x16 qa = gemm_tile<32, 16>(wq, lo, 0, xn, bias_tile(L.bias + HD * hd, h));
        pipe_mfma<16>();
        const size_t off = (size_t)my_board * KV_ELEMS + (size_t)hd * SEQ * HD + 16 * h;
        float sc[SEQ], m = -3.0e38f;
        for (int key = 0; key < SEQ; ++key) {
            const bf16x8 k0 = *reinterpret_cast<const bf16x8 *>(ws_k + off + key * HD);
            const bf16x8 k1 = *reinterpret_cast<const bf16x8 *>(ws_k + off + key * HD + 8);
            float d = 0.f;
            // q is rounded to bf16 like the operand of the main kernel's K Q^T product
            for (int i = 0; i < 8; ++i) d += (float)(__bf16)qa[i] * (float)k0[i] + (float)(__bf16)qa[8 + i] * (float)k1[i];
            d = xhalf_sum(d);
            sc[key] = d * 0.17677669529663687f;
            m = fmaxf(m, sc[key]);
        }
        float sum = 0.f;
        for (int key = 0; key < SEQ; ++key) {
            sc[key] = __expf(sc[key] - m);
            sum += sc[key];
        }
        const float inv = 1.0f / sum;
        float o[16];
        for (int i = 0; i < 16; ++i) o[i] = 0.f;
        for (int key = 0; key < SEQ; ++key) {
            const bf16x8 v0 = *reinterpret_cast<const bf16x8 *>(ws_v + off + key * HD);
            const bf16x8 v1 = *reinterpret_cast<const bf16x8 *>(ws_v + off + key * HD + 8);
            const float p = (float)(__bf16)sc[key];  // bf16 probabilities (unnormalised), as the MFMA operand would be
            for (int i = 0; i < 8; ++i) {
                o[i] += p * (float)v0[i];
                o[8 + i] += p * (float)v1[i];
            }
        }
        // O row of this token, head dims in KPERM order (the out-proj slice is packed for accumulator-fed operands):
        // registers 8s..8s+7 are positions 16s + 8h .. +7
        for (int s = 0; s < 2; ++s) {
            bf16x8 ov;
            for (int q = 0; q < 8; ++q) ov[q] = (__bf16)(o[8 * s + q] * inv);
            *reinterpret_cast<bf16x8 *>(L.o + tok * ST32 + 2 * (16 * s + 8 * h)) = ov;
        }
        __syncthreads();  // O visible; this head's Q tile is free
        if (hd + 1 < NH) {
            dma_q(hd + 1);
            dma_wo(hd + 1);
        }
        bf16x8 of[2];
        for (int ks = 0; ks < 2; ++ks) of[ks] = load_p(L.o + tok * ST32, ks, h);
        const char *wo = wo_of(hd);
        for (int j = 0; j < 8; ++j) R[j] = gemm_tile<4, 2>(wo, lo, j, of, R[j]);
        pipe_mfma<16>();
        dma_wait_all();
        __syncthreads();  // next head's tiles landed; O of this head is free
    }

    // ================= feed-forward block =================
    auto w1_of = [&](int c) -> char * { return L.w + ((c & 1) ? 2 * TILE : 0); };
    auto w2_of = [&](int c) -> char * { return (c & 1) ? L.w2b : L.w + 4 * TILE; };
    auto dma_ffn = [&](int c) {
        dma_tile<32>(w1_of(c), blob, lw + 2u * (unsigned)(WO_1 + FFC * c * D), D, FFC, lo, w);
        dma_tile<8>(w2_of(c), blob, lw + 2u * (unsigned)(WO_2 + FFC * c), FF, D, lo, w);
    };
    dma_ffn(0);
    stage_f32(L.bias, P + PO_BO, D, tid);
    __syncthreads();
    for (int j = 0; j < 8; ++j) {
        const f32x16 b = bias_tile(L.bias + 32 * j, h);
        for (int i = 0; i < 16; ++i) R[j][i] += b[i];
    }
    layer_norm(R, xn);
    __syncthreads();
    stage_f32(L.bias, P + PO_B1, FF, tid);
    dma_wait_all();
    __syncthreads();
#pragma nounroll
    for (int c = 0; c < FF / FFC; ++c) {
        if (c + 1 < FF / FFC) dma_ffn(c + 1);  // lands while this stage computes
        const char *t1 = w1_of(c), *t2 = w2_of(c);
        f32x16 h0 = gemm_tile<32, 16>(t1, lo, 0, xn, bias_tile(L.bias + FFC * c, h));
        f32x16 h1 = gemm_tile<32, 16>(t1, lo, 1, xn, bias_tile(L.bias + FFC * c + 32, h));
        pipe_mfma<32>();
        bf16x8 hf[4];
        relu_frag_from_acc(h0, hf);
        relu_frag_from_acc(h1, hf + 2);
        for (int j = 0; j < 8; ++j) R[j] = gemm_tile<8, 4>(t2, lo, j, hf, R[j]);
        pipe_mfma<32>();
        dma_wait_all();
        __syncthreads();  // next stage landed; this stage's tiles are free
    }
    if (tok_valid) {
        float *dst = features + (board0 + tok) * D;
        for (int j = 0; j < 8; ++j)
            for (int g = 0; g < 4; ++g) {
                const f32x4 b = *reinterpret_cast<const f32x4 *>(P + PO_B2 + 32 * j + 8 * g + 4 * h);
                f32x4 v;
                for (int q = 0; q < 4; ++q) v[q] = R[j][4 * g + q] + b[q];
                *reinterpret_cast<f32x4 *>(dst + 32 * j + 8 * g + 4 * h) = v;
            }
    }
}

}  // namespace

#ifdef G2048_STAMPS
extern "C" int g2048_debug_stamps(unsigned long long *out /*host [24]*/, int reset) {
    if (hipMemcpyFromSymbol(out, HIP_SYMBOL(g_stamps), sizeof(unsigned long long) * N_STAMPS) != hipSuccess) return -1;
    if (reset) {
        unsigned long long z[N_STAMPS] = {0};
        if (hipMemcpyToSymbol(HIP_SYMBOL(g_stamps), z, sizeof(z)) != hipSuccess) return -1;
    }
    return N_STAMPS;
}
#endif

extern "C" int64_t g2048_policy_encoder_workspace_bytes(int64_t B) {
    return B <= 0 ? 0 : B * (2 * KV_ELEMS * (int64_t)sizeof(__bf16) + D * (int64_t)sizeof(float));
}

extern "C" int g2048_policy_encoder(const uint8_t *boards, const float *embed_table, const float *cls_token,
                                    const void *weights_bf16, const float *params_f32, int n_layers,
                                    float *features, int64_t B, void *workspace, void *stream) {
    if (!boards || !embed_table || !cls_token || !weights_bf16 || !params_f32 || !features || n_layers < 1 || B <= 0)
        return G2048_EINVAL;
    if (((uintptr_t)weights_bf16 & 15) || ((uintptr_t)params_f32 & 15) || ((uintptr_t)embed_table & 15) ||
        ((uintptr_t)cls_token & 15) || ((uintptr_t)features & 15) || ((uintptr_t)workspace & 15))
        return G2048_EINVAL;
    // the dynamic-LDS limit is a per-device function attribute: set on every call (no latch, no global state)
    const void *main_fn = workspace ? reinterpret_cast<const void *>(k_encoder_main<MODE_HEAD>)
                                    : reinterpret_cast<const void *>(k_encoder_main<MODE_FULL>);
    if (hipFuncSetAttribute(main_fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sizeof(LdsMain)) != hipSuccess ||
        (workspace && hipFuncSetAttribute(reinterpret_cast<const void *>(k_encoder_tail), hipFuncAttributeMaxDynamicSharedMemorySize,
                                          (int)sizeof(LdsTail)) != hipSuccess))
        return -(1000 + (int)hipGetLastError());
    const __bf16 *wb = reinterpret_cast<const __bf16 *>(weights_bf16);
    const unsigned blocks = (unsigned)((B + NBOARD - 1) / NBOARD);
    if (!workspace) {
        hipLaunchKernelGGL(k_encoder_main<MODE_FULL>, dim3(blocks), dim3(THREADS), sizeof(LdsMain), (hipStream_t)stream, boards,
                           embed_table, cls_token, wb, params_f32, n_layers, features, B, (__bf16 *)nullptr, (__bf16 *)nullptr,
                           (float *)nullptr);
    } else {
        __bf16 *ws_k = reinterpret_cast<__bf16 *>(workspace), *ws_v = ws_k + B * KV_ELEMS;
        float *ws_r = reinterpret_cast<float *>(ws_v + B * KV_ELEMS);
        hipLaunchKernelGGL(k_encoder_main<MODE_HEAD>, dim3(blocks), dim3(THREADS), sizeof(LdsMain), (hipStream_t)stream, boards,
                           embed_table, cls_token, wb, params_f32, n_layers, features, B, ws_k, ws_v, ws_r);
        hipLaunchKernelGGL(k_encoder_tail, dim3((unsigned)((B + NTOK - 1) / NTOK)), dim3(THREADS), sizeof(LdsTail),
                           (hipStream_t)stream, wb, params_f32, n_layers, features, B, ws_k, ws_v, ws_r);
    }
    const hipError_t e = hipGetLastError();
    return e == hipSuccess ? 0 : -(1000 + (int)e);
}
