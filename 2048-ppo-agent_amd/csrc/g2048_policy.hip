// Fused Transformer-encoder forward for policy inference on gfx950 (bf16 MFMA, f32 accumulate).
//
// Replaces, for rollouts, the PyTorch path of PPOAgent.features() (reference src/ppo/ppo_agent.py:103-106,
// src/ppo/transformer_encoder.py:150-190: embedding + 2-D positional code + CLS token + L pre-norm encoder
// layers, "cls" reduction) at the reference's default shape: d_model 256, 8 heads of 32, feed-forward 1024,
// 17 tokens per board.  One workgroup (4 waves) carries 7 boards = 119 tokens (+9 pad) through ALL layers:
// the residual stream never leaves registers, the only activations that touch LDS are the per-head Q/K/V/O
// tiles, weights stream from L2 through LDS, and nothing but the 16-byte boards is read from / the 1 KiB CLS
// feature rows written to HBM.
//
// Orientation: every GEMM is computed transposed, Y^T[out_feature][token] = W[out][in] . X^T[in][token], with
// v_mfma_f32_32x32x16_bf16: the A operand is a weight tile (row-major [out][in], exactly nn.Linear's layout),
// the B operand holds tokens on lanes.  Each wave owns 32 tokens; its residual R^T[256][32] is 8 accumulator
// tiles (128 VGPRs).  In this orientation LayerNorm is a per-lane reduction over registers (+1 exchange with
// lane^32), and an accumulator tile is directly the B operand of the next GEMM (LN -> QKV, LN -> FFN1,
// relu(FFN1) -> FFN2) with the k-order permutation the hardware layout implies (see frag_from_acc).
//
// Numerics mirror torch.autocast(bf16): GEMM inputs rounded to bf16, f32 accumulation, f32 residual stream,
// f32 LayerNorm / softmax statistics.
//
// Two-kernel form (when the caller passes a workspace).  Only the CLS row of the LAST layer is read by the "cls"
// reduction, yet that layer's Q / out-proj / feed-forward weights (1.26 MB) would be streamed for 7 useful tokens per
// workgroup.  MODE_HEAD runs layers 0..L-2 as described, then only LayerNorm + the K/V projections of layer L-1, and
// parks K, V (bf16, 17 KB per board) and the CLS residual row in HBM.  MODE_TAIL batches the CLS tokens of 128 boards
// per workgroup (one board per lane) through the rest of layer L-1: Q projection, attention of that single query
// over its board's 17 keys in-lane (K/V from HBM), out-proj, feed-forward - the same code with 18x fewer workgroups.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

#include "../../include/g2048.h"

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

__constant__ int g_dbg = 0;

constexpr int D = 256, NH = 8, HD = 32, FF = 1024, SEQ = 17, NBOARD = 7, NTOK = 128, THREADS = 256;
constexpr int W_LAYER = 3 * D * D + D * D + FF * D + D * FF;        // bf16 elements per layer
constexpr int P_LAYER = D + D + 3 * D + D + D + D + FF + D;         // f32 elements per layer
// offsets inside the per-layer blobs
constexpr int WO_QKV = 0, WO_O = 3 * D * D, WO_1 = WO_O + D * D, WO_2 = WO_1 + FF * D;
constexpr int PO_LN1G = 0, PO_LN1B = D, PO_BQKV = 2 * D, PO_BO = 5 * D, PO_LN2G = 6 * D, PO_LN2B = 7 * D,
              PO_B1 = 8 * D, PO_B2 = 8 * D + FF;

// LDS images.
//  * Weight tiles are filled by LDS-DMA (global_load_lds_dwordx4: every wave-instruction lands 64 x 16 bytes
//    contiguously, the SOURCE address is per lane), so they are unpadded [rows][K] images whose 16-byte chunks are
//    XOR-swizzled inside a row: chunk q of row r lives at q ^ swz(r).  With that, the 16 lanes of a ds_read_b128
//    lane group (16 rows, distinct mod 16) always hit 16 different 16-byte bank slots.
//  * Tiles written by the kernel itself (Q, K, O: [token][32]; V^T: [32][7 x 32]) use rows padded by 16 bytes.
constexpr int ST32 = 2 * 32 + 16;
constexpr int FFC = 64;               // feed-forward hidden units per pipeline stage
constexpr int VT_COLS = NBOARD * 32;  // V^T keeps every board's 17 keys in its own 32-aligned column slot
constexpr int STVT = 2 * VT_COLS + 16;
constexpr int QK_ROWS = NTOK + 32;    // attention units read 32 rows starting at 17*b
constexpr int TILE = 32 * D * 2;      // bytes of a [32][256] (= [256][32]) bf16 tile: 16 KiB

struct LdsAttn {  // per-head activations; idle during the feed-forward block (then the start hosts a weight tile)
    char q[QK_ROWS * ST32], k[QK_ROWS * ST32];
    char vt[HD * STVT];
    char o[NTOK * ST32];
};
struct Lds {
    // attention block: wq @0, wk @TILE, wv @2 TILE, wo @3 TILE / 4 TILE (alternating heads).  feed-forward: w1[0] @0, w1[1] @2 TILE, w2[0] @4 TILE
    // (each 2 TILE), w2[1] aliases `act`.
    char w[6 * TILE];
    union {
        LdsAttn a;
        char w2b[2 * TILE];
    } act;
    float gamma[D], beta[D];
    float bias[FF];
};
static_assert(sizeof(LdsAttn) >= 2 * TILE, "attention scratch must be able to host one feed-forward weight tile");
static_assert(sizeof(Lds) <= 160 * 1024, "LDS budget");

__device__ __forceinline__ int rowof(int reg, int h) { return (reg & 3) + 8 * (reg >> 2) + 4 * h; }

// swizzles: CPR = 16-byte chunks per row
template <int CPR> __device__ __forceinline__ int swz(int r);
template <> __device__ __forceinline__ int swz<32>(int r) { return r & 15; }
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
// order KPERM = [0 1 2 3 8 9 10 11 4 5 6 7 12 13 14 15] (the host packs the weights that way, the kernel writes V^T
// that way), so the same contiguous read delivers exactly those k.
template <int CPR>
__device__ __forceinline__ bf16x8 load_w(const char *tile, const LaneOff &o, int mt, int ks) {
    if (CPR == 32) return *reinterpret_cast<const bf16x8 *>(tile + o.a256[ks & 7] + (ks >> 3) * 256 + mt * 32 * 512);
    if (CPR == 8) return *reinterpret_cast<const bf16x8 *>(tile + o.a64[ks] + mt * 32 * 128);
    return *reinterpret_cast<const bf16x8 *>(tile + o.a32[ks] + mt * 32 * 64);
}
// same from a padded kernel-written tile
__device__ __forceinline__ bf16x8 load_p(const char *row, int ks, int h) {
    return *reinterpret_cast<const bf16x8 *>(row + 2 * (16 * ks + 8 * h));
}
__device__ __forceinline__ int kperm_pos(int k) {  // storage column of logical k under KPERM
    const int p = k & 15, q = (p & 3) | ((p & 4) << 1) | ((p & 8) >> 1);
    return (k & ~15) | q;
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

// acc += W_tile[32 rows from `row`][16*NK] . B.  The operand reads are ordinary loads here; pipe_mfma() after a group
// of these calls (one basic block) tells the scheduler to keep DEPTH reads in flight ahead of a back-to-back MFMA
// chain: with one wave per SIMD nothing else hides the ~100-cycle LDS latency, and left alone the compiler emits
// read -> wait -> MFMA per k-step.
template <int CPR, int NK>
__device__ __forceinline__ f32x16 gemm_tile(const char *tile, const LaneOff &o, int mt, const bf16x8 *b, f32x16 acc) {
    bf16x8 a[NK];
    for (int ks = 0; ks < NK; ++ks) a[ks] = load_w<CPR>(tile, o, mt, ks);
    for (int ks = 0; ks < NK; ++ks) acc = mfma(a[ks], b[ks], acc);
    return acc;
}
template <int TOTAL, int DEPTH = 8>
__device__ __forceinline__ void pipe_mfma() {
    __builtin_amdgcn_sched_group_barrier(0x100, DEPTH, 0);
    for (int i = 0; i < TOTAL - DEPTH; ++i) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
    }
    __builtin_amdgcn_sched_group_barrier(0x008, DEPTH, 0);
}

// LDS-DMA of a [rows][CPR*8] bf16 tile (global row stride ld elements) into a swizzled LDS image.  Asynchronous:
// complete for this wave after s_waitcnt vmcnt(0), for the other waves after the following barrier.
// Source address = uniform tile base + uniform per-instruction step + per-lane offset from LaneOff.
template <int CPR>
__device__ __forceinline__ void dma_tile(char *dst, const __bf16 *src, int ld, int rows, const LaneOff &o, int w) {
    if (g_dbg & 1) return;  // timing-only switch (development): no weight stream, outputs are garbage
    const int n_inst = rows * CPR / (64 * 4);       // wave-instructions per wave
    const int rows_per_inst = 4 * 64 / CPR;         // rows covered by one instruction of all four waves
    const char *base = reinterpret_cast<const char *>(src);
    for (int t = 0; t < n_inst; ++t) {
        const unsigned lane_off = CPR == 32 ? o.s256[t & 1] : (CPR == 8 ? o.s64 : o.s32);
        const char *g = base + (size_t)t * rows_per_inst * ld * 2 + lane_off;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)g,
                                         (__attribute__((address_space(3))) void *)(dst + (t * 4 + w) * 1024), 16, 0, 0);
    }
}
__device__ __forceinline__ void dma_wait_all() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }

__device__ __forceinline__ void stage_f32(float *dst, const float *src, int n, int tid) {
    for (int i = tid; i < n; i += THREADS) dst[i] = src[i];
}

// LayerNorm over the 256 features of each token (lane = token; this lane holds rows 4h + ... of every tile, the
// partner lane^32 the others) -> bf16 B-operand fragments for a K = 256 product (k-step 2j + s).
__device__ __forceinline__ void layer_norm(const f32x16 r[8], const float *gamma, const float *beta, int h,
                                           bf16x8 out[16]) {
    float s = 0.f;
    for (int j = 0; j < 8; ++j)
        for (int i = 0; i < 16; ++i) s += r[j][i];
    s += __shfl_xor(s, 32);
    const float mean = s * (1.0f / D);
    float v = 0.f;
    for (int j = 0; j < 8; ++j)
        for (int i = 0; i < 16; ++i) {
            const float d = r[j][i] - mean;
            v += d * d;
        }
    v += __shfl_xor(v, 32);
    const float rstd = rsqrtf(v * (1.0f / D) + 1e-5f);
    for (int j = 0; j < 8; ++j)
        for (int g = 0; g < 4; ++g) {
            const int f0 = 32 * j + 8 * g + 4 * h;
            const f32x4 gm = *reinterpret_cast<const f32x4 *>(gamma + f0);
            const f32x4 bt = *reinterpret_cast<const f32x4 *>(beta + f0);
            for (int q = 0; q < 4; ++q) {
                const int i = 4 * g + q;
                const float y = (r[j][i] - mean) * rstd * gm[q] + bt[q];
                out[2 * j + (i >> 3)][i & 7] = (__bf16)y;
            }
        }
}

constexpr int MODE_FULL = 0, MODE_HEAD = 1, MODE_TAIL = 2;
// HBM workspace of the two-kernel form, per board: K and V of the last layer as [head][key][lane half][16] bf16 (the 16
// values a lane half holds for one (token, head) are contiguous), then the CLS residual row (256 f32).
constexpr int64_t KV_ELEMS = (int64_t)NH * SEQ * HD;  // per board, per K or V

template <int MODE>
__global__ void __launch_bounds__(THREADS, 1)
k_encoder(const uint8_t *__restrict__ boards, const float *__restrict__ table, const float *__restrict__ cls,
          const __bf16 *__restrict__ wblob, const float *__restrict__ pblob, int n_layers,
          float *__restrict__ features, int64_t B, __bf16 *__restrict__ ws_k, __bf16 *__restrict__ ws_v,
          float *__restrict__ ws_r) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    Lds &L = *reinterpret_cast<Lds *>(smem);
    LdsAttn &A = L.act.a;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, r = lane & 31, h = lane >> 5;
    const int64_t board0 = (int64_t)blockIdx.x * (MODE == MODE_TAIL ? NTOK : NBOARD);
    char *const wq = L.w, *const wk = L.w + TILE, *const wv = L.w + 2 * TILE;
    LaneOff lo;
    lane_offsets(lo, r, h, w, lane);

    // ---- this lane's token.  MODE_TAIL: token = the CLS token of board board0 + tok.
    const int tok = 32 * w + r;                     // 0..127 inside the tile
    const int tb = MODE == MODE_TAIL ? tok : tok / SEQ;         // board in tile
    const int tc = MODE == MODE_TAIL ? 0 : tok - tb * SEQ;      // position (0 = CLS)
    const bool tok_real = MODE == MODE_TAIL ? true : tb < NBOARD;
    const bool tok_valid = tok_real && board0 + tb < B;
    const int64_t my_board = tok_valid ? board0 + tb : (B - 1);  // clamped: loads stay in bounds, stores are guarded

    // ---- embedding + positional code + CLS: R^T[f][tok]   (MODE_TAIL: the parked CLS residual row)
    f32x16 R[8];
    if (MODE == MODE_TAIL) {
        const float *src = ws_r + my_board * D;
        for (int j = 0; j < 8; ++j)
            for (int i = 0; i < 16; ++i) R[j][i] = src[32 * j + rowof(i, h)];
    } else {
        const float *src = cls;
        if (tok_valid && tc != 0) src = table + ((size_t)(tc - 1) * 31 + boards[(board0 + tb) * 16 + (tc - 1)]) * D;
        const float keep = tok_valid ? 1.0f : 0.0f;
        for (int j = 0; j < 8; ++j)
            for (int i = 0; i < 16; ++i) R[j][i] = keep * src[32 * j + rowof(i, h)];
    }

#pragma nounroll
    for (int layer = (MODE == MODE_TAIL ? n_layers - 1 : 0); layer < (MODE == MODE_HEAD ? n_layers - 1 : n_layers); ++layer) {
        const __bf16 *W = wblob + (size_t)layer * W_LAYER;
        const float *P = pblob + (size_t)layer * P_LAYER;
        auto dma_qkv = [&](int hd) {
            dma_tile<32>(wq, W + WO_QKV + (size_t)(0 * D + HD * hd) * D, D, HD, lo, w);
            if (MODE != MODE_TAIL) {
                dma_tile<32>(wk, W + WO_QKV + (size_t)(1 * D + HD * hd) * D, D, HD, lo, w);
                dma_tile<32>(wv, W + WO_QKV + (size_t)(2 * D + HD * hd) * D, D, HD, lo, w);
            }
        };
        auto wo_of = [&](int hd) -> char * { return L.w + (3 + (hd & 1)) * TILE; };  // double-buffered
        auto dma_wo = [&](int hd) { dma_tile<4>(wo_of(hd), W + WO_O + HD * hd, D, D, lo, w); };

        // ================= attention block =================
        __syncthreads();  // previous layer's feed-forward tiles (incl. the one aliasing `act`) are no longer read
        dma_qkv(0);
        dma_wo(0);
        stage_f32(L.gamma, P + PO_LN1G, D, tid);
        stage_f32(L.beta, P + PO_LN1B, D, tid);
        stage_f32(L.bias, P + PO_BQKV, 3 * D, tid);
        // zero the attention tiles: padded rows/columns are read (and multiplied by 0) but never written
        for (int i = tid; i < (int)sizeof(LdsAttn) / 16; i += THREADS)
            reinterpret_cast<uint4 *>(&A)[i] = make_uint4(0, 0, 0, 0);
        dma_wait_all();
        __syncthreads();
        bf16x8 xn[16];
        layer_norm(R, L.gamma, L.beta, h, xn);

#pragma nounroll
        for (int hd = 0; hd < NH; ++hd) {
            if (MODE == MODE_TAIL) {
                // Q^T [32 d][32 boards] of the CLS tokens; the single query of a board attends to that board's 17 keys
                // in-lane: this lane half holds 16 of the 32 head dims (d = rowof(i, h)), exactly the 16 contiguous
                // values MODE_HEAD parked per (board, head, key, half)
                const f32x16 zero = {0};
                f32x16 qa = gemm_tile<32, 16>(wq, lo, 0, xn, zero);
                pipe_mfma<16>();
                for (int i = 0; i < 16; ++i) qa[i] += L.bias[0 * D + HD * hd + rowof(i, h)];
                const size_t off = (size_t)my_board * KV_ELEMS + (size_t)hd * SEQ * HD + 16 * h;
                float sc[SEQ], m = -3.0e38f;
                for (int key = 0; key < SEQ; ++key) {
                    const bf16x8 k0 = *reinterpret_cast<const bf16x8 *>(ws_k + off + key * HD);
                    const bf16x8 k1 = *reinterpret_cast<const bf16x8 *>(ws_k + off + key * HD + 8);
                    float d = 0.f;
                    // q is rounded to bf16 like the operand of the full kernel's K Q^T product
                    for (int i = 0; i < 8; ++i) d += (float)(__bf16)qa[i] * (float)k0[i] + (float)(__bf16)qa[8 + i] * (float)k1[i];
                    d += __shfl_xor(d, 32);
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
                    const float p = (float)(__bf16)(sc[key] * inv);  // bf16 probabilities, as the MFMA operand would be
                    for (int i = 0; i < 8; ++i) {
                        o[i] += p * (float)v0[i];
                        o[8 + i] += p * (float)v1[i];
                    }
                }
                for (int g = 0; g < 4; ++g) {
                    bf16x4 ov;
                    for (int q = 0; q < 4; ++q) ov[q] = (__bf16)o[4 * g + q];
                    *reinterpret_cast<bf16x4 *>(A.o + tok * ST32 + 2 * (8 * g + 4 * h)) = ov;
                }
                __syncthreads();  // O visible; this head's Q tile is free
                if (hd + 1 < NH) {
                    dma_qkv(hd + 1);
                    dma_wo(hd + 1);
                }
                bf16x8 of[2];
                for (int ks = 0; ks < 2; ++ks) of[ks] = load_p(A.o + tok * ST32, ks, h);
                const char *wo = wo_of(hd);
                for (int j = 0; j < 8; ++j) R[j] = gemm_tile<4, 2>(wo, lo, j, of, R[j]);
                pipe_mfma<16>();
                dma_wait_all();
                __syncthreads();  // next head's tiles landed; O of this head is free
                continue;
            }
            // Q^T, K^T, V^T [32 d][32 tok] for this wave's tokens
            const f32x16 zero = {0};
            f32x16 qa = gemm_tile<32, 16>(wq, lo, 0, xn, zero);
            f32x16 ka = gemm_tile<32, 16>(wk, lo, 0, xn, zero);
            f32x16 va = gemm_tile<32, 16>(wv, lo, 0, xn, zero);
            pipe_mfma<48>();
            for (int g = 0; g < 4; ++g) {
                bf16x4 qv, kv;
                for (int q = 0; q < 4; ++q) {
                    const int i = 4 * g + q, d = rowof(i, h);
                    qv[q] = (__bf16)(qa[i] + L.bias[0 * D + HD * hd + d]);
                    kv[q] = (__bf16)(ka[i] + L.bias[1 * D + HD * hd + d]);
                    va[i] += L.bias[2 * D + HD * hd + d];
                }
                *reinterpret_cast<bf16x4 *>(A.q + tok * ST32 + 2 * (8 * g + 4 * h)) = qv;
                *reinterpret_cast<bf16x4 *>(A.k + tok * ST32 + 2 * (8 * g + 4 * h)) = kv;
            }
            if (tok_real) {
                char *vcol = A.vt + 2 * (32 * tb + kperm_pos(tc));
                for (int i = 0; i < 16; ++i) *reinterpret_cast<__bf16 *>(vcol + rowof(i, h) * STVT) = (__bf16)va[i];
            }
            __syncthreads();  // Q/K/V^T visible; in_proj tiles of this head are free
            if (hd + 1 < NH) {  // next head's weights land while this head's attention and out_proj run
                dma_qkv(hd + 1);
                dma_wo(hd + 1);
            }
            // attention of board b: S^T[key][query] = K Q^T, softmax over keys (registers), O^T = V^T P^T
            for (int b = w; b < NBOARD; b += 4) {
                const int t0 = SEQ * b;
                f32x16 s = {0};
                for (int ks = 0; ks < 2; ++ks)
                    s = mfma(load_p(A.k + (t0 + r) * ST32, ks, h), load_p(A.q + (t0 + r) * ST32, ks, h), s);
                float m = -3.0e38f;
                for (int i = 0; i < 16; ++i) {
                    s[i] = rowof(i, h) < SEQ ? s[i] * 0.17677669529663687f : -3.0e38f;
                    m = fmaxf(m, s[i]);
                }
                m = fmaxf(m, __shfl_xor(m, 32));
                float sum = 0.f;
                for (int i = 0; i < 16; ++i) {
                    s[i] = rowof(i, h) < SEQ ? __expf(s[i] - m) : 0.0f;
                    sum += s[i];
                }
                sum += __shfl_xor(sum, 32);
                const float inv = 1.0f / sum;
                for (int i = 0; i < 16; ++i) s[i] *= inv;
                bf16x8 pfr[2];
                frag_from_acc(s, pfr);
                f32x16 o = {0};
                for (int ks = 0; ks < 2; ++ks) o = mfma(load_p(A.vt + r * STVT + 2 * 32 * b, ks, h), pfr[ks], o);
                if (r < SEQ)
                    for (int g = 0; g < 4; ++g) {
                        bf16x4 ov;
                        for (int q = 0; q < 4; ++q) ov[q] = (__bf16)o[4 * g + q];
                        *reinterpret_cast<bf16x4 *>(A.o + (t0 + r) * ST32 + 2 * (8 * g + 4 * h)) = ov;
                    }
            }
            __syncthreads();  // O visible
            // R^T += Wo[:, head] . O^T
            bf16x8 of[2];
            for (int ks = 0; ks < 2; ++ks) of[ks] = load_p(A.o + tok * ST32, ks, h);
            const char *wo = wo_of(hd);
            for (int j = 0; j < 8; ++j) R[j] = gemm_tile<4, 2>(wo, lo, j, of, R[j]);
            pipe_mfma<16>();
            dma_wait_all();
            __syncthreads();  // next head's tiles landed; Q/K/V^T/O of this head are free
        }

        // ================= feed-forward block =================
        auto w1_of = [&](int c) -> char * { return L.w + ((c & 1) ? 2 * TILE : 0); };
        auto w2_of = [&](int c) -> char * { return (c & 1) ? L.act.w2b : L.w + 4 * TILE; };
        auto dma_ffn = [&](int c) {
            dma_tile<32>(w1_of(c), W + WO_1 + (size_t)(FFC * c) * D, D, FFC, lo, w);
            dma_tile<8>(w2_of(c), W + WO_2 + FFC * c, FF, D, lo, w);
        };
        dma_ffn(0);
        stage_f32(L.bias, P + PO_BO, D, tid);
        stage_f32(L.gamma, P + PO_LN2G, D, tid);
        stage_f32(L.beta, P + PO_LN2B, D, tid);
        __syncthreads();
        for (int j = 0; j < 8; ++j)
            for (int i = 0; i < 16; ++i) R[j][i] += L.bias[32 * j + rowof(i, h)];
        layer_norm(R, L.gamma, L.beta, h, xn);
        __syncthreads();
        stage_f32(L.bias, P + PO_B1, FF, tid);
        dma_wait_all();
        __syncthreads();
#pragma nounroll
        for (int c = 0; c < FF / FFC; ++c) {
            if (c + 1 < FF / FFC) dma_ffn(c + 1);  // lands while this stage computes
            const char *t1 = w1_of(c), *t2 = w2_of(c);
            const f32x16 zero = {0};
            f32x16 h0 = gemm_tile<32, 16>(t1, lo, 0, xn, zero);
            f32x16 h1 = gemm_tile<32, 16>(t1, lo, 1, xn, zero);
            pipe_mfma<32>();
            for (int i = 0; i < 16; ++i) {
                h0[i] = fmaxf(h0[i] + L.bias[FFC * c + rowof(i, h)], 0.0f);
                h1[i] = fmaxf(h1[i] + L.bias[FFC * c + 32 + rowof(i, h)], 0.0f);
            }
            bf16x8 hf[4];
            frag_from_acc(h0, hf);
            frag_from_acc(h1, hf + 2);
            for (int j = 0; j < 8; ++j) R[j] = gemm_tile<8, 4>(t2, lo, j, hf, R[j]);
            pipe_mfma<32>();
            dma_wait_all();
            __syncthreads();  // next stage landed; this stage's tiles are free
        }
        for (int j = 0; j < 8; ++j)
            for (int i = 0; i < 16; ++i) R[j][i] += P[PO_B2 + 32 * j + rowof(i, h)];
    }


    if (MODE == MODE_HEAD) {
        // ---- last layer, K/V only: LayerNorm, K^T and V^T of every head -> HBM, CLS residual row -> HBM
        const __bf16 *W = wblob + (size_t)(n_layers - 1) * W_LAYER;
        const float *P = pblob + (size_t)(n_layers - 1) * P_LAYER;
        auto dma_kv = [&](int hd) {
            dma_tile<32>(wk, W + WO_QKV + (size_t)(1 * D + HD * hd) * D, D, HD, lo, w);
            dma_tile<32>(wv, W + WO_QKV + (size_t)(2 * D + HD * hd) * D, D, HD, lo, w);
        };
        __syncthreads();  // previous layer's feed-forward tiles are no longer read
        dma_kv(0);
        stage_f32(L.gamma, P + PO_LN1G, D, tid);
        stage_f32(L.beta, P + PO_LN1B, D, tid);
        stage_f32(L.bias, P + PO_BQKV, 3 * D, tid);
        dma_wait_all();
        __syncthreads();
        if (tok_valid && tc == 0) {
            float *dst = ws_r + (board0 + tb) * D;
            for (int j = 0; j < 8; ++j)
                for (int i = 0; i < 16; ++i) dst[32 * j + rowof(i, h)] = R[j][i];
        }
        bf16x8 xn[16];
        layer_norm(R, L.gamma, L.beta, h, xn);
#pragma nounroll
        for (int hd = 0; hd < NH; ++hd) {
            const f32x16 zero = {0};
            f32x16 ka = gemm_tile<32, 16>(wk, lo, 0, xn, zero);
            f32x16 va = gemm_tile<32, 16>(wv, lo, 0, xn, zero);
            pipe_mfma<32>();
            if (tok_valid) {
                const size_t off = (size_t)(board0 + tb) * KV_ELEMS + ((size_t)hd * SEQ + tc) * HD + 16 * h;
                bf16x8 k0, k1, v0, v1;
                for (int i = 0; i < 8; ++i) {
                    k0[i] = (__bf16)(ka[i] + L.bias[1 * D + HD * hd + rowof(i, h)]);
                    k1[i] = (__bf16)(ka[8 + i] + L.bias[1 * D + HD * hd + rowof(8 + i, h)]);
                    v0[i] = (__bf16)(va[i] + L.bias[2 * D + HD * hd + rowof(i, h)]);
                    v1[i] = (__bf16)(va[8 + i] + L.bias[2 * D + HD * hd + rowof(8 + i, h)]);
                }
                *reinterpret_cast<bf16x8 *>(ws_k + off) = k0;
                *reinterpret_cast<bf16x8 *>(ws_k + off + 8) = k1;
                *reinterpret_cast<bf16x8 *>(ws_v + off) = v0;
                *reinterpret_cast<bf16x8 *>(ws_v + off + 8) = v1;
            }
            __syncthreads();  // this head's K/V weight tiles are free
            if (hd + 1 < NH) dma_kv(hd + 1);
            dma_wait_all();
            __syncthreads();
        }
        return;
    }

    // ---- CLS rows out (MODE_TAIL: every lane's token is one)
    if (tok_valid && tc == 0) {
        float *dst = features + (board0 + tb) * D;
        for (int j = 0; j < 8; ++j)
            for (int i = 0; i < 16; ++i) dst[32 * j + rowof(i, h)] = R[j][i];
    }
}

}  // namespace

extern "C" int64_t g2048_policy_encoder_workspace_bytes(int64_t B) {
    return B <= 0 ? 0 : B * (2 * KV_ELEMS * (int64_t)sizeof(__bf16) + D * (int64_t)sizeof(float));
}

extern "C" int g2048_policy_encoder(const uint8_t *boards, const float *embed_table, const float *cls_token,
                                    const void *weights_bf16, const float *params_f32, int n_layers,
                                    float *features, int64_t B, void *workspace, void *stream) {
    if (!boards || !embed_table || !cls_token || !weights_bf16 || !params_f32 || !features || n_layers < 1 || B <= 0)
        return G2048_EINVAL;
    if (((uintptr_t)weights_bf16 & 15) || ((uintptr_t)params_f32 & 15) || ((uintptr_t)embed_table & 3) ||
        ((uintptr_t)workspace & 15))
        return G2048_EINVAL;
    static bool attr_set = false;  // benign race: idempotent
    if (!attr_set) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(k_encoder<MODE_FULL>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)sizeof(Lds));
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(k_encoder<MODE_HEAD>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)sizeof(Lds));
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(k_encoder<MODE_TAIL>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)sizeof(Lds));
        attr_set = true;
    }
    static int dbg_set = 0;
    if (!dbg_set) {
        dbg_set = 1;
        if (getenv("G2048_ENCODER_DBG")) {
            const int v = atoi(getenv("G2048_ENCODER_DBG"));
            (void)hipMemcpyToSymbol(HIP_SYMBOL(g_dbg), &v, sizeof(int));
        }
    }
    const __bf16 *wb = reinterpret_cast<const __bf16 *>(weights_bf16);
    const unsigned blocks = (unsigned)((B + NBOARD - 1) / NBOARD);
    if (!workspace) {
        hipLaunchKernelGGL(k_encoder<MODE_FULL>, dim3(blocks), dim3(THREADS), sizeof(Lds), (hipStream_t)stream, boards, embed_table,
                           cls_token, wb, params_f32, n_layers, features, B, (__bf16 *)nullptr, (__bf16 *)nullptr, (float *)nullptr);
    } else {
        __bf16 *ws_k = reinterpret_cast<__bf16 *>(workspace), *ws_v = ws_k + B * KV_ELEMS;
        float *ws_r = reinterpret_cast<float *>(ws_v + B * KV_ELEMS);
        hipLaunchKernelGGL(k_encoder<MODE_HEAD>, dim3(blocks), dim3(THREADS), sizeof(Lds), (hipStream_t)stream, boards, embed_table,
                           cls_token, wb, params_f32, n_layers, features, B, ws_k, ws_v, ws_r);
        hipLaunchKernelGGL(k_encoder<MODE_TAIL>, dim3((unsigned)((B + NTOK - 1) / NTOK)), dim3(THREADS), sizeof(Lds), (hipStream_t)stream,
                           boards, embed_table, cls_token, wb, params_f32, n_layers, features, B, ws_k, ws_v, ws_r);
    }
    const hipError_t e = hipGetLastError();
    return e == hipSuccess ? 0 : -(1000 + (int)e);
}
