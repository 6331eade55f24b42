// Y[T][N] = X[T][K] . W[N][K]^T (+ bias) in bf16 with f32 accumulation: nn.Linear for tall-skinny activations
// (T = 34 816 tokens per PPO minibatch, K and N in {256, 512, 768, 1024}) on gfx950 MFMA.
//
// hipBLASLt runs these shapes at 130-360 TFLOP/s (35 us for 34816x256x256, whose operands are 36 MB); they are really
// bandwidth problems with tiny weights.  Layout of this kernel:
//   * computed transposed, like the rollout encoder: Y^T[n][token] = W[n][k] . X^T[k][token] with
//     v_mfma_f32_32x32x16_bf16 - the weight tile is the A operand in nn.Linear's own [N][K] layout, tokens sit on lanes;
//   * a workgroup (4 waves) owns 128 tokens x a 128-wide slice of N; its weight slice streams L2 -> LDS by LDS-DMA in
//     K-chunks of 128 (32 KiB, XOR-swizzled so that every A fragment is one conflict-free ds_read_b128), two chunks in
//     flight; 64 KiB of LDS per workgroup = two workgroups per CU, so one's weight stream hides behind the other's MFMAs;
//   * the B operand (8 consecutive k of one token per lane) is loaded straight from global memory, 16 bytes per lane
//     per k-step - each lane walks its own row, the rows of a wave stay L1-resident across the K loop;
//   * epilogue: lanes l and l+32 exchange half their accumulators (v_permlane32_swap) so that every lane owns 64
//     consecutive outputs of its token = 128 contiguous bytes, adds the bias and stores bf16.
// The same kernel computes dX = dY . W with the transposed weight W^T[K][N] as its "weight".
// Two fused epilogues for the feed-forward block of the encoder layer (K <= 256 kernel only):
//   EPI_RELU_DROPOUT  y = dropout(relu(x W1^T + b1)) - linear1 forward without the bf16 round trip of the pre-activation;
//                     It can also leave one bit per output (non-zero or not) in a side buffer, in the layout of the
//                     kernel's accumulators: 64 bits per lane and tile.
//   EPI_MASK_COLSUM   dz = (dy W2) / keep where that bit is set, else 0, plus the column sums of dz (= linear1's bias
//                     gradient): the input gradient of linear2 fused with the backward of relu + dropout (replaces
//                     g2048_relu_dropout_bwd's pass over two [T][1024] matrices; the mask costs 1/16 of re-reading
//                     the activation).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/g2048.h"
#include "g2048_colsum_final.h"

namespace {

enum { EPI_NONE = 0, EPI_RELU_DROPOUT = 1, EPI_MASK_COLSUM = 2 };
struct Epi {
    float inv_keep;                 // 1 / (1 - p_drop)
    uint32_t thr16, s0, s1;         // EPI_RELU_DROPOUT: keep threshold on 16 hash bits, seed words
    const uint64_t *seed_state;     // optional device-resident word mixed into the seed (hipGraph replays)
    int64_t row_elems;              // elements per output row in the dropout index (= N)
    uint2 *bits;                    // EPI_RELU_DROPOUT (optional, written) / EPI_MASK_COLSUM (read): [tiles][N/128][256] x 64 bits
    float *partial;                 // [gridDim.x][N] column sums of this workgroup's tiles
    int N;
    uint32_t hi_term;               // EPI_RELU_DROPOUT: contribution of the seed's high word, constant per launch
};

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int TOK = 128, NS = 128, KC = 128, THREADS = 256;
constexpr int CHUNK_BYTES = NS * KC * 2;  // one K-chunk of the weight slice: 32 KiB
constexpr int NBUF = 2;

__device__ __forceinline__ uint32_t pack2(float a, float b) {
    const __bf16 x = (__bf16)a, y = (__bf16)b;
    return (uint32_t) * reinterpret_cast<const uint16_t *>(&x) | ((uint32_t) * reinterpret_cast<const uint16_t *>(&y) << 16);
}

// LDS image of a chunk: [128 rows n][16 chunks of 16 B], chunk q of row r stored at q ^ (r & 15).
// DMA: wave-instruction t of wave w fills LDS bytes [(4t + w) * 1024, +1024) = rows 4(4t + w) .. +3; lane i supplies
// physical chunk p = i & 15 of row 4(4t + w) + (i >> 4).
__device__ __forceinline__ void dma_chunk(char *dst, const __bf16 *w_slice, int64_t ldw, int kc, int w, int lane) {
    const char *base = reinterpret_cast<const char *>(w_slice) + (size_t)kc * KC * 2;
    for (int t = 0; t < NS / 16; ++t) {
        const int row = 4 * (4 * t + w) + (lane >> 4);
        const int q = (lane & 15) ^ (row & 15);
        const char *g = base + (size_t)row * ldw * 2 + q * 16;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)g,
                                         (__attribute__((address_space(3))) void *)(dst + (4 * t + w) * 1024), 16, 0, 0);
    }
}

// Shared epilogue.  acc[j][i] = Y^T[n0 + 32j + (i&3) + 8(i>>2) + 4h][token r]: a lane holds columns 8g + 4h .. +3 of every
// group g of 8.  Pack to bf16, then exchange between lane halves (v_permlane32_swap: lanes 32-63 of the first operand
// swap with lanes 0-31 of the second) so that the lower half owns the even groups and the upper half the odd ones,
// 8 consecutive outputs = one 16-byte store each.  Executed by all 64 lanes (the swap needs EXEC all ones).
template <bool HAS_BIAS, int EPI>
__device__ __forceinline__ void store_tile(f32x16 acc[4], const float *__restrict__ bias, __bf16 *yrow, int n0, int h, bool valid,
                                           const Epi &E, int64_t tok, uint2 *bits_slot, uint2 bits_in, float *colacc) {
    uint32_t obits[2] = {0u, 0u};  // bit 16 j + i of the pair: output (j, i) of this lane is non-zero
    // element index of output (j = 0, i = 0) halved: one hash covers two neighbouring columns.  32 bits are enough for the
    // index (a wrap only repeats masks after 2^33 outputs); the seed's high word enters through E.hi_term
    const uint32_t pair0 = (uint32_t)(((uint64_t)tok * (uint64_t)E.row_elems + (uint64_t)(n0 + 4 * h)) >> 1);
    for (int j = 0; j < 4; ++j) {
        if (HAS_BIAS)
            for (int g = 0; g < 4; ++g) {
                const float4 bv = *reinterpret_cast<const float4 *>(bias + n0 + 32 * j + 8 * g + 4 * h);
                acc[j][4 * g + 0] += bv.x; acc[j][4 * g + 1] += bv.y; acc[j][4 * g + 2] += bv.z; acc[j][4 * g + 3] += bv.w;
            }
        if (EPI == EPI_RELU_DROPOUT) {
            for (int i = 0; i < 16; i += 2) {
                float a = fmaxf(acc[j][i], 0.f), b = fmaxf(acc[j][i + 1], 0.f);
                if (E.thr16) {  // one 32-bit hash per pair of neighbouring columns, 16 bits each
                    uint32_t x = (pair0 + (uint32_t)(16 * j + ((i & 3) >> 1) + 4 * (i >> 2))) * 0x9E3779B1u + E.hi_term;
                    x ^= x >> 16; x *= 0x7FEB352Du; x ^= x >> 15; x *= 0x846CA68Bu; x ^= x >> 16;
                    a = (x & 0xFFFFu) >= E.thr16 ? a * E.inv_keep : 0.f;
                    b = (x >> 16) >= E.thr16 ? b * E.inv_keep : 0.f;
                }
                acc[j][i] = a; acc[j][i + 1] = b;
            }
        }
        if (EPI == EPI_MASK_COLSUM) {
            const uint32_t w = j < 2 ? bits_in.x : bits_in.y;
            for (int i = 0; i < 16; ++i) acc[j][i] = ((w >> (16 * (j & 1) + i)) & 1u) ? acc[j][i] * E.inv_keep : 0.f;
        }
        for (int m = 0; m < 2; ++m) {
            uint32_t ax = pack2(acc[j][8 * m + 0], acc[j][8 * m + 1]), ay = pack2(acc[j][8 * m + 2], acc[j][8 * m + 3]);
            uint32_t bx = pack2(acc[j][8 * m + 4], acc[j][8 * m + 5]), by = pack2(acc[j][8 * m + 6], acc[j][8 * m + 7]);
            if (EPI == EPI_RELU_DROPOUT) {  // non-zero AFTER rounding to bf16: exactly what the weight-gradient GEMM will read
                const uint32_t pk[4] = {ax, ay, bx, by};
                for (int q = 0; q < 4; ++q) {
                    const uint32_t nz = ((pk[q] & 0x7FFFu) ? 1u : 0u) | ((pk[q] & 0x7FFF0000u) ? 2u : 0u);
                    obits[j >> 1] |= nz << (16 * (j & 1) + 8 * m + 2 * q);
                }
            }
            const auto sx = __builtin_amdgcn_permlane32_swap(ax, bx, false, false);
            const auto sy = __builtin_amdgcn_permlane32_swap(ay, by, false, false);
            uint32_t o[4] = {sx[0], sy[0], sx[1], sy[1]};
            if (EPI == EPI_MASK_COLSUM) {
                for (int q = 0; q < 4; ++q) {
                    if (!valid) o[q] = 0u;
                    colacc[8 * (2 * j + m) + 2 * q] += __uint_as_float(o[q] << 16);
                    colacc[8 * (2 * j + m) + 2 * q + 1] += __uint_as_float(o[q] & 0xFFFF0000u);
                }
            }
            if (valid) *reinterpret_cast<uint4 *>(yrow + 32 * j + 16 * m) = make_uint4(o[0], o[1], o[2], o[3]);
        }
    }
    if (EPI == EPI_RELU_DROPOUT && bits_slot) *bits_slot = make_uint2(obits[0], obits[1]);
}

// K > 256: the weight slice streams through two LDS buffers once per 128-token tile.
template <bool HAS_BIAS>
__global__ void __launch_bounds__(THREADS, 2)
k_linear(const __bf16 *__restrict__ x, int64_t ldx, const __bf16 *__restrict__ wgt, int64_t ldw, const float *__restrict__ bias,
         __bf16 *__restrict__ y, int64_t ldy, int64_t T, int K) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, r = lane & 31, h = lane >> 5;
    const int64_t tok0 = (int64_t)blockIdx.x * TOK + 32 * w;
    const int n0 = blockIdx.y * NS;
    const __bf16 *w_slice = wgt + (size_t)n0 * ldw;
    const int n_chunks = K / KC;

    // this lane's token row (clamped: rows past T are computed on row T-1 and not stored)
    int64_t tok = tok0 + r;
    const bool valid = tok < T;
    if (!valid) tok = T - 1;
    const uint4 *xrow = reinterpret_cast<const uint4 *>(x + tok * ldx) + h;  // chunk 2ks + h of the row

    // vmcnt retires in issue order: the B operand of chunk c is always issued BEFORE the DMA of the chunk after it, so
    // "at most 8 outstanding" below means: everything up to this chunk landed, only the next chunk's DMA may be in flight
    uint4 bq[8];
    for (int ks = 0; ks < 8; ++ks) bq[ks] = xrow[2 * ks];
    dma_chunk(smem, w_slice, ldw, 0, w, lane);
    if (n_chunks > 1) dma_chunk(smem + CHUNK_BYTES, w_slice, ldw, 1, w, lane);

    f32x16 acc[4];
    for (int j = 0; j < 4; ++j)
        for (int i = 0; i < 16; ++i) acc[j][i] = 0.f;

    // A-fragment offsets inside a chunk image: row 32j + r, chunk (2ks + h) ^ (r & 15)
    int aoff[8];
    for (int ks = 0; ks < 8; ++ks) aoff[ks] = r * 256 + (((2 * ks + h) ^ (r & 15)) * 16);

    for (int c = 0; c < n_chunks; ++c) {
        // chunk c landed? (at most one younger chunk may still be in flight)
        if (c + 1 < n_chunks) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        const char *tile = smem + (c % NBUF) * CHUNK_BYTES;
        bf16x8 b[8];
        for (int ks = 0; ks < 8; ++ks) b[ks] = *reinterpret_cast<const bf16x8 *>(&bq[ks]);
        if (c + 1 < n_chunks)
            for (int ks = 0; ks < 8; ++ks) bq[ks] = xrow[2 * (8 * (c + 1) + ks)];  // next chunk's B operand
        for (int ks = 0; ks < 8; ++ks)
            for (int j = 0; j < 4; ++j) {
                const bf16x8 a = *reinterpret_cast<const bf16x8 *>(tile + aoff[ks] + j * 32 * 256);
                acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b[ks], acc[j], 0, 0, 0);
            }
        __syncthreads();  // every wave is done with this buffer
        if (c + NBUF < n_chunks) dma_chunk(smem + (c % NBUF) * CHUNK_BYTES, w_slice, ldw, c + NBUF, w, lane);
    }
    store_tile<HAS_BIAS, EPI_NONE>(acc, bias, y + (tok0 + r) * ldy + n0 + 8 * h, n0, h, valid, Epi{}, 0, nullptr, make_uint2(0u, 0u), nullptr);
}

// K <= 256, round 3: weights stationary in REGISTERS, tokens through LDS.
// Round 2's kernel kept the weight slice in LDS and let every lane walk its own token row in global memory (the B operand: 16 bytes
// per lane and k-step).  A wave-instruction of that kind is 64 separate 16-byte requests (lane = row); rocprofv3 showed the waves
// stalled at instruction issue for 58 % of their cycles (SQ_WAIT_INST_ANY) at 2.0-2.4 TB/s, and neither the instruction order nor the
// store pattern moved it (NOTES.md 3).  Here the roles are swapped:
//   * the waves of a workgroup (8, or 4 for widths that are not multiples of 256) split the slice's OUTPUT features, one 32-row weight
//     tile each: a wave's whole weight tile over K <= 256 is 16 fragments = 64 registers, loaded once per workgroup from nn.Linear's
//     row-major layout;
//   * 64-token tiles of X are copied into LDS by LDS-DMA (`global_load_lds`, whole 512-byte rows: perfectly coalesced, no registers),
//     double buffered, 16-byte chunks XOR-swizzled by row so that every B fragment is one conflict-free ds_read_b128; every wave
//     multiplies its weight tile against both 32-token blocks;
//   * the output tile goes through an LDS staging tile of its own (row-major, swizzled) and leaves as full rows of the slice.
// One workgroup per CU (2 x 32 KiB of X + 32 KiB of staging + bias + mask words); the order of fetch, MFMAs, epilogue, wait and stores
// inside the tile loop is what the comment in front of k_linear_ws is about.
// Accumulator tile of wave w, token block b: acc[b][i] = Y^T[n0 + 32 w + rowof(i, h)][token 32 b + r].
__device__ __forceinline__ int rowof(int i, int h) { return (i & 3) + 8 * (i >> 2) + 4 * h; }

// One LDS-DMA wave-instruction (BYTES per lane, lane l lands at lds + BYTES l), written as inline assembly ON PURPOSE: for the builtin
// the compiler's wait-count pass makes every later LDS read of the wave wait for the DMA (it cannot tell the buffers of one dynamic
// LDS array apart), i.e. `s_waitcnt vmcnt(0)` right after the fetch that is meant to stay in flight for a whole tile.  The waits for
// these fetches are therefore all explicit (`s_waitcnt vmcnt(0)` + barrier in k_linear_ws).  Unknown to the compiler, they can only
// make ITS counted waits longer, never shorter (vmcnt retires in order and they are younger than what it waits for or it waits for 0).
// (m0 is "reserved" for the compiler; it writes it only right in front of its own LDS-DMA builtins, which this kernel does not use)
#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Winline-asm"
template <int BYTES>
__device__ __forceinline__ void dma_async(const void *g, void *lds) {
    const uint32_t l = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)(__attribute__((address_space(3))) void *)lds);
    if (BYTES == 16) asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(g), "s"(l) : "memory", "m0");
    else asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dword %0, off" ::"v"(g), "s"(l) : "memory", "m0");
}

// the same with a scalar base and a 32-bit per-lane offset: no vector arithmetic per instruction
__device__ __forceinline__ void dma_async16(const void *sbase, uint32_t voff, void *lds) {
    const uint32_t l = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)(__attribute__((address_space(3))) void *)lds);
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(voff), "s"(sbase), "s"(l) : "memory", "m0");
}
#pragma clang diagnostic pop

// X tile [64 tokens][K] -> LDS rows of K * 2 bytes, chunk q of row r at q ^ (r & 15); K = 256: 32 wave-instructions of 1 KiB (2 rows
// each), dealt to the workgroup's NF waves: wave f issues instructions f, f + NF, ...
constexpr int TOKW = 64;  // tokens per tile of k_linear_ws
template <int K, int NF>
struct XTileDma {
    static constexpr int CPR = K / 8, RPI = 64 / CPR, N_INST = TOKW / RPI;  // 16-byte chunks per row, rows per instruction
    static constexpr int NVAR = 16 / RPI / NF > 0 ? 16 / RPI / NF : 1;       // distinct swizzles among ONE wave's instructions
    static_assert(N_INST % NF == 0, "every wave issues the same number of fetch instructions");
    uint32_t voff[NVAR];  // per-lane byte offset of this fetcher's instruction j (j % NVAR decides the swizzle): lane's row * ldx * 2 + 16 q
    int sub, p;
    __device__ __forceinline__ void init(int64_t ldx, int lane, int f) {
        sub = lane / CPR, p = lane % CPR;
        for (int v = 0; v < NVAR; ++v)
            voff[v] = (uint32_t)(sub * ldx * 2) + 16u * (uint32_t)(p ^ ((RPI * (NF * v + f) + sub) & 15));
    }
    // whole tile inside [0, T): scalar row base per instruction, nothing on the vector ALU
    __device__ __forceinline__ void full(char *dst, const __bf16 *x, int64_t ldx, int64_t tok0, int f) const {
        const char *base = reinterpret_cast<const char *>(x + tok0 * ldx);
#pragma unroll
        for (int j = 0; j < N_INST / NF; ++j) {
            const int n = NF * j + f;
            dma_async16(base + (int64_t)n * RPI * ldx * 2, voff[j % NVAR], dst + n * 1024);
        }
    }
    // last, partial tile: rows past T read row T - 1 (computed and not stored)
    __device__ __forceinline__ void clamped(char *dst, const __bf16 *x, int64_t ldx, int64_t tok0, int64_t T, int f) const {
        for (int j = 0; j < N_INST / NF; ++j) {
            const int n = NF * j + f;
            const int row = n * RPI + sub;
            int64_t tok = tok0 + row;
            if (tok >= T) tok = T - 1;
            dma_async<16>(reinterpret_cast<const char *>(x + tok * ldx) + (p ^ (row & 15)) * 16, dst + n * 1024);
        }
    }
};

// Diagnostic build only (-DG2048_WS_STAMPS, tools/stamps_linear.py; never compiled into the product library): cycle stamps at the phase
// boundaries of k_linear_ws, summed per phase by waves 0 and 1 of every 16th workgroup row of slice 0.
#ifdef G2048_WS_STAMPS
constexpr int WS_PHASES = 8;
__device__ unsigned long long g_ws_stamps[2][WS_PHASES];
struct WsStamps {
    unsigned long long last, acc[WS_PHASES];
    __device__ __forceinline__ static unsigned long long now() {
        unsigned long long t;
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
        __builtin_amdgcn_sched_barrier(0);
        return t;
    }
    __device__ __forceinline__ void start() {
        for (int i = 0; i < WS_PHASES; ++i) acc[i] = 0;
        last = now();
    }
    __device__ __forceinline__ void mark(int k) {
        const unsigned long long t = now();
        acc[k] += t - last;
        last = t;
    }
    __device__ __forceinline__ void flush(int lane, int w) {
        if (lane == 0 && w < 2 && blockIdx.y == 0 && blockIdx.x % 16 == 0)
            for (int i = 0; i < WS_PHASES; ++i) atomicAdd(&g_ws_stamps[w == 0 ? 0 : 1][i], acc[i]);
    }
};
#define WS_STAMP(k) stamps.mark(k)
#else
#define WS_STAMP(k)
#endif

// workgroup barrier that waits for this wave's LDS traffic only: __syncthreads() also drains vmcnt, i.e. would make the storing waves
// wait for their global stores at every barrier
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// The loop of a workgroup over its 64-token tiles (round 3, final form).  A wave's vmcnt counts loads AND stores, so "wait for my
// fetch" right after "store my share of the previous tile" waits for the stores too (measured: double buffering with the wait at the top
// of the tile changed nothing).  The order that avoids it - every wave does the same:
//   barrier 1                      tile i is in X buffer p (published by barrier 3 of the previous tile)
//   issue my fetch instructions of tile i + 1 into buffer p ^ 1 (LDS-DMA: X rows, mask words)
//   bias -> accumulators, 2 KS MFMAs against buffer p (B fragments 8 reads ahead)
//   epilogue -> staging tile (a buffer of its own: with the X buffer doubling as staging tile a third barrier sat here, 29.0 -> 28.0 us)
//   s_waitcnt vmcnt(0)             my fetch (issued ~2 000 cycles ago) AND my stores of tile i - 1 (~3 000 cycles ago): both old
//   barrier 3                      staging tile complete, tile i + 1 published
//   copy my share of the staging tile to global memory: full row segments, 16 bytes per lane; never waited for here
// Forms measured on the way (N = 1024, tools/stamps_linear.py for the cycles per tile):
//   * dedicated fetching (2) and storing (2..4) waves: the coupling is gone as well, but a wave issues one 1 KiB vector-memory
//     instruction per ~60 cycles (eight waves together one per ~20), so the 1 000 cycles of fetch issue and the 1 500 of copy-out of those
//     few waves sat on the tile's critical path: 28.2-28.9 us, the same as this form, with more code;
//   * the same with a third X buffer and the fetch issued while the storing waves copy out: 30.3 us (fetches and stores contend for
//     the CU's one texture addresser; overlapping them makes both slower);
//   * a private staging tile per wave and ONE barrier per tile, so that the waves of a SIMD drift apart and one's MFMAs run under the
//     other's epilogue: 39.8 us - the 64-byte row segments a single wave can store are half cache lines.
// What is left (stamps): of ~5 000 cycles per tile the MFMA phase takes 2 170 for 1 024 cycles of MFMA per wave - both waves of a SIMD
// are in it at the same time and the matrix pipe idles through the other phases (epilogue 630, copy-out 750, fetch issue 450,
// barriers ~900).  Token block 0's epilogue runs in the shadow of block 1's MFMAs (hand-cut steps with sched_barrier fences as in the
// rollout encoder; with sched_group_barrier masks the solver dropped the whole pipeline): 120.9 -> 111.6 us per minibatch for the
// ReLU + dropout variant, 103.5 -> 101.9 for the masked backward, +-0 for the plain one (same box).  Hashing block 1's keep bits during
// block 0's MFMAs as well: no further gain.
// Two independent workgroups per CU would interleave the phases, but need <= 128 registers per wave; the weight tile alone is 64.
// WAVES = 8: a 256-wide slice of N per workgroup - the X tile goes through the CU's vector-memory path once per 256 outputs instead of
// once per 128 (30.9 -> 28.9 us); WAVES = 4: 128-wide, for widths that are not multiples of 256.  One workgroup per CU either way.
template <bool HAS_BIAS, int EPI, int KS, int WAVES>  // KS = K / 16: 16 (K = 256) or 8 (K = 128)
__global__ void __launch_bounds__(64 * WAVES, 1)
k_linear_ws(const __bf16 *__restrict__ x, int64_t ldx, const __bf16 *__restrict__ wgt, int64_t ldw, const float *__restrict__ bias,
            __bf16 *__restrict__ y, int64_t ldy, int64_t T, Epi E) {
    constexpr int THREADS = 64 * WAVES, NSW = 32 * WAVES;  // slice width
    constexpr int CPR_Y = NSW / 8;                          // 16-byte chunks per staging row
    constexpr int K = 16 * KS;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, r = lane & 31, h = lane >> 5;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int n0 = blockIdx.y * NSW;
    constexpr int row_bytes = K * 2, xbytes = TOKW * row_bytes, sbytes = TOKW * 2 * NSW;

    // ---- this wave's weight tile: rows n0 + 32 w .. + 31, all of K (fragment ks = columns 16 ks + 8 h .. + 7 of row .. + r)
    bf16x8 wf[KS];
    {
        const __bf16 *wrow = wgt + (size_t)(n0 + 32 * w + r) * ldw + 8 * h;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) wf[ks] = *reinterpret_cast<const bf16x8 *>(wrow + 16 * ks);
    }
    // Two X buffers: the fetch of tile i + 1 is issued right after the barrier that publishes tile i.  (Issuing it later, during the
    // copy-out of tile i - with a third buffer to keep the lead - measured SLOWER, 30.3 vs 28.5 us at N = 1024: fetches and stores go
    // through the same texture addresser.)
    constexpr int NB = 2;
    // LDS: the X buffers, the output staging tile, the slice's bias (it enters through the accumulators' initial value, 4 broadcast
    // reads per tile), mask words
    char *const stage = smem + NB * xbytes;
    float *const bias_l = reinterpret_cast<float *>(smem + NB * xbytes + sbytes);
    uint32_t *const bits_l = reinterpret_cast<uint32_t *>(smem + NB * xbytes + sbytes + NSW * 4);  // [NB][THREADS]
    for (int i = tid; i < NSW; i += THREADS) bias_l[i] = HAS_BIAS ? bias[n0 + i] : 0.f;
    if (EPI == EPI_RELU_DROPOUT) {
        if (E.seed_state) {
            const uint64_t sd = *E.seed_state;  // same mixing as the other dropout kernels (g2048_layernorm.hip)
            E.s0 ^= (uint32_t)sd * 0x9E3779B1u;
            E.s1 += (uint32_t)(sd >> 32) * 0x85EBCA77u + (uint32_t)sd;
        }
        E.hi_term = E.s0 ^ (E.s1 * 0x85EBCA77u);
    }
    const int slices = gridDim.y;
    float colacc[EPI == EPI_MASK_COLSUM ? 16 : 1];  // per-lane sums over this workgroup's tokens of the lane's output rows
    for (int q = 0; q < (EPI == EPI_MASK_COLSUM ? 16 : 1); ++q) colacc[q] = 0.f;
    const int64_t n_tiles64 = (T + TOKW - 1) / TOKW;
    // one 32-bit word per thread and 64-token tile: bit 16 b + i <-> accumulator (b, i) of this lane
    auto bits_ptr = [&](int64_t tile) -> uint32_t * {
        return reinterpret_cast<uint32_t *>(E.bits) + (tile * slices + blockIdx.y) * THREADS;
    };
    XTileDma<K, WAVES> dma;
    dma.init(ldx, lane, w);
    auto fetch = [&](int64_t tile, int buf) {  // this wave's share
        if ((tile + 1) * TOKW <= T) dma.full(smem + buf * xbytes, x, ldx, tile * TOKW, w);
        else dma.clamped(smem + buf * xbytes, x, ldx, tile * TOKW, T, w);
        if (EPI == EPI_MASK_COLSUM) {
            const uint32_t *g = bits_ptr(tile) + lane;
            dma_async<4>(g + 64 * w, bits_l + buf * THREADS + 64 * w);
        }
    };
    if ((int64_t)blockIdx.x < n_tiles64) fetch(blockIdx.x, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // (the first barrier 1 publishes the first tile)
#ifdef G2048_WS_STAMPS
    WsStamps stamps;
    stamps.start();
#endif
    int buf = 0;
    for (int64_t tile = blockIdx.x; tile < n_tiles64; tile += gridDim.x, buf ^= 1) {
        const int64_t tok0 = tile * TOKW;
        const char *const xb = smem + buf * xbytes;
        lds_barrier();  // every wave is done with the staging tile (its reads for the copy-out of the previous tile)
        WS_STAMP(1);
        if (tile + gridDim.x < n_tiles64) fetch(tile + gridDim.x, buf ^ 1);
        WS_STAMP(2);
        const uint32_t bits_in = EPI == EPI_MASK_COLSUM ? bits_l[buf * THREADS + tid] : 0u;
        f32x16 acc[2];
        for (int g = 0; g < 4; ++g) {
            const float4 bv = *reinterpret_cast<const float4 *>(bias_l + 32 * w + 8 * g + 4 * h);
            for (int b = 0; b < 2; ++b) {
                acc[b][4 * g + 0] = bv.x; acc[b][4 * g + 1] = bv.y; acc[b][4 * g + 2] = bv.z; acc[b][4 * g + 3] = bv.w;
            }
        }
        // ---- epilogue pieces.  Lane = token 32 b + r, registers 4g..4g+3 = output features n0 + 32 w + 8g + 4h .. +3.
        const int nw = n0 + 32 * w;
        uint32_t obits = 0u;  // bit 16 b + i: output (b, i) of this lane is non-zero after rounding
        // ReLU + dropout of the pair of columns (i, i + 1), i = 2 pi: one 32-bit hash per pair, 16 bits each (index = element index / 2,
        // the seed's high word through E.hi_term: the convention of the round-2 kernel); p = 0: thr16 = 0 keeps everything
        auto keep_pair = [&](int pi, uint32_t pair0) -> uint32_t {  // bit 0 / 1: column i / i + 1 is kept
            const int i = 2 * pi;
            uint32_t xh = (pair0 + (uint32_t)(((i & 3) >> 1) + 4 * (i >> 2))) * 0x9E3779B1u + E.hi_term;
            xh ^= xh >> 16; xh *= 0x7FEB352Du; xh ^= xh >> 15; xh *= 0x846CA68Bu; xh ^= xh >> 16;
            return ((xh & 0xFFFFu) >= E.thr16 ? 1u : 0u) | ((xh >> 16) >= E.thr16 ? 2u : 0u);
        };
        auto drop_pair = [&](int b, int pi, uint32_t keep) {
            const int i = 2 * pi;
            const float a = fmaxf(acc[b][i], 0.f), c = fmaxf(acc[b][i + 1], 0.f);
            acc[b][i] = (keep & 1u) ? a * E.inv_keep : 0.f;
            acc[b][i + 1] = (keep & 2u) ? c * E.inv_keep : 0.f;
        };
        // the four outputs of group g: forward's mask (masked backward), rounding, non-zero bits, column sums, staging tile
        auto finish_g = [&](int b, int g, bool valid) {
            if (EPI == EPI_MASK_COLSUM) {
                const uint32_t wbits = bits_in >> (16 * b + 4 * g);
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[b][4 * g + j] = ((wbits >> j) & 1u) ? acc[b][4 * g + j] * E.inv_keep : 0.f;
            }
            const uint32_t lo = pack2(acc[b][4 * g + 0], acc[b][4 * g + 1]), hi = pack2(acc[b][4 * g + 2], acc[b][4 * g + 3]);
            if (EPI == EPI_RELU_DROPOUT) {
                const uint32_t nz = ((lo & 0x7FFFu) ? 1u : 0u) | ((lo & 0x7FFF0000u) ? 2u : 0u) | ((hi & 0x7FFFu) ? 4u : 0u) |
                                    ((hi & 0x7FFF0000u) ? 8u : 0u);
                obits |= nz << (16 * b + 4 * g);
            }
            if (EPI == EPI_MASK_COLSUM) {  // what at::sum over the bf16 tensor would add (rows past T add nothing)
                colacc[4 * g + 0] += valid ? __uint_as_float(lo << 16) : 0.f;
                colacc[4 * g + 1] += valid ? __uint_as_float(lo & 0xFFFF0000u) : 0.f;
                colacc[4 * g + 2] += valid ? __uint_as_float(hi << 16) : 0.f;
                colacc[4 * g + 3] += valid ? __uint_as_float(hi & 0xFFFF0000u) : 0.f;
            }
            // staging tile: rows of 2 NSW bytes, 16-byte chunk c of row q at c ^ (q & (CPR_Y - 1)); this piece = half a chunk
            const int trow = 32 * b + r, c = 4 * w + g;
            *reinterpret_cast<uint2 *>(stage + trow * (2 * NSW) + ((c ^ (trow & (CPR_Y - 1))) * 16) + 8 * h) = make_uint2(lo, hi);
        };
        const bool valid0 = tok0 + r < T, valid1 = tok0 + 32 + r < T;
        uint32_t pair0[2];
        for (int b = 0; b < 2; ++b)
            pair0[b] = (uint32_t)(((uint64_t)((b ? valid1 : valid0) ? tok0 + 32 * b + r : T - 1) * (uint64_t)E.row_elems + (uint64_t)(nw + 4 * h)) >> 1);
        // ---- 2 KS steps of one MFMA each, in program order (sched_barrier between them: left alone the compiler issues the MFMAs back to
        // back and the vector work after them).  A step = the MFMA, the B fragment PRE steps ahead, and - during token block 1 - a slice
        // of token block 0's epilogue, which then runs in the shadow of the matrix pipe (both waves of a SIMD are in this phase together
        // and keep the pipe saturated; the vector ALU is idle otherwise).
        {
            constexpr int PRE = 8, N = 2 * KS;  // step s: block s / KS, k-step s % KS
            bf16x8 q[PRE];
            auto frag = [&](int s) {
                const int b = s / KS, ks = s % KS;
                return *reinterpret_cast<const bf16x8 *>(xb + (32 * b + r) * row_bytes + (((2 * ks + h) ^ (r & 15)) * 16));
            };
#pragma unroll
            for (int s = 0; s < PRE; ++s) q[s] = frag(s);
#pragma unroll
            for (int s = 0; s < N; ++s) {
                __builtin_amdgcn_sched_barrier(0);
                const int b = s / KS, ks = s % KS;
                acc[b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[ks], q[s % PRE], acc[b], 0, 0, 0);
                if (s + PRE < N) q[s % PRE] = frag(s + PRE);
                constexpr int SPG = KS / 4;  // steps per output group
                const int g = ks / SPG, t = ks % SPG;
                if (b == 1) {  // slice ks of block 0's epilogue (KS = 16: pairs at steps 4g and 4g + 2, the group's outputs at 4g + 3)
                    if (EPI == EPI_RELU_DROPOUT) {
                        if (SPG >= 4) {
                            if (t == 0) drop_pair(0, 2 * g, keep_pair(2 * g, pair0[0]));
                            if (t == 2) drop_pair(0, 2 * g + 1, keep_pair(2 * g + 1, pair0[0]));
                        } else if (t == 0) {
                            drop_pair(0, 2 * g, keep_pair(2 * g, pair0[0]));
                            drop_pair(0, 2 * g + 1, keep_pair(2 * g + 1, pair0[0]));
                        }
                    }
                    if (t == SPG - 1) finish_g(0, g, valid0);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        WS_STAMP(3);
        // ---- token block 1's epilogue
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            if (EPI == EPI_RELU_DROPOUT) {
                drop_pair(1, 2 * g, keep_pair(2 * g, pair0[1]));
                drop_pair(1, 2 * g + 1, keep_pair(2 * g + 1, pair0[1]));
            }
            finish_g(1, g, valid1);
        }
        if (EPI == EPI_RELU_DROPOUT && E.bits) bits_ptr(tile)[tid] = obits;
        WS_STAMP(5);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // my part of the next tile has landed; my stores of the previous tile are out
        WS_STAMP(0);
        lds_barrier();  // the output tile is complete, the next X tile is published
        WS_STAMP(6);
        {
            constexpr int PIECES = TOKW * CPR_Y, STORERS = THREADS, ROUNDS = PIECES / STORERS;
            static_assert(PIECES % STORERS == 0, "every lane moves the same number of 16-byte pieces");
            const int e0 = tid;
            uint4 v[ROUNDS];
#pragma unroll
            for (int i = 0; i < ROUNDS; ++i) {  // all LDS reads first, then the stores
                const int e = e0 + STORERS * i;
                v[i] = *reinterpret_cast<const uint4 *>(stage + (e / CPR_Y) * (2 * NSW) + (e % CPR_Y) * 16);
            }
#pragma unroll
            for (int i = 0; i < ROUNDS; ++i) {
                const int e = e0 + STORERS * i;
                const int trow = e / CPR_Y, p = e % CPR_Y, c = p ^ (trow & (CPR_Y - 1));
                if (tok0 + trow < T) *reinterpret_cast<uint4 *>(y + (tok0 + trow) * ldy + n0 + 8 * c) = v[i];
            }
        }
        WS_STAMP(7);
        // (the next iteration's barrier separates these reads of the staging tile from the fetch of tile i + 2 into this buffer)
    }
#ifdef G2048_WS_STAMPS
    stamps.flush(lane, w);
#endif
    if (EPI == EPI_MASK_COLSUM) {
        // column sums: register q of lane (r, h) = output feature n0 + 32 w + rowof(q, h) over this lane's tokens
        for (int q = 0; q < 16; ++q) {
            float v = colacc[q];
            for (int m = 16; m >= 1; m >>= 1) v += __shfl_xor(v, m);
            if (r == 0) E.partial[(int64_t)blockIdx.x * E.N + n0 + 32 * w + rowof(q, h)] = v;
        }
    }
}

}  // namespace

namespace {

// workgroup rows of k_linear_ws: 512 resident workgroups shared between the N-slices.  XCD locality: workgroup (x, y) has linear id
// x + y * groups and the dispatcher deals linear ids round-robin over the 8 XCDs, so the N-slices y of one token tile x share an XCD
// (and its L2 copy of the X rows) only if groups % 8 == 0.  Measured in round 2 with N = 768 (groups 85): 98.8 MB of HBM reads per
// launch for a 17.8 MB X (rocprofv3 FETCH_SIZE).
inline int ws_waves(int N) { return N % 256 == 0 ? 8 : 4; }
inline int64_t ws_groups(int64_t T, int N) {
    const int waves = ws_waves(N);
    const int64_t n_tiles = (T + TOKW - 1) / TOKW;
    int64_t groups = 256 / (N / (32 * waves));
    if (groups < 1) groups = 1;
    if (groups >= 8) groups -= groups % 8;
    return groups > n_tiles ? n_tiles : groups;
}

inline bool operands_ok(const void *x, int64_t ldx, const void *weight, int64_t ldw, const void *y, int64_t ldy, int64_t T, int K,
                        int N, const void *bias) {
    return x && weight && y && T > 0 && K >= KC && K % KC == 0 && N >= NS && N % NS == 0 && ldx >= K && ldw >= K && ldy >= N &&
           !(ldx & 7) && !(ldw & 7) && !(ldy & 7) && !(((uintptr_t)x | (uintptr_t)weight | (uintptr_t)y | (uintptr_t)bias) & 15);
}

// K <= 256: weights-stationary kernel; 256-wide slices (8 waves, one workgroup per CU) when N allows, else 128-wide (4 waves, two per CU)
template <bool HAS_BIAS, int EPI, int KS, int WAVES>
int launch_ws(const __bf16 *x, int64_t ldx, const __bf16 *w, int64_t ldw, const float *bias, __bf16 *y, int64_t ldy, int64_t T, int N,
              const Epi &E, int64_t groups, hipStream_t stream) {
    const void *fn = reinterpret_cast<const void *>(k_linear_ws<HAS_BIAS, EPI, KS, WAVES>);
    // two X buffers + the output staging tile + bias + mask words
    const int lds = 2 * TOKW * 16 * KS * 2 + TOKW * 2 * 32 * WAVES + 32 * WAVES * 4 + 2 * 64 * WAVES * 4;
    // the dynamic-LDS limit is a per-device attribute of the one kernel this call launches: set per call, no latch
    if (hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, lds) != hipSuccess) return -(1000 + (int)hipGetLastError());
    hipLaunchKernelGGL((k_linear_ws<HAS_BIAS, EPI, KS, WAVES>), dim3((unsigned)groups, (unsigned)(N / (32 * WAVES))), dim3(64 * WAVES), lds,
                       stream, x, ldx, w, ldw, bias, y, ldy, T, E);
    const hipError_t e = hipGetLastError();
    return e == hipSuccess ? 0 : -(1000 + (int)e);
}
template <bool HAS_BIAS, int EPI>
int launch_stationary(const __bf16 *x, int64_t ldx, const __bf16 *w, int64_t ldw, const float *bias, __bf16 *y, int64_t ldy, int64_t T,
                      int K, int N, const Epi &E, int *groups_out, hipStream_t stream) {
    if (K != 256 && K != 128) return G2048_EINVAL;
    const int64_t groups = ws_groups(T, N);
    if (groups_out) *groups_out = (int)groups;
    const bool wide = ws_waves(N) == 8;
    if (K == 256)
        return wide ? launch_ws<HAS_BIAS, EPI, 16, 8>(x, ldx, w, ldw, bias, y, ldy, T, N, E, groups, stream)
                    : launch_ws<HAS_BIAS, EPI, 16, 4>(x, ldx, w, ldw, bias, y, ldy, T, N, E, groups, stream);
    return wide ? launch_ws<HAS_BIAS, EPI, 8, 8>(x, ldx, w, ldw, bias, y, ldy, T, N, E, groups, stream)
                : launch_ws<HAS_BIAS, EPI, 8, 4>(x, ldx, w, ldw, bias, y, ldy, T, N, E, groups, stream);
}

}  // namespace

extern "C" int g2048_linear_bf16(const void *x, int64_t ldx, const void *weight, int64_t ldw, const float *bias, void *y,
                                 int64_t ldy, int64_t T, int K, int N, void *stream) {
    if (!operands_ok(x, ldx, weight, ldw, y, ldy, T, K, N, bias)) return G2048_EINVAL;
    const __bf16 *xp = (const __bf16 *)x, *wp = (const __bf16 *)weight;
    __bf16 *yp = (__bf16 *)y;
    if (K <= NBUF * KC)
        return bias ? launch_stationary<true, EPI_NONE>(xp, ldx, wp, ldw, bias, yp, ldy, T, K, N, Epi{}, nullptr, (hipStream_t)stream)
                    : launch_stationary<false, EPI_NONE>(xp, ldx, wp, ldw, bias, yp, ldy, T, K, N, Epi{}, nullptr, (hipStream_t)stream);
    const void *fn = bias ? reinterpret_cast<const void *>(k_linear<true>) : reinterpret_cast<const void *>(k_linear<false>);
    if (hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, NBUF * CHUNK_BYTES) != hipSuccess)
        return -(1000 + (int)hipGetLastError());
    const dim3 grid((unsigned)((T + TOK - 1) / TOK), (unsigned)(N / NS));
    if (bias)
        hipLaunchKernelGGL(k_linear<true>, grid, dim3(THREADS), NBUF * CHUNK_BYTES, (hipStream_t)stream, xp, ldx, wp, ldw, bias, yp, ldy, T, K);
    else
        hipLaunchKernelGGL(k_linear<false>, grid, dim3(THREADS), NBUF * CHUNK_BYTES, (hipStream_t)stream, xp, ldx, wp, ldw, bias, yp, ldy, T,
                           K);
    const hipError_t e = hipGetLastError();
    return e == hipSuccess ? 0 : -(1000 + (int)e);
}

extern "C" int64_t g2048_ffn_mask_bytes(int64_t T, int N) {
    return (T <= 0 || N < NS || N % NS) ? 0 : ((T + TOK - 1) / TOK) * (N / NS) * THREADS * (int64_t)sizeof(uint2);
}

extern "C" int g2048_linear_relu_dropout_bf16(const void *x, int64_t ldx, const void *weight, int64_t ldw, const float *bias, void *y,
                                              int64_t ldy, int64_t T, int K, int N, float p_drop, uint64_t seed,
                                              const uint64_t *seed_state, void *mask_bits, void *stream) {
    if (!operands_ok(x, ldx, weight, ldw, y, ldy, T, K, N, bias) || !bias || K > NBUF * KC || !(p_drop >= 0.f && p_drop < 1.f))
        return G2048_EINVAL;
    Epi E{};
    E.inv_keep = 1.0f / (1.0f - p_drop);
    E.thr16 = (uint32_t)(p_drop * 65536.0f + 0.5f);
    E.s0 = (uint32_t)seed; E.s1 = (uint32_t)(seed >> 32);
    E.seed_state = p_drop > 0.f ? seed_state : nullptr;
    E.row_elems = N;
    E.bits = (uint2 *)mask_bits;
    if ((uintptr_t)mask_bits & 7) return G2048_EINVAL;
    return launch_stationary<true, EPI_RELU_DROPOUT>((const __bf16 *)x, ldx, (const __bf16 *)weight, ldw, bias, (__bf16 *)y, ldy, T, K, N,
                                                     E, nullptr, (hipStream_t)stream);
}

extern "C" int64_t g2048_linear_mask_bwd_workspace_floats(int64_t T, int N) {
    return (T <= 0 || N < NS || N % NS) ? 0 : (int64_t)512 * N;  // at most 512 / (N / 128) workgroup rows of N partial sums
}

extern "C" int64_t g2048_linear_mask_bwd_partial_rows(int64_t T, int N) {
    if (T <= 0 || N < NS || N % NS) return 0;
    return ws_groups(T, N);
}

extern "C" int g2048_linear_mask_bwd_bf16(const void *dy, int64_t lddy, const void *weight_t, int64_t ldw, const void *mask_bits,
                                          void *dz, int64_t lddz, float *dbias, float *workspace, int64_t T, int K, int N,
                                          float p_drop, void *stream) {
    if (!operands_ok(dy, lddy, weight_t, ldw, dz, lddz, T, K, N, nullptr) || K > NBUF * KC || !mask_bits ||
        ((uintptr_t)mask_bits & 7) || !workspace || ((uintptr_t)workspace & 15) || !(p_drop >= 0.f && p_drop < 1.f))
        return G2048_EINVAL;
    Epi E{};
    E.inv_keep = 1.0f / (1.0f - p_drop);
    E.bits = (uint2 *)const_cast<void *>(mask_bits);
    E.partial = workspace;
    E.N = N;
    int groups = 0;
    const int rc = launch_stationary<false, EPI_MASK_COLSUM>((const __bf16 *)dy, lddy, (const __bf16 *)weight_t, ldw, nullptr, (__bf16 *)dz,
                                                             lddz, T, K, N, E, &groups, (hipStream_t)stream);
    if (rc || !dbias) return rc;  // dbias NULL: the partial rows stay in the workspace for g2048_reduce_jobs
    hipLaunchKernelGGL(k_colsum_final, dim3((unsigned)((N + CF_COLS - 1) / CF_COLS)), dim3(CF_COLS * CF_SLICES), 0, (hipStream_t)stream,
                       workspace, groups, N, dbias);
    const hipError_t e = hipGetLastError();
    return e == hipSuccess ? 0 : -(1000 + (int)e);
}

#ifdef G2048_WS_STAMPS
extern "C" int g2048_debug_ws_stamps(unsigned long long *out /*host [2][8]*/, int reset) {
    if (hipMemcpyFromSymbol(out, HIP_SYMBOL(g_ws_stamps), sizeof(g_ws_stamps)) != hipSuccess) return -1;
    if (reset) {
        static const unsigned long long zero[2][WS_PHASES] = {};
        if (hipMemcpyToSymbol(HIP_SYMBOL(g_ws_stamps), zero, sizeof(zero)) != hipSuccess) return -1;
    }
    return 0;
}
#endif
