// Weight gradient of a Linear over the whole minibatch on gfx950: parts[s][n][k] = sum over the tokens t of slice s of dY[t][n] X[t][k].
//
// Reference: autograd of every nn.Linear of nn.TransformerEncoderLayer (src/ppo/transformer_encoder.py:138-148) under the update of
// src/ppo/ppo_trainer.py:409-437; in PyTorch one `dY^T @ X` per Linear, here until round 3 a 16-slice batched hipBLASLt GEMM
// (30-35 us per [1024 x 256] gradient at 34 816 tokens, 2.4-2.8 TB/s on exactly its algorithmic bytes).
//
// Both operands are token-major ([T][N] and [T][K]), and the reduction runs over the tokens: every MFMA operand wants 8 consecutive
// TOKENS of one column per lane.  The tiles go to LDS as they lie in memory (LDS-DMA, whole rows) and come out transposed with
// ds_read_b64_tr_b16; 16-byte chunk c of token row t sits at c ^ ((t & 3) << 2), which spreads the four rows a transposed read touches
// over all 64 banks.  A workgroup owns a [BN x 128] block of the gradient for one slice of the token axis; its eight waves hold the block
// in accumulator registers for the whole launch, three or four token stages of 64 rows are in LDS (counted vmcnt: the kernel only loads until
// its epilogue), the block leaves through LDS as full rows.  Blocks of the same token slice are dealt to the same XCD, so the re-reads
// of a token row by the other blocks hit that XCD's L2.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/g2048.h"

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int TOKS = 64, BK = 128;

__device__ __forceinline__ int rowof(int i, int h) { return (i & 3) + 8 * (i >> 2) + 4 * h; }

// one LDS-DMA wave-instruction, 16 bytes per lane (lane l lands at lds + 16 l), scalar row base + per-lane byte offset.  Inline assembly
// on purpose: for the builtin the compiler makes every later LDS read wait for the DMA (see g2048_linear.hip); the waits are explicit.
#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Winline-asm"
__device__ __forceinline__ void dma16(const void *sbase, uint32_t voff, uint32_t lds_addr) {
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(voff), "s"(sbase), "s"(lds_addr) : "memory", "m0");
}
#pragma clang diagnostic pop
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

__device__ __forceinline__ bf16x8 tr_pair(const char *p0, const char *p1) {
    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4 *)p0);
    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4 *)p1);
    const s16x8 v = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
    return __builtin_bit_cast(bf16x8, v);
}

// stage buffers in LDS: as many as fit, at most 4 (bn: rows of the gradient block = columns of the dY tile)
__host__ __device__ constexpr int dw_nbuf(int bn) { return (160 * 1024) / (TOKS * (bn + BK) * 2) > 4 ? 4 : (160 * 1024) / (TOKS * (bn + BK) * 2); }

// NTW: 32-row tiles of the n axis per wave; WN x 2 waves: block = 32 NTW WN rows x 128 columns.  Eight waves (two per SIMD) matter more
// than the tile shape: one wave alone issues an instruction every ~5 cycles, and per 16 MFMAs (512 cycles of the matrix pipe) a wave also
// issues 32 transposed reads, their address arithmetic and 8 x ~10 instructions of fetch bookkeeping.
template <int NTW, int WN>
__device__ __forceinline__ void dweight_block(const __bf16 *__restrict__ dy, int64_t lddy, const __bf16 *__restrict__ x, int64_t ldx,
                                              __bf16 *__restrict__ parts, float *__restrict__ colsum, int64_t T, int N, int K, int slices,
                                              int k_blocks, int block_id, bool parts_f32 = false) {
    constexpr int BN = 32 * NTW * WN, KTW = 2, NW = 2 * WN, THREADS = 64 * NW;
    constexpr int NBUF = dw_nbuf(BN), DIST = NBUF - 1;  // stage buffers; stages in flight ahead of the one being multiplied
    constexpr int ROWA = BN * 2, ROWB = BK * 2;            // LDS row bytes of the two tiles
    constexpr int ABYTES = TOKS * ROWA, BBYTES = TOKS * ROWB, STAGE = ABYTES + BBYTES;
    constexpr int RPI_A = 1024 / ROWA, RPI_B = 1024 / ROWB;  // token rows per DMA instruction
    constexpr int NI_A = TOKS / RPI_A / NW, NI_B = TOKS / RPI_B / NW, PER = NI_A + NI_B;  // instructions per wave and stage
    static_assert(NI_A >= 1 && NI_B >= 1 && (RPI_A * NW) % 4 == 0 && (RPI_B * NW) % 4 == 0, "fetch instructions per wave");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, r = lane & 31, h = lane >> 5;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6), wn = w >> 1, wk = w & 1;
    // workgroup -> (slice, block): the blocks of one token slice have equal ids modulo `slices`, a multiple of 8, so they share an XCD
    // (block_id = blockIdx.x minus the job's first workgroup, itself a multiple of 8)
    const int slice = block_id % slices, blk = block_id / slices, nb = blk / k_blocks, kb = blk % k_blocks;
    const int n0 = nb * BN, k0 = kb * BK;
    const int64_t per_slice = T / slices, tok_first = (int64_t)slice * per_slice;
    const int n_stages = (int)(per_slice / TOKS);
    const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) void *)smem;

    // ---- DMA addressing.  A: instruction i (this wave: i = w + NW j) covers token rows RPI_A i .. + RPI_A - 1; lane l -> row sub, physical
    // chunk pc = l % (ROWA / 16), logical chunk pc ^ ((row & 3) << 2).  (row & 3) = (RPI_A w + sub) & 3 for every j (NW RPI_A j = 0 mod 4).
    const int subA = lane / (ROWA / 16), pcA = lane % (ROWA / 16), subB = lane / (ROWB / 16), pcB = lane % (ROWB / 16);
    const uint32_t voffA = (uint32_t)(subA * lddy * 2) + 16u * (uint32_t)(pcA ^ (((RPI_A * w + subA) & 3) << 2));
    const uint32_t voffB = (uint32_t)(subB * ldx * 2) + 16u * (uint32_t)(pcB ^ (((RPI_B * w + subB) & 3) << 2));
    const char *const gA = reinterpret_cast<const char *>(dy + tok_first * lddy + n0);
    const char *const gB = reinterpret_cast<const char *>(x + tok_first * ldx + k0);
    // one of the PER fetch instructions of a stage (this wave's share), dealt out between the MFMAs of the stage before
    auto fetch_one = [&](int stage, int j) {
        const uint32_t buf = lds0 + (uint32_t)((stage % NBUF) * STAGE);
        const int i = w + NW * (j < NI_A ? j : j - NI_A);
        if (j < NI_A) dma16(gA + ((int64_t)stage * TOKS + RPI_A * i) * lddy * 2, voffA, buf + 1024u * i);
        else dma16(gB + ((int64_t)stage * TOKS + RPI_B * i) * ldx * 2, voffB, buf + ABYTES + 1024u * i);
    };
    auto fetch = [&](int stage) {
#pragma unroll
        for (int j = 0; j < PER; ++j) fetch_one(stage, j);
    };

    // ---- transposed reads.  Group of 16 lanes g2 = (lane >> 4) & 1 takes columns 16 g2 .. + 15 of a 32-column tile, lane 4 q + p of the
    // group supplies row q, columns 4 p .. + 3; the lane's operand = tokens 16 ks + 8 h + {0..3} (first read), + {4..7} (second).
    const int q = (lane >> 2) & 3, p = lane & 3, g2 = (lane >> 4) & 1;
    const int cl = 2 * g2 + (p >> 1);  // low two bits of the 16-byte chunk index inside the tile's 64 bytes
    const uint32_t laneA = (uint32_t)((8 * h + q) * ROWA + 8 * (p & 1) + 16 * cl), laneB = (uint32_t)((8 * h + q) * ROWB + 8 * (p & 1) + 16 * cl);
    uint32_t colA[NTW], colB[KTW];  // 64 * (tile ^ q): where the tile's 64-byte group of this lane's row sits
    for (int t = 0; t < NTW; ++t) colA[t] = laneA + 64u * (uint32_t)((NTW * wn + t) ^ q);
    for (int t = 0; t < KTW; ++t) colB[t] = laneB + 64u * (uint32_t)((KTW * wk + t) ^ q);

    // column sums of dY over the slice's tokens (the Linear's bias gradient) ride along in the workgroups of column block 0: a wave's
    // A operand already holds 8 tokens of one column per lane
    const bool do_cs = colsum != nullptr && kb == 0 && wk == 0;
    float cs[NTW];
    for (int t = 0; t < NTW; ++t) cs[t] = 0.f;
    f32x16 acc[NTW][KTW];
    for (int a = 0; a < NTW; ++a)
        for (int b = 0; b < KTW; ++b)
            for (int i = 0; i < 16; ++i) acc[a][b][i] = 0.f;

    auto frags = [&](const char *A, const char *B, int ks, bf16x8 (&fa)[NTW], bf16x8 (&fb)[KTW]) {
#pragma unroll
        for (int t = 0; t < NTW; ++t) fa[t] = tr_pair(A + colA[t] + (16 * ks) * ROWA, A + colA[t] + (16 * ks + 4) * ROWA);
#pragma unroll
        for (int t = 0; t < KTW; ++t) fb[t] = tr_pair(B + colB[t] + (16 * ks) * ROWB, B + colB[t] + (16 * ks + 4) * ROWB);
    };
    constexpr int KSTEPS = TOKS / 16, MPK = NTW * KTW;               // k-steps per stage, MFMAs per k-step
    constexpr int DMA_EVERY = (KSTEPS * MPK) / PER > 0 ? (KSTEPS * MPK) / PER : 1;  // one fetch instruction per DMA_EVERY MFMAs
    for (int d = 0; d < DIST; ++d)
        if (d < n_stages) fetch(d);
    for (int s = 0; s < n_stages; ++s) {
        // stage s has landed (the younger stages' instructions may stay in flight: this wave's queue holds loads only, retired in order)
        const int younger = n_stages - 1 - s < DIST - 1 ? n_stages - 1 - s : DIST - 1;
        if (DIST >= 3 && younger >= 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * PER) : "memory");
        else if (younger >= 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PER) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        lds_barrier();  // ... for every wave; and every wave is done with stage s - 1, whose buffer stage s + DIST takes
        const bool more = s + DIST < n_stages;
        const char *A = smem + (s % NBUF) * STAGE, *B = A + ABYTES;
        // every transposed read of the stage first, then the MFMAs as their operands arrive, with the fetch instructions of stage
        // s + DIST dealt out between them (a wave issues one 1 KiB LDS-DMA instruction per ~60 cycles: all at once they cost 500-700
        // cycles in front of every stage).  What bounds the loop are the transposed reads: ds_read_b64_tr_b16 moves 64 bytes per clock
        // and CU (half of ds_read_b128), and a wave tile of a x b MFMA tiles needs 2 (a + b) / (a b) of them per MFMA - 3 for the
        // 32 x 64 tile of the [128 x 128] block, 2 for the 64 x 64 tile of the [256 x 128] block (measured with the fetches switched
        // off: 22-23 us per [1024 x 256] gradient either way, 0.65 us per 64-token stage; reads one or two k-steps ahead of their
        // MFMAs instead: the same or slower; two groups of four waves that split a stage's k-steps over the whole block with 64 x 64
        // tiles, partial blocks added through LDS in the epilogue: 31.4 vs 29.6 us, 344 vs 314 us per minibatch)
        bf16x8 fa[KSTEPS][NTW], fb[KSTEPS][KTW];
#pragma unroll
        for (int ks = 0; ks < KSTEPS; ++ks) frags(A, B, ks, fa[ks], fb[ks]);
        if (do_cs) {
#pragma unroll
            for (int ks = 0; ks < KSTEPS; ++ks)
#pragma unroll
                for (int t = 0; t < NTW; ++t) {
                    const uint4 u = __builtin_bit_cast(uint4, fa[ks][t]);
                    cs[t] += (__uint_as_float(u.x << 16) + __uint_as_float(u.x & 0xFFFF0000u)) + (__uint_as_float(u.y << 16) + __uint_as_float(u.y & 0xFFFF0000u)) +
                             (__uint_as_float(u.z << 16) + __uint_as_float(u.z & 0xFFFF0000u)) + (__uint_as_float(u.w << 16) + __uint_as_float(u.w & 0xFFFF0000u));
                }
        }
#pragma unroll
        for (int ks = 0; ks < KSTEPS; ++ks) {
#pragma unroll
            for (int a = 0; a < NTW; ++a)
#pragma unroll
                for (int b = 0; b < KTW; ++b) {
                    const int m = ks * MPK + a * KTW + b;
                    acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[ks][a], fb[ks][b], acc[a][b], 0, 0, 0);
                    if (m % DMA_EVERY == DMA_EVERY - 1 && m / DMA_EVERY < PER && more) fetch_one(s + DIST, m / DMA_EVERY);
                }
        }
    }
    if (colsum != nullptr && kb == 0) {  // (wave-uniform; the shuffle needs every lane of the wave)
#pragma unroll
        for (int t = 0; t < NTW; ++t) {
            const float v = cs[t] + __shfl_xor(cs[t], 32);
            if (wk == 0 && h == 0) colsum[(int64_t)slice * N + n0 + 32 * (NTW * wn + t) + r] = v;
        }
    }
    // ---- epilogue: the block as bf16 through LDS (rows of 256 bytes), then full rows to parts[slice][n0 ..][k0 ..]
    lds_barrier();
    if (parts_f32) {  // (uniform) f32 partials: the A/B switch of round 4 (G2048_DWEIGHT_PARTS=f32x8); [BN][128] f32 <= the stage buffers
        float *const sm = reinterpret_cast<float *>(smem);
#pragma unroll
        for (int a = 0; a < NTW; ++a)
#pragma unroll
            for (int b = 0; b < KTW; ++b)
#pragma unroll
                for (int i = 0; i < 16; ++i) sm[(32 * (NTW * wn + a) + rowof(i, h)) * BK + 32 * (KTW * wk + b) + r] = acc[a][b][i];
        lds_barrier();
        float *const out32 = reinterpret_cast<float *>(parts) + ((int64_t)slice * N + n0) * K + k0;
        for (int e = tid; e < BN * (BK / 4); e += THREADS) {
            const int nl = e / (BK / 4), c = e % (BK / 4);
            *reinterpret_cast<uint4 *>(out32 + (int64_t)nl * K + 4 * c) = *reinterpret_cast<const uint4 *>(sm + nl * BK + 4 * c);
        }
        return;
    }
#pragma unroll
    for (int a = 0; a < NTW; ++a)
#pragma unroll
        for (int b = 0; b < KTW; ++b)
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int nl = 32 * (NTW * wn + a) + rowof(i, h), kl = 32 * (KTW * wk + b) + r;
                reinterpret_cast<__bf16 *>(smem)[nl * BK + kl] = (__bf16)acc[a][b][i];
            }
    lds_barrier();
    __bf16 *const out = parts + ((int64_t)slice * N + n0) * K + k0;
    for (int e = tid; e < BN * (BK / 8); e += THREADS) {
        const int nl = e / (BK / 8), c = e % (BK / 8);
        *reinterpret_cast<uint4 *>(out + (int64_t)nl * K + 8 * c) = *reinterpret_cast<const uint4 *>(smem + nl * ROWB + 16 * c);
    }
}

template <int NTW, int WN>
__global__ void __launch_bounds__(128 * WN, 1)
k_dweight(const __bf16 *__restrict__ dy, int64_t lddy, const __bf16 *__restrict__ x, int64_t ldx, __bf16 *__restrict__ parts,
          float *__restrict__ colsum, int64_t T, int N, int K, int slices, int k_blocks) {
    dweight_block<NTW, WN>(dy, lddy, x, ldx, parts, colsum, T, N, K, slices, k_blocks, (int)blockIdx.x);
}

// several products in one launch ([128 x 128] blocks): the weight gradients of a whole backward pass, deferred to its end by the caller
// (GradSink) - no ramp-up and drain per product, and the small ones (a [256 x 256] gradient is 128 workgroups) share the chip
struct DwJobs {
    g2048_dwg_job job[G2048_DWG_MAX_JOBS];
    int32_t first_block[G2048_DWG_MAX_JOBS + 1];
    int32_t n_jobs;
};
__global__ void __launch_bounds__(512, 1)
k_dweight_jobs(DwJobs J) {
    int j = 0;
    while (j + 1 < J.n_jobs && (int)blockIdx.x >= J.first_block[j + 1]) ++j;  // <= 16 entries, uniform
    const g2048_dwg_job &Q = J.job[j];
    dweight_block<1, 4>((const __bf16 *)Q.dy, Q.lddy, (const __bf16 *)Q.x, Q.ldx, (__bf16 *)Q.parts, Q.colsum, Q.T, Q.N, Q.K, Q.slices, Q.K / BK,
                        (int)blockIdx.x - J.first_block[j], Q.parts_f32 != 0);
}

}  // namespace

extern "C" int g2048_dweight_bf16(const void *dy, int64_t lddy, const void *x, int64_t ldx, void *parts, float *colsum, int64_t T, int N,
                                  int K, int slices, int block_rows, void *stream) {
    if (!dy || !x || !parts || T <= 0 || N < 128 || N % 128 || K < BK || K % BK || slices < 1 || (slices >= 8 && slices % 8) ||
        T % ((int64_t)TOKS * slices) || lddy < N || ldx < K || (lddy & 7) || (ldx & 7) ||
        (((uintptr_t)dy | (uintptr_t)x | (uintptr_t)parts) & 15) || ((uintptr_t)colsum & 3) || lddy * 2 * 4 >= (1ll << 31) || ldx * 2 * 4 >= (1ll << 31) ||
        (block_rows != 0 && block_rows != 128 && block_rows != 256) || (block_rows == 256 && N % 256))
        return G2048_EINVAL;
    const int k_blocks = K / BK;
    // rows of the gradient per workgroup: 256 when N allows (a [256 x 256] gradient: 128, twice the workgroups), or as asked
    const bool wide = block_rows ? block_rows == 256 : (N % 256 == 0 && (int64_t)N * K > 256 * 256);
    const int bn = wide ? 256 : 128;
    const dim3 grid((unsigned)(slices * (N / bn) * k_blocks));
    const void *fn = wide ? reinterpret_cast<const void *>(k_dweight<2, 4>) : reinterpret_cast<const void *>(k_dweight<1, 4>);
    const int lds = dw_nbuf(bn) * TOKS * (bn + BK) * 2;
    if (hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, lds) != hipSuccess) return -(1000 + (int)hipGetLastError());
    if (wide)
        hipLaunchKernelGGL((k_dweight<2, 4>), grid, dim3(512), lds, (hipStream_t)stream, (const __bf16 *)dy, lddy, (const __bf16 *)x, ldx,
                           (__bf16 *)parts, colsum, T, N, K, slices, k_blocks);
    else
        hipLaunchKernelGGL((k_dweight<1, 4>), grid, dim3(512), lds, (hipStream_t)stream, (const __bf16 *)dy, lddy, (const __bf16 *)x, ldx,
                           (__bf16 *)parts, colsum, T, N, K, slices, k_blocks);
    const hipError_t e = hipGetLastError();
    return e == hipSuccess ? 0 : -(1000 + (int)e);
}

static bool dwg_ok(const void *dy, int64_t lddy, const void *x, int64_t ldx, const void *parts, const float *colsum, int64_t T, int N, int K,
                   int slices) {
    return dy && x && parts && T > 0 && N >= 128 && N % 128 == 0 && K >= BK && K % BK == 0 && slices >= 1 && !(slices >= 8 && slices % 8) &&
           T % ((int64_t)TOKS * slices) == 0 && lddy >= N && ldx >= K && !(lddy & 7) && !(ldx & 7) &&
           !(((uintptr_t)dy | (uintptr_t)x | (uintptr_t)parts) & 15) && !((uintptr_t)colsum & 3) && lddy * 2 * 4 < (1ll << 31) &&
           ldx * 2 * 4 < (1ll << 31);
}

extern "C" int g2048_dweight_jobs(const g2048_dwg_job *jobs, int n_jobs, void *stream) {
    if (!jobs || n_jobs < 1 || n_jobs > G2048_DWG_MAX_JOBS) return G2048_EINVAL;
    DwJobs J;
    J.n_jobs = n_jobs;
    int64_t blocks = 0;
    for (int i = 0; i < n_jobs; ++i) {
        const g2048_dwg_job &q = jobs[i];
        // (a first workgroup that is a multiple of 8 keeps a job's token slices on their XCDs)
        if (!dwg_ok(q.dy, q.lddy, q.x, q.ldx, q.parts, q.colsum, q.T, q.N, q.K, q.slices) || q.slices % 8) return G2048_EINVAL;
        J.job[i] = q;
        J.first_block[i] = (int32_t)blocks;
        blocks += (int64_t)q.slices * (q.N / 128) * (q.K / BK);
    }
    J.first_block[n_jobs] = (int32_t)blocks;
    if (blocks > 65535 * 16) return G2048_EINVAL;
    const int lds = dw_nbuf(128) * TOKS * (128 + BK) * 2;
    if (hipFuncSetAttribute(reinterpret_cast<const void *>(k_dweight_jobs), hipFuncAttributeMaxDynamicSharedMemorySize, lds) != hipSuccess)
        return -(1000 + (int)hipGetLastError());
    hipLaunchKernelGGL(k_dweight_jobs, dim3((unsigned)blocks), dim3(512), lds, (hipStream_t)stream, J);
    const hipError_t e = hipGetLastError();
    return e == hipSuccess ? 0 : -(1000 + (int)e);
}
