// The MLP policy's update (BASELINE.json configs[1]: 4 096 boards, flattened one-hot -> 512 -> 512 trunk + the reference's actor / critic
// heads, src/ppo/ppo_agent.py:72-87 of the reference for the heads) as a handful of launches instead of ~45.
//
// At minibatch 2048 every kernel of this policy is a few microseconds of work, and a node of the replayed hipGraph costs ~4.5 us whatever
// it does (profiles/round4_mlp_update_timeline.txt): the update is bound by its NODE COUNT.  A whole-network kernel is no way out - every
// workgroup would stream all 3.1 MB of weights through one CU's memory path (>= 50 us per pass: the fused CLS tail's floor) - so the
// layers stay separate launches, but each launch does everything that belongs to its layer:
//   k_mlp_embed_fwd   trunk_in on packed boards: one-hot x W^T is a sum of 16 weight columns; + bias + ReLU; also leaves the one-hot
//                     matrix (bf16 [M][512], a column of ones at 496) that the grouped weight-gradient launch multiplies with
//   k_gemm_jobs       a table of small GEMMs per launch, y = epi(sum_s x_s . w_s^T): up to two K-segments per job (the actor's and the
//                     critic's first layers back-propagate into the SAME trunk gradient), bias + ReLU (forward) or the ReLU mask of a
//                     saved activation (backward) in the epilogue, several jobs per launch (the two heads' layers side by side)
//   k_mlp_out_fwd     both heads' output layers (4 logits + 1 value per row) on the vector ALU
//   k_mlp_out_bwd     their input gradients with the ReLU mask, and the first stage of their weight gradients
// The five 512-wide weight gradients, their bias gradients and the trunk_in gradient (x = the one-hot matrix) go through
// g2048_dweight_jobs / g2048_reduce_jobs like the Transformer's.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/g2048.h"
#include "g2048_mfma.h"

namespace {

using namespace g2048_mfma;

__device__ __forceinline__ float bf2f(uint32_t hi16) { return __uint_as_float(hi16 << 16); }
__device__ __forceinline__ uint32_t f2bf(float f) {
    const __bf16 b = (__bf16)f;
    return *reinterpret_cast<const uint16_t *>(&b);
}
inline int mlp_done() {
    const hipError_t e = hipGetLastError();
    return e == hipSuccess ? 0 : -(1000 + (int)e);
}

// ---- trunk_in forward ---------------------------------------------------------------------------------------------------
// wt: bf16 [496][512] = the TRANSPOSE of trunk_in.weight [512][496] (class-major: row 31 c + v is what cell c holding exponent v adds);
// one wavefront per board row, 8 output features per lane (one 16-byte load per table row: 1 KB per wave-instruction)
constexpr int ME_D = 512, ME_CELLS = 16, ME_CLASSES = 31;

__global__ void __launch_bounds__(256)
k_mlp_embed_fwd(const uint8_t *__restrict__ boards, const uint16_t *__restrict__ wt, const float *__restrict__ bias,
                uint16_t *__restrict__ y, uint16_t *__restrict__ onehot, int64_t M) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= M) return;
    const uint4 bq = *reinterpret_cast<const uint4 *>(boards + row * 16);
    const uint32_t bw[4] = {bq.x, bq.y, bq.z, bq.w};
    int idx[ME_CELLS];
#pragma unroll
    for (int c = 0; c < ME_CELLS; ++c) {
        const int v = (int)((bw[c >> 2] >> (8 * (c & 3))) & 0xFFu);
        idx[c] = ME_CLASSES * c + (v < ME_CLASSES ? v : ME_CLASSES - 1);  // (the env never exceeds 17)
    }
    uint4 t[ME_CELLS];
#pragma unroll
    for (int c = 0; c < ME_CELLS; ++c) t[c] = reinterpret_cast<const uint4 *>(wt + (size_t)idx[c] * ME_D)[lane];
    float acc[8];
    const float4 b0 = reinterpret_cast<const float4 *>(bias)[2 * lane], b1 = reinterpret_cast<const float4 *>(bias)[2 * lane + 1];
    acc[0] = b0.x; acc[1] = b0.y; acc[2] = b0.z; acc[3] = b0.w; acc[4] = b1.x; acc[5] = b1.y; acc[6] = b1.z; acc[7] = b1.w;
#pragma unroll
    for (int c = 0; c < ME_CELLS; ++c) {  // ascending cells: a fixed summation order
        const uint32_t w[4] = {t[c].x, t[c].y, t[c].z, t[c].w};
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            acc[2 * q] += bf2f(w[q] & 0xFFFFu);
            acc[2 * q + 1] += bf2f(w[q] >> 16);
        }
    }
    uint32_t o[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) o[q] = f2bf(fmaxf(acc[2 * q], 0.f)) | (f2bf(fmaxf(acc[2 * q + 1], 0.f)) << 16);
    reinterpret_cast<uint4 *>(y + row * ME_D)[lane] = make_uint4(o[0], o[1], o[2], o[3]);
    if (onehot) {  // elements 8 lane .. 8 lane + 7 of the row's one-hot vector; column 496 = 1 (onehot^T dY then carries the column sums
                   // of dY = trunk_in's bias gradient in its row 496), columns 497.. stay zero
        uint32_t h[4] = {lane == (ME_CELLS * ME_CLASSES) / 8 ? 0x3F80u : 0u, 0u, 0u, 0u};
#pragma unroll
        for (int c = 0; c < ME_CELLS; ++c)
            if ((idx[c] >> 3) == lane) h[(idx[c] & 7) >> 1] |= 0x3F80u << (16 * (idx[c] & 1));  // bf16 1.0
        reinterpret_cast<uint4 *>(onehot + row * ME_D)[lane] = make_uint4(h[0], h[1], h[2], h[3]);
    }
}

// ---- a table of small GEMMs ------------------------------------------------------------------------------------------------
// Workgroup (4 waves) = one [64 rows x 64 outputs] tile of one job; computed transposed (Y^T = W X^T: the weight tile is the A operand
// in nn.Linear's [N][K] layout, rows sit on lanes); K in chunks of 64 through two LDS stages (register staging: everything visible to
// the compiler's wait counts); the output tile leaves through LDS as full 128-byte row segments.  32 KB of LDS: four or five workgroups
// share a CU and hide each other's round trips.  (Measured and dropped, round 4: the whole K extent of both tiles requested at once by
// LDS-DMA, one wait, one barrier - 128 KB of LDS, one workgroup per CU: 57.8 instead of 52.8 us for the six launches of a minibatch.  What
// a launch costs is the 128 KB every 64 x 64 tile pulls through its CU's memory path, and co-resident small workgroups keep that path
// busier than one large one.)
constexpr int GJ_TM = 64, GJ_TN = 64, GJ_KC = 64, GJ_THREADS = 256;
constexpr int GJ_TILE = GJ_TM * GJ_KC * 2;  // 8 KB: one operand tile, rows of 128 bytes, 16-byte piece p of row r at p ^ ((r >> 1) & 7)
// ((r >> 1): two consecutive 128-byte rows fill the 64 banks once, so rows r and r + 2 must not share a piece slot)

struct GemmJobs {
    g2048_gemm_job job[G2048_GEMM_MAX_JOBS];
    int32_t first_tile[G2048_GEMM_MAX_JOBS + 1];  // n-tiles of job j: first_tile[j] .. first_tile[j + 1]
    int32_t n_jobs;
    int64_t M;
};

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

__global__ void __launch_bounds__(GJ_THREADS)
k_gemm_jobs(GemmJobs J) {
    __shared__ __attribute__((aligned(16))) char smem[2 * 2 * GJ_TILE];  // [stage][x | w]
    const int tid = threadIdx.x, lane = tid & 63, r = lane & 31, h = lane >> 5;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6), wn = w >> 1, wm = w & 1;
    int j = 0;
    while (j + 1 < J.n_jobs && (int)blockIdx.y >= J.first_tile[j + 1]) ++j;  // (<= 8 entries, uniform)
    const g2048_gemm_job &Q = J.job[j];
    const int n0 = ((int)blockIdx.y - J.first_tile[j]) * GJ_TN;
    const int64_t m0 = (int64_t)blockIdx.x * GJ_TM;
    // this thread's two 16-byte pieces of each operand tile: piece e = tid + 256 i -> tile row e / 8, piece e % 8
    const int prow = tid >> 3, pp = tid & 7;
    int64_t xrow[2];
    int wrow[2];
    uint32_t loff[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int rr = prow + 32 * i;
        xrow[i] = m0 + rr < J.M ? m0 + rr : J.M - 1;  // rows past M are computed on the last row and not stored
        wrow[i] = n0 + rr;
        loff[i] = (uint32_t)(rr * 128 + ((pp ^ ((rr >> 1) & 7)) << 4));
    }
    const int ch0 = Q.k[0] / GJ_KC, n_chunks = ch0 + Q.k[1] / GJ_KC;
    auto load = [&](int c, u32x4 (&gx)[2], u32x4 (&gw)[2]) {
        const int s = c < ch0 ? 0 : 1, kc = (c < ch0 ? c : c - ch0) * GJ_KC;
        const __bf16 *xs = (const __bf16 *)Q.x[s], *ws = (const __bf16 *)Q.w[s];
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            gx[i] = *reinterpret_cast<const u32x4 *>(xs + xrow[i] * Q.ldx[s] + kc + 8 * pp);
            gw[i] = *reinterpret_cast<const u32x4 *>(ws + (int64_t)wrow[i] * Q.ldw[s] + kc + 8 * pp);
        }
    };
    auto store = [&](int stage, const u32x4 (&gx)[2], const u32x4 (&gw)[2]) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            *reinterpret_cast<u32x4 *>(smem + stage * 2 * GJ_TILE + loff[i]) = gx[i];
            *reinterpret_cast<u32x4 *>(smem + stage * 2 * GJ_TILE + GJ_TILE + loff[i]) = gw[i];
        }
    };
    f32x16 acc = zero_tile();
    if (Q.bias) acc = bias_tile(Q.bias, n0 + 32 * wn, h);
    u32x4 gx[2], gw[2];
    load(0, gx, gw);
    store(0, gx, gw);
    if (n_chunks > 1) load(1, gx, gw);
    lds_barrier();
    // fragment offsets: row 32 wn + r (weight) / 32 wm + r (x), piece (2 kk + h) ^ ((row >> 1) & 7)
    const uint32_t aoff = (uint32_t)((32 * wn + r) * 128), boff = (uint32_t)((32 * wm + r) * 128), sw = (uint32_t)((r >> 1) & 7);
    for (int c = 0; c < n_chunks; ++c) {
        const char *xs = smem + (c & 1) * 2 * GJ_TILE, *ws = xs + GJ_TILE;
        bf16x8 a[4], b[4];
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            a[kk] = *reinterpret_cast<const bf16x8 *>(ws + aoff + (((2 * kk + h) ^ sw) << 4));
            b[kk] = *reinterpret_cast<const bf16x8 *>(xs + boff + (((2 * kk + h) ^ sw) << 4));
        }
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) acc = mfma(a[kk], b[kk], acc);
        if (c + 1 < n_chunks) {
            store((c + 1) & 1, gx, gw);  // (the other stage: last read in iteration c - 1, every wave is past that barrier)
            if (c + 2 < n_chunks) load(c + 2, gx, gw);
        }
        lds_barrier();
    }
    // ---- epilogue: acc[i] = Y^T[n0 + 32 wn + rowof(i, h)][m0 + 32 wm + r]; ReLU or the mask of a saved activation; tile -> LDS (rows of
    // 128 bytes = 64 outputs, the operand tiles' swizzle) -> full row segments
    const int trow = 32 * wm + r;
    const int64_t mrow = m0 + trow < J.M ? m0 + trow : J.M - 1;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        float v[4] = {acc[4 * g], acc[4 * g + 1], acc[4 * g + 2], acc[4 * g + 3]};
        const int nl = 32 * wn + 8 * g + 4 * h;  // first of this lane's 4 consecutive outputs, inside the tile
        if (Q.relu)
            for (int q = 0; q < 4; ++q) v[q] = fmaxf(v[q], 0.f);
        if (Q.act) {
            const uint2 am = *reinterpret_cast<const uint2 *>((const uint16_t *)Q.act + mrow * Q.ldact + n0 + nl);
            const uint32_t aw[2] = {am.x, am.y};
            for (int q = 0; q < 4; ++q) v[q] = ((aw[q >> 1] >> (16 * (q & 1))) & 0x7FFFu) ? v[q] : 0.f;  // (activations are >= 0: > 0 == non-zero)
        }
        const uint32_t lo = f2bf(v[0]) | (f2bf(v[1]) << 16), hi = f2bf(v[2]) | (f2bf(v[3]) << 16);
        *reinterpret_cast<uint2 *>(smem + trow * 128 + ((((nl >> 3)) ^ ((trow >> 1) & 7)) << 4) + 8 * ((nl >> 2) & 1)) = make_uint2(lo, hi);
    }
    lds_barrier();
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int rr = prow + 32 * i;
        if (m0 + rr < J.M)
            *reinterpret_cast<u32x4 *>((uint16_t *)Q.y + (m0 + rr) * Q.ldy + n0 + 8 * pp) = *reinterpret_cast<const u32x4 *>(smem + loff[i]);
    }
}

// ---- the heads' output layers -------------------------------------------------------------------------------------------------
// h2: bf16 [M][1024] = actor's second hidden layer | critic's; w3: bf16 [5][512] = actor.4.weight (4 rows) then critic.4.weight (1 row).
constexpr int MO_H = 512;

__global__ void __launch_bounds__(256)
k_mlp_out_fwd(const uint16_t *__restrict__ h2, const uint16_t *__restrict__ w3, float *__restrict__ logits, float *__restrict__ values,
              int64_t M) {
    __shared__ float wl[5 * MO_H];
    for (int i = threadIdx.x; i < 5 * MO_H; i += 256) wl[i] = bf2f(w3[i]);
    __syncthreads();
    // 16 lanes per row: lane part p takes elements p + 16 i of each 512-wide half (coalesced 32-byte runs per half-row)
    const int part = threadIdx.x & 15;
    const int64_t row = (int64_t)blockIdx.x * 16 + (threadIdx.x >> 4);
    const int64_t rc = row < M ? row : M - 1;
    float s[5] = {0.f, 0.f, 0.f, 0.f, 0.f};
    const uint16_t *hr = h2 + rc * (2 * MO_H);
    for (int i = 0; i < MO_H / 16; ++i) {
        const int e = part + 16 * i;
        const float a = bf2f(hr[e]), c = bf2f(hr[MO_H + e]);
        s[0] = __builtin_fmaf(a, wl[e], s[0]);
        s[1] = __builtin_fmaf(a, wl[MO_H + e], s[1]);
        s[2] = __builtin_fmaf(a, wl[2 * MO_H + e], s[2]);
        s[3] = __builtin_fmaf(a, wl[3 * MO_H + e], s[3]);
        s[4] = __builtin_fmaf(c, wl[4 * MO_H + e], s[4]);
    }
#pragma unroll
    for (int o = 0; o < 5; ++o)
        for (int m = 8; m >= 1; m >>= 1) s[o] += __shfl_xor(s[o], m);
    if (part == 0 && row < M) {
        reinterpret_cast<float4 *>(logits)[row] = make_float4(s[0], s[1], s[2], s[3]);
        values[row] = s[4];
    }
}

// d h2 = [W_a3^T d logits | W_c3^T d value] where h2 > 0 (bf16 [M][1024]), and this workgroup's share of the output layers' weight
// gradients: partial[blockIdx][5][512] f32 = sum over its 8 rows of d out[row][o] * h2[row][half(o)][:]  (fixed order; summed by
// g2048_reduce_jobs).  Thread t owns columns 2t, 2t + 1 of each 512-wide half (4-byte loads and stores, 256 contiguous bytes per wave);
// 8 rows per workgroup = 256 workgroups at minibatch 2048 (32 rows: 64 workgroups, 13 us of serial 2-byte accesses - now ~6).
constexpr int MO_ROWS = 8;

__global__ void __launch_bounds__(256)
k_mlp_out_bwd(const float *__restrict__ dlogits, const float *__restrict__ dvalues, const uint16_t *__restrict__ h2,
              const uint16_t *__restrict__ w3, uint16_t *__restrict__ dh2, float *__restrict__ partial, int64_t M) {
    __shared__ float dl[MO_ROWS][5];
    const int64_t r0 = (int64_t)blockIdx.x * MO_ROWS;
    const int t = threadIdx.x;
    uint32_t hq[MO_ROWS][2];  // this thread's pair of columns of every row, both halves: all loads in flight before the first use
#pragma unroll
    for (int rr = 0; rr < MO_ROWS; ++rr) {
        const int64_t row = r0 + rr < M ? r0 + rr : M - 1;
        const uint32_t *hr = reinterpret_cast<const uint32_t *>(h2 + row * (2 * MO_H));
        hq[rr][0] = hr[t];
        hq[rr][1] = hr[MO_H / 2 + t];
    }
    if (t < MO_ROWS * 5) {
        const int rr = t / 5, o = t % 5;
        const int64_t row = r0 + rr;
        // (rounded to bf16, as the unfused path hands the gradients to its bf16 GEMMs)
        dl[rr][o] = row < M ? bf2f(f2bf(o < 4 ? dlogits[row * 4 + o] : dvalues[row])) : 0.f;
    }
    float wv[2][5], pw[2][5];  // [column of the pair][output]
#pragma unroll
    for (int o = 0; o < 5; ++o) {
        const uint32_t w = reinterpret_cast<const uint32_t *>(w3 + o * MO_H)[t];
        wv[0][o] = bf2f(w & 0xFFFFu), wv[1][o] = bf2f(w >> 16);
        pw[0][o] = pw[1][o] = 0.f;
    }
    __syncthreads();
#pragma unroll
    for (int rr = 0; rr < MO_ROWS; ++rr) {  // ascending rows: a fixed summation order
        const int64_t row = r0 + rr;
        if (row >= M) break;
        uint32_t oa = 0u, oc = 0u;
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            const uint32_t ha = (hq[rr][0] >> (16 * c)) & 0xFFFFu, hc = (hq[rr][1] >> (16 * c)) & 0xFFFFu;
            const float fa = bf2f(ha), fc = bf2f(hc);
            float ga = 0.f;
#pragma unroll
            for (int o = 0; o < 4; ++o) {
                ga = __builtin_fmaf(dl[rr][o], wv[c][o], ga);
                pw[c][o] = __builtin_fmaf(dl[rr][o], fa, pw[c][o]);
            }
            pw[c][4] = __builtin_fmaf(dl[rr][4], fc, pw[c][4]);
            oa |= ((ha & 0x7FFFu) ? f2bf(ga) : 0u) << (16 * c);
            oc |= ((hc & 0x7FFFu) ? f2bf(dl[rr][4] * wv[c][4]) : 0u) << (16 * c);
        }
        uint32_t *dr = reinterpret_cast<uint32_t *>(dh2 + row * (2 * MO_H));
        dr[t] = oa;
        dr[MO_H / 2 + t] = oc;
    }
#pragma unroll
    for (int o = 0; o < 5; ++o)
        reinterpret_cast<float2 *>(partial + ((int64_t)blockIdx.x * 5 + o) * MO_H)[t] = make_float2(pw[0][o], pw[1][o]);
}

}  // namespace

extern "C" int g2048_mlp_embed_fwd(const uint8_t *boards, const void *wt, const float *bias, void *y, void *onehot, int64_t M, void *stream) {
    if (!boards || !wt || !bias || !y || M <= 0 || (((uintptr_t)boards | (uintptr_t)wt | (uintptr_t)bias | (uintptr_t)y | (uintptr_t)onehot) & 15))
        return G2048_EINVAL;
    hipLaunchKernelGGL(k_mlp_embed_fwd, dim3((unsigned)((M + 3) / 4)), dim3(256), 0, (hipStream_t)stream, boards, (const uint16_t *)wt, bias,
                       (uint16_t *)y, (uint16_t *)onehot, M);
    return mlp_done();
}

extern "C" int g2048_gemm_jobs(const g2048_gemm_job *jobs, int n_jobs, int64_t M, void *stream) {
    if (!jobs || n_jobs < 1 || n_jobs > G2048_GEMM_MAX_JOBS || M <= 0) return G2048_EINVAL;
    GemmJobs J;
    J.n_jobs = n_jobs;
    J.M = M;
    int tiles = 0;
    for (int i = 0; i < n_jobs; ++i) {
        const g2048_gemm_job &q = jobs[i];
        if (!q.x[0] || !q.w[0] || !q.y || q.N < GJ_TN || q.N % GJ_TN || q.k[0] < GJ_KC || q.k[0] % GJ_KC || q.k[1] < 0 || q.k[1] % GJ_KC ||
            (q.k[1] && (!q.x[1] || !q.w[1])) || q.ldx[0] < q.k[0] || q.ldw[0] < q.k[0] || (q.k[1] && (q.ldx[1] < q.k[1] || q.ldw[1] < q.k[1])) ||
            q.ldy < q.N || (q.act && q.ldact < q.N) ||
            ((q.ldx[0] | q.ldw[0] | q.ldx[1] | q.ldw[1] | q.ldy) & 7) || (q.ldact & 3) ||
            (((uintptr_t)q.x[0] | (uintptr_t)q.w[0] | (uintptr_t)q.x[1] | (uintptr_t)q.w[1] | (uintptr_t)q.y | (uintptr_t)q.bias) & 15) ||
            ((uintptr_t)q.act & 7))
            return G2048_EINVAL;
        J.job[i] = q;
        J.first_tile[i] = tiles;
        tiles += q.N / GJ_TN;
    }
    J.first_tile[n_jobs] = tiles;
    const int64_t mt = (M + GJ_TM - 1) / GJ_TM;
    if (mt > 0x7FFFFFFF || tiles > 65535) return G2048_EINVAL;
    hipLaunchKernelGGL(k_gemm_jobs, dim3((unsigned)mt, (unsigned)tiles), dim3(GJ_THREADS), 0, (hipStream_t)stream, J);
    return mlp_done();
}

extern "C" int g2048_mlp_out_fwd(const void *h2, const void *w3, float *logits, float *values, int64_t M, void *stream) {
    if (!h2 || !w3 || !logits || !values || M <= 0 || ((uintptr_t)logits & 15) || (((uintptr_t)h2 | (uintptr_t)w3) & 1)) return G2048_EINVAL;
    hipLaunchKernelGGL(k_mlp_out_fwd, dim3((unsigned)((M + 15) / 16)), dim3(256), 0, (hipStream_t)stream, (const uint16_t *)h2,
                       (const uint16_t *)w3, logits, values, M);
    return mlp_done();
}

extern "C" int64_t g2048_mlp_out_bwd_partial_rows(int64_t M) { return M <= 0 ? 0 : (M + MO_ROWS - 1) / MO_ROWS; }

extern "C" int g2048_mlp_out_bwd(const float *dlogits, const float *dvalues, const void *h2, const void *w3, void *dh2, float *partial,
                                 int64_t M, void *stream) {
    if (!dlogits || !dvalues || !h2 || !w3 || !dh2 || !partial || M <= 0 || (((uintptr_t)partial | (uintptr_t)h2 | (uintptr_t)w3 | (uintptr_t)dh2) & 7)) return G2048_EINVAL;
    hipLaunchKernelGGL(k_mlp_out_bwd, dim3((unsigned)((M + MO_ROWS - 1) / MO_ROWS)), dim3(256), 0, (hipStream_t)stream, dlogits, dvalues,
                       (const uint16_t *)h2, (const uint16_t *)w3, (uint16_t *)dh2, partial, M);
    return mlp_done();
}
