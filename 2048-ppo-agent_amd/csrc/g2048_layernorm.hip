// Fused residual-add + dropout + LayerNorm (+ bf16 cast) for the PPO update, forward and backward, d_model 256.
//
// In the pre-norm encoder layers of the policy (reference: nn.TransformerEncoderLayer(norm_first=True) built at
// src/ppo/transformer_encoder.py:138-148) every sub-layer ends with `x = x + dropout(branch)` and the next one starts
// with `h = LayerNorm(x)`; under bf16 autocast PyTorch runs that as dropout (bf16) + add (f32) + LayerNorm (f32) +
// cast to bf16 for the next GEMM: four memory-bound kernels forward and six backward over [tokens, 256].  Here it is
// one kernel each way: one wavefront per token row (4 features per lane, one 16-byte load), statistics by wave
// reduction, dropout mask recomputed from (seed, element index) in the backward, gamma/beta gradients accumulated in
// registers over the rows of a workgroup and flushed with one float atomic per column.
//   x_new = x + dropout(a)                 (f32 residual stream; a = bf16 branch output, may be absent)
//   h     = bf16( (x_new - mean) * rstd * gamma + beta )
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/g2048.h"
#include "g2048_colsum_final.h"

namespace {

constexpr int D = 256, WAVES = 4;
// rows per workgroup of the add+LN backward: 32 for the 34 816-token activations (1 088 workgroups: 26.4 us per launch; 64 rows
// = 544 workgroups 30.8 us; 16 rows no faster and twice the partial rows for g2048_reduce_jobs); 8 for the 2048-row ones of the
// CLS-only layer (32 workgroups whose waves walk 16 rows each, two dependent wave reductions per row, took 12-16 us for 2 MB)
__host__ __device__ inline int ln_rows_per_block(int64_t T) { return T >= 16384 ? 32 : 8; }

__device__ __forceinline__ float wave_sum(float v) {
    for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m);
    return v;
}
__device__ __forceinline__ bool keep_elem(uint32_t s0, uint32_t s1, uint32_t thr, uint64_t idx) {
    uint32_t x = (uint32_t)idx * 0x9E3779B1u ^ s0;
    x ^= (uint32_t)(idx >> 32) * 0x85EBCA77u + s1;
    x ^= x >> 16; x *= 0x7FEB352Du; x ^= x >> 15; x *= 0x846CA68Bu; x ^= x >> 16;
    return (x >> 8) >= thr;
}
__device__ __forceinline__ void mix_seed_state(const uint64_t *seed_state, uint32_t &s0, uint32_t &s1) {
    if (seed_state) {
        const uint64_t s = *seed_state;
        s0 ^= (uint32_t)s * 0x9E3779B1u;
        s1 += (uint32_t)(s >> 32) * 0x85EBCA77u + (uint32_t)s;
    }
}
__device__ __forceinline__ float bf2f(uint32_t hi16) { return __uint_as_float(hi16 << 16); }
__device__ __forceinline__ uint32_t f2bf(float f) {
    const __bf16 b = (__bf16)f;
    return *reinterpret_cast<const uint16_t *>(&b);
}

// x: f32 rows of 256 with row stride x_rs (elements); a: bf16 [T][256] or null; outputs contiguous [T][256]
__global__ void __launch_bounds__(64 * WAVES)
k_add_ln_fwd(const float *__restrict__ x, int64_t x_rs, const uint16_t *__restrict__ a, const float *__restrict__ gamma,
             const float *__restrict__ beta, float *__restrict__ x_new, uint16_t *__restrict__ h,
             float *__restrict__ mean_out, float *__restrict__ rstd_out, int64_t T, float eps, float inv_keep,
             uint32_t thr, uint32_t s0, uint32_t s1, const uint64_t *seed_state) {
    mix_seed_state(seed_state, s0, s1);
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    // gamma NULL: no LayerNorm, h = bf16(x + dropout(a)) (the last sub-layer of the encoder: its output feeds the heads in bf16)
    const bool norm = gamma != nullptr;
    const float4 zero4 = make_float4(0.f, 0.f, 0.f, 0.f);
    const float4 g = norm ? reinterpret_cast<const float4 *>(gamma)[lane] : zero4;
    const float4 bt = norm ? reinterpret_cast<const float4 *>(beta)[lane] : zero4;
    // the next row's loads are issued before this row's two wave reductions (as in the backward kernel)
    const int64_t stride = (int64_t)gridDim.x * WAVES;
    int64_t row = (int64_t)blockIdx.x * WAVES + w;
    float4 vn = make_float4(0.f, 0.f, 0.f, 0.f);
    uint2 an = make_uint2(0u, 0u);
    if (row < T) {
        vn = reinterpret_cast<const float4 *>(x + row * x_rs)[lane];
        if (a) an = reinterpret_cast<const uint2 *>(a + row * D)[lane];
    }
    for (; row < T; row += stride) {
        float4 v = vn;
        const uint2 ab = an;
        if (row + stride < T) {
            vn = reinterpret_cast<const float4 *>(x + (row + stride) * x_rs)[lane];
            if (a) an = reinterpret_cast<const uint2 *>(a + (row + stride) * D)[lane];
        }
        if (a) {
            float av[4] = {bf2f(ab.x & 0xFFFFu), bf2f(ab.x >> 16), bf2f(ab.y & 0xFFFFu), bf2f(ab.y >> 16)};
            if (thr) {
                const uint64_t base = (uint64_t)row * D + 4 * lane;
                for (int q = 0; q < 4; ++q) av[q] = keep_elem(s0, s1, thr, base + q) ? av[q] * inv_keep : 0.0f;
            }
            v.x += av[0]; v.y += av[1]; v.z += av[2]; v.w += av[3];
            if (x_new) reinterpret_cast<float4 *>(x_new + row * D)[lane] = v;
        }
        if (!norm) {
            reinterpret_cast<uint2 *>(h + row * D)[lane] = make_uint2(f2bf(v.x) | (f2bf(v.y) << 16), f2bf(v.z) | (f2bf(v.w) << 16));
            continue;
        }
        const float mean = wave_sum(v.x + v.y + v.z + v.w) * (1.0f / D);
        const float dx = v.x - mean, dy = v.y - mean, dz = v.z - mean, dw = v.w - mean;
        const float rstd = rsqrtf(wave_sum(dx * dx + dy * dy + dz * dz + dw * dw) * (1.0f / D) + eps);
        const uint32_t lo = f2bf(dx * rstd * g.x + bt.x) | (f2bf(dy * rstd * g.y + bt.y) << 16);
        const uint32_t hi = f2bf(dz * rstd * g.z + bt.z) | (f2bf(dw * rstd * g.w + bt.w) << 16);
        reinterpret_cast<uint2 *>(h + row * D)[lane] = make_uint2(lo, hi);
        if (lane == 0) {
            mean_out[row] = mean;
            rstd_out[row] = rstd;
        }
    }
}

// xn: the tensor that was normalised (x_new, or x itself when there was no branch), row stride xn_rs.
// g_x: gradient flowing into x_new from the residual stream (f32, may be null): row r of it belongs to token row r * g_x_period
// (period 1: one row per token; period 17: only the CLS rows of [B][17][256] carry a gradient, g_x is [B][256]);
// g_h: bf16 [T][256].
// dx (f32 [T][256]) = g_x + dLN;  da (bf16 [T][256], may be null) = dropout-masked dx.
// partial[blockIdx][3][256]: this workgroup's column sums of g_h * xhat (-> dgamma), g_h (-> dbeta) and the bf16 values
// written to da (-> the bias gradient of the Linear that produced a); summed over workgroups by k_colsum_final.
__global__ void __launch_bounds__(64 * WAVES)
k_add_ln_bwd(const float *__restrict__ xn, int64_t xn_rs, const float *__restrict__ g_x, const uint16_t *__restrict__ g_h,
             const float *__restrict__ mean_in, const float *__restrict__ rstd_in, const float *__restrict__ gamma,
             float *__restrict__ dx, uint16_t *__restrict__ da, float *__restrict__ partial,
             int64_t T, float inv_keep, uint32_t thr, uint32_t s0, uint32_t s1, const uint64_t *seed_state, int g_x_period) {
    mix_seed_state(seed_state, s0, s1);
    __shared__ float red[WAVES][3][D];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const bool norm = gamma != nullptr;  // NULL: the forward had no LayerNorm (h = bf16(x_new)): dx = g_x + g_h
    const float4 g = norm ? reinterpret_cast<const float4 *>(gamma)[lane] : make_float4(0.f, 0.f, 0.f, 0.f);
    float dg[4] = {0, 0, 0, 0}, db[4] = {0, 0, 0, 0}, dsum[4] = {0, 0, 0, 0};
    const int ROWS_PER_BLOCK = ln_rows_per_block(T);
    const int64_t row0 = (int64_t)blockIdx.x * ROWS_PER_BLOCK;
    // the loads of the next row are issued before this row's reductions (two dependent wave reductions per row would
    // otherwise leave one row's 40 bytes per lane in flight)
    struct RowIn { float4 v, gx; uint2 gb; float mean, rstd; };
    auto fetch = [&](int64_t row) -> RowIn {
        RowIn in;
        in.v = make_float4(0.f, 0.f, 0.f, 0.f);
        in.mean = in.rstd = 0.f;
        if (norm) {  // (uniform)
            in.v = reinterpret_cast<const float4 *>(xn + row * xn_rs)[lane];
            in.mean = mean_in[row];
            in.rstd = rstd_in[row];
        }
        in.gb = reinterpret_cast<const uint2 *>(g_h + row * D)[lane];
        in.gx = make_float4(0.f, 0.f, 0.f, 0.f);
        if (g_x) {
            if (g_x_period == 1) in.gx = reinterpret_cast<const float4 *>(g_x + row * D)[lane];
            else if (row % g_x_period == 0) in.gx = reinterpret_cast<const float4 *>(g_x + (row / g_x_period) * D)[lane];
        }
        return in;
    };
    RowIn nxt;
    if (row0 + w < T) nxt = fetch(row0 + w);
    for (int r = w; r < ROWS_PER_BLOCK; r += WAVES) {
        const int64_t row = row0 + r;
        if (row >= T) break;
        const RowIn cur = nxt;
        if (r + WAVES < ROWS_PER_BLOCK && row + WAVES < T) nxt = fetch(row + WAVES);
        const float4 v = cur.v;
        const uint2 gb = cur.gb;
        const float gh[4] = {bf2f(gb.x & 0xFFFFu), bf2f(gb.x >> 16), bf2f(gb.y & 0xFFFFu), bf2f(gb.y >> 16)};
        const float mean = cur.mean, rstd = cur.rstd;
        const float xh[4] = {(v.x - mean) * rstd, (v.y - mean) * rstd, (v.z - mean) * rstd, (v.w - mean) * rstd};
        const float gg[4] = {g.x, g.y, g.z, g.w};
        float dxh[4], s1sum = 0.f, s2sum = 0.f;
        for (int q = 0; q < 4; ++q) {
            dxh[q] = gh[q] * gg[q];
            s1sum += dxh[q];
            s2sum += dxh[q] * xh[q];
            dg[q] += gh[q] * xh[q];
            db[q] += gh[q];
        }
        const float4 gx = cur.gx;
        float o[4] = {gx.x, gx.y, gx.z, gx.w};
        if (norm) {
            const float c1 = wave_sum(s1sum) * (1.0f / D), c2 = wave_sum(s2sum) * (1.0f / D);
            for (int q = 0; q < 4; ++q) o[q] += rstd * (dxh[q] - c1 - xh[q] * c2);
        } else {
            for (int q = 0; q < 4; ++q) o[q] += gh[q];
        }
        reinterpret_cast<float4 *>(dx + row * D)[lane] = make_float4(o[0], o[1], o[2], o[3]);
        if (da) {
            if (thr) {
                const uint64_t base = (uint64_t)row * D + 4 * lane;
                for (int q = 0; q < 4; ++q) o[q] = keep_elem(s0, s1, thr, base + q) ? o[q] * inv_keep : 0.0f;
            }
            uint32_t b[4];
            for (int q = 0; q < 4; ++q) {
                b[q] = f2bf(o[q]);
                dsum[q] += bf2f(b[q]);  // what at::sum over the bf16 tensor would add
            }
            reinterpret_cast<uint2 *>(da + row * D)[lane] = make_uint2(b[0] | (b[1] << 16), b[2] | (b[3] << 16));
        }
    }
    for (int q = 0; q < 4; ++q) {
        red[w][0][4 * lane + q] = dg[q];
        red[w][1][4 * lane + q] = db[q];
        red[w][2][4 * lane + q] = dsum[q];
    }
    __syncthreads();
    for (int c = threadIdx.x; c < 3 * D; c += 64 * WAVES) {
        const int which = c / D, col = c - which * D;
        float s = 0.f;
        for (int ww = 0; ww < WAVES; ++ww) s += red[ww][which][col];
        partial[(int64_t)blockIdx.x * 3 * D + c] = s;
    }
}

inline int done() {
    const hipError_t e = hipGetLastError();
    return e == hipSuccess ? 0 : -(1000 + (int)e);
}

}  // namespace

extern "C" int g2048_add_ln_fwd(const float *x, int64_t x_row_stride, const void *a, const float *gamma, const float *beta,
                                float *x_new, void *h, float *mean, float *rstd, int64_t T, float eps, float p_drop,
                                uint64_t seed, const uint64_t *seed_state, void *stream) {
    if (!x || !h || T <= 0 || (gamma && (!beta || !mean || !rstd || (a && !x_new))) || !(p_drop >= 0.f && p_drop < 1.f) ||
        (x_row_stride & 3) || (((uintptr_t)x | (uintptr_t)gamma | (uintptr_t)beta | (uintptr_t)x_new) & 15) ||
        (((uintptr_t)a | (uintptr_t)h) & 7))
        return G2048_EINVAL;
    const uint32_t thr = a ? (uint32_t)(p_drop * 16777216.0f) : 0u;
    // four workgroups per CU whose waves walk ~8 rows each with the next row's loads in flight: 21.6 us per [34 816 x 256] launch
    // (one row per wave, 8 192 workgroups: 24.6 us)
    const int64_t blocks = (T + WAVES - 1) / WAVES;
    hipLaunchKernelGGL(k_add_ln_fwd, dim3((unsigned)(blocks < 1024 ? blocks : 1024)), dim3(64 * WAVES), 0, (hipStream_t)stream, x,
                       x_row_stride, (const uint16_t *)a, gamma, beta, x_new, (uint16_t *)h, mean, rstd, T, eps,
                       1.0f / (1.0f - p_drop), thr, (uint32_t)seed, (uint32_t)(seed >> 32), seed_state);
    return done();
}

// ---- column sums (bias gradients) ------------------------------------------------------------------------------------
// out[c] = sum over rows of x[r][c], x bf16 or f32 [T][N] (row stride in elements), f32 accumulation, fixed summation
// order (bit-reproducible).  Two launches: per-workgroup partial sums of an interleaved subset of the rows, then one
// pass over the partials.  Replaces at::sum(dim=0) in the backward of every Linear of the update: that kernel's
// cross-workgroup stage relies on a memset of its semaphores, which a replayed hipGraph does not reproduce reliably on
// this stack (bias gradients differ from eager on every batch but the captured one, tools/debug_graph_grads.py).
namespace {

constexpr int CS_THREADS = 256, CS_VEC = 4;

template <bool BF16>
__global__ void __launch_bounds__(CS_THREADS)
k_colsum_partial(const void *__restrict__ x, int64_t rs, int64_t T, int N, float *__restrict__ partial) {
    __shared__ float red[CS_THREADS][CS_VEC];
    const int cols_v = N / CS_VEC, rows_per_pass = CS_THREADS / cols_v;
    const int cv = threadIdx.x % cols_v, rr = threadIdx.x / cols_v;
    float acc[CS_VEC] = {0.f, 0.f, 0.f, 0.f};
    if (rr < rows_per_pass) {
        auto load = [&](int64_t r) -> float4 {
            if (BF16) {
                const uint2 v = *reinterpret_cast<const uint2 *>((const uint16_t *)x + r * rs + CS_VEC * cv);
                return make_float4(bf2f(v.x & 0xFFFFu), bf2f(v.x >> 16), bf2f(v.y & 0xFFFFu), bf2f(v.y >> 16));
            }
            return *reinterpret_cast<const float4 *>((const float *)x + r * rs + CS_VEC * cv);
        };
        const int64_t step = (int64_t)gridDim.x * rows_per_pass;
        int64_t r = (int64_t)blockIdx.x * rows_per_pass + rr;
        // four rows in flight per thread (a wide matrix leaves one row per pass and workgroup: a single 8-byte load per
        // thread and iteration reached 2.4 TB/s); the order of the additions stays fixed
        for (; r + 3 * step < T; r += 4 * step) {
            const float4 a = load(r), b = load(r + step), c = load(r + 2 * step), d = load(r + 3 * step);
            acc[0] += (a.x + b.x) + (c.x + d.x); acc[1] += (a.y + b.y) + (c.y + d.y);
            acc[2] += (a.z + b.z) + (c.z + d.z); acc[3] += (a.w + b.w) + (c.w + d.w);
        }
        for (; r < T; r += step) {
            const float4 a = load(r);
            acc[0] += a.x; acc[1] += a.y; acc[2] += a.z; acc[3] += a.w;
        }
    }
    for (int q = 0; q < CS_VEC; ++q) red[threadIdx.x][q] = acc[q];
    __syncthreads();
    if (rr == 0) {
        for (int k = 1; k < rows_per_pass; ++k)
            for (int q = 0; q < CS_VEC; ++q) acc[q] += red[threadIdx.x + k * cols_v][q];
        reinterpret_cast<float4 *>(partial + (int64_t)blockIdx.x * N)[cv] = make_float4(acc[0], acc[1], acc[2], acc[3]);
    }
}

}  // namespace

extern "C" int64_t g2048_colsum_workspace_floats(int64_t T, int N) {
    if (T <= 0 || N <= 0) return 0;
    return (int64_t)G2048_COLSUM_MAX_GROUPS * N;
}

namespace {
inline int64_t colsum_groups(int64_t T, int n) {
    const int rows_per_pass = CS_THREADS / (n / CS_VEC);
    int64_t G = (T + rows_per_pass - 1) / rows_per_pass;
    // at least ~16 rows per workgroup, at most MAX_GROUPS workgroups
    G = (G + 15) / 16;
    if (G > G2048_COLSUM_MAX_GROUPS) G = G2048_COLSUM_MAX_GROUPS;
    return G < 1 ? 1 : G;
}
}  // namespace

extern "C" int64_t g2048_colsum_partial_rows(int64_t T, int N) {
    return (T <= 0 || N < CS_VEC || N % CS_VEC || N > CS_THREADS * CS_VEC) ? 0 : colsum_groups(T, N);
}

extern "C" int g2048_colsum(const void *x, int is_bf16, int64_t row_stride, int64_t T, int N, float *workspace, float *out,
                            void *stream) {
    if (!out && N > CS_THREADS * CS_VEC) return G2048_EINVAL;  // first stage only: one column tile
    if (!x || !workspace || T <= 0 || N < CS_VEC || N % CS_VEC || row_stride % CS_VEC || row_stride < N ||
        ((uintptr_t)x & (is_bf16 ? 7 : 15)) || ((uintptr_t)workspace & 15))
        return G2048_EINVAL;
    // wider matrices are summed in column tiles of at most CS_THREADS * CS_VEC (1024) columns, each with its own slice of
    // the workspace (MAX_GROUPS * tile floats, so the slices of all tiles fit in MAX_GROUPS * N)
    constexpr int TILE_N = CS_THREADS * CS_VEC;
    for (int c0 = 0; c0 < N; c0 += TILE_N) {
        const int n = N - c0 < TILE_N ? N - c0 : TILE_N;
        const void *xt = is_bf16 ? (const void *)((const uint16_t *)x + c0) : (const void *)((const float *)x + c0);
        float *ws = workspace + (int64_t)G2048_COLSUM_MAX_GROUPS * c0;
        const int64_t G = colsum_groups(T, n);
        if (is_bf16)
            hipLaunchKernelGGL(k_colsum_partial<true>, dim3((unsigned)G), dim3(CS_THREADS), 0, (hipStream_t)stream, xt, row_stride, T,
                               n, ws);
        else
            hipLaunchKernelGGL(k_colsum_partial<false>, dim3((unsigned)G), dim3(CS_THREADS), 0, (hipStream_t)stream, xt, row_stride, T,
                               n, ws);
        if (out)
            hipLaunchKernelGGL(k_colsum_final, dim3((unsigned)((n + CF_COLS - 1) / CF_COLS)), dim3(CF_COLS * CF_SLICES), 0,
                               (hipStream_t)stream, ws, (int)G, n, out + c0);
    }
    return done();
}

extern "C" int64_t g2048_add_ln_bwd_workspace_floats(int64_t T) {
    return T <= 0 ? 0 : ((T + ln_rows_per_block(T) - 1) / ln_rows_per_block(T)) * 3 * D;
}

extern "C" int g2048_add_ln_bwd(const float *x_norm, int64_t x_row_stride, const float *g_x, const void *g_h, const float *mean,
                                const float *rstd, const float *gamma, float *dx, void *da, float *dparams, float *workspace,
                                int64_t T, float p_drop, uint64_t seed, const uint64_t *seed_state, int g_x_period, void *stream) {
    if (g_x_period < 1) return G2048_EINVAL;
    if (!g_h || !dx || !workspace || T <= 0 || (x_row_stride & 3) || (gamma && (!x_norm || !mean || !rstd)) ||
        !(p_drop >= 0.f && p_drop < 1.f) || (((uintptr_t)x_norm | (uintptr_t)g_x | (uintptr_t)dx | (uintptr_t)gamma) & 15) ||
        (((uintptr_t)g_h | (uintptr_t)da) & 7))
        return G2048_EINVAL;
    const uint32_t thr = da ? (uint32_t)(p_drop * 16777216.0f) : 0u;
    const int64_t blocks = (T + ln_rows_per_block(T) - 1) / ln_rows_per_block(T);
    hipLaunchKernelGGL(k_add_ln_bwd, dim3((unsigned)blocks), dim3(64 * WAVES), 0, (hipStream_t)stream, x_norm, x_row_stride, g_x,
                       (const uint16_t *)g_h, mean, rstd, gamma, dx, (uint16_t *)da, workspace, T, 1.0f / (1.0f - p_drop), thr,
                       (uint32_t)seed, (uint32_t)(seed >> 32), seed_state, g_x_period);
    if (dparams)
        hipLaunchKernelGGL(k_colsum_final, dim3(3 * D / CF_COLS), dim3(CF_COLS * CF_SLICES), 0, (hipStream_t)stream, workspace, (int)blocks,
                           3 * D, dparams);
    return done();
}

// ---- feed-forward activation: y = dropout(relu(x)) ----------------------------------------------------------------------
// Forward: one pass instead of relu + dropout (+ a saved bool mask).  Backward: dx = dy / (1 - p) where y != 0 (y is
// non-zero exactly where the unit was active AND kept, so neither x nor a mask is saved), fused with the column sums of
// dx = the bias gradient of the Linear in front (replaces masked_scale + threshold_backward + a column-sum pass).
namespace {

constexpr int RD_THREADS = 256, RD_VEC = 8;
// rows per workgroup: 64 for the 34 816-token activations, 8 for the 2048-row ones of the CLS-only layer (32 workgroups of 64
// rows took 20 us for 8 MB)
__host__ __device__ inline int rd_rows_per_block(int64_t T) { return T >= 16384 ? 64 : 8; }

__global__ void __launch_bounds__(RD_THREADS)
k_relu_dropout_fwd(const uint4 *__restrict__ x, uint4 *__restrict__ y, int64_t n_vec, float inv_keep, uint32_t thr16, uint32_t s0,
                   uint32_t s1, const uint64_t *seed_state) {
    mix_seed_state(seed_state, s0, s1);
    for (int64_t v = (int64_t)blockIdx.x * RD_THREADS + threadIdx.x; v < n_vec; v += (int64_t)gridDim.x * RD_THREADS) {
        const uint4 in = x[v];
        const uint32_t w[4] = {in.x, in.y, in.z, in.w};
        uint32_t o[4];
        for (int q = 0; q < 4; ++q) {
            uint32_t hsh = (uint32_t)(4 * v + q) * 0x9E3779B1u ^ s0;
            hsh ^= (uint32_t)((uint64_t)(4 * v + q) >> 32) * 0x85EBCA77u + s1;
            hsh ^= hsh >> 16; hsh *= 0x7FEB352Du; hsh ^= hsh >> 15; hsh *= 0x846CA68Bu; hsh ^= hsh >> 16;
            const float a = bf2f(w[q] & 0xFFFFu), b = bf2f(w[q] >> 16);
            const float ya = (a > 0.f && (hsh & 0xFFFFu) >= thr16) ? a * inv_keep : 0.f;
            const float yb = (b > 0.f && (hsh >> 16) >= thr16) ? b * inv_keep : 0.f;
            o[q] = f2bf(ya) | (f2bf(yb) << 16);
        }
        y[v] = make_uint4(o[0], o[1], o[2], o[3]);
    }
}

// dy, y, dx: bf16 [T][F] (dx may alias dy); partial[blockIdx][F]
__global__ void __launch_bounds__(RD_THREADS)
k_relu_dropout_bwd(const uint4 *__restrict__ dy, const uint4 *__restrict__ y, uint4 *__restrict__ dx, float *__restrict__ partial,
                   int64_t T, int F, float inv_keep) {
    __shared__ float red[RD_THREADS][RD_VEC];
    const int cols_v = F / RD_VEC, rows_per_pass = RD_THREADS / cols_v;
    const int cv = threadIdx.x % cols_v, rr = threadIdx.x / cols_v;
    float acc[RD_VEC] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    const int rpb = rd_rows_per_block(T);
    const int64_t row0 = (int64_t)blockIdx.x * rpb;
    if (rr < rows_per_pass) {
        for (int r = rr; r < rpb; r += rows_per_pass) {
            const int64_t row = row0 + r;
            if (row >= T) break;
            const int64_t v = row * cols_v + cv;
            const uint4 g = dy[v], yy = y[v];
            const uint32_t gw[4] = {g.x, g.y, g.z, g.w}, yw[4] = {yy.x, yy.y, yy.z, yy.w};
            uint32_t o[4];
            for (int q = 0; q < 4; ++q) {
                const uint32_t lo = (yw[q] & 0x7FFFu) ? f2bf(bf2f(gw[q] & 0xFFFFu) * inv_keep) : 0u;
                const uint32_t hi = (yw[q] & 0x7FFF0000u) ? f2bf(bf2f(gw[q] >> 16) * inv_keep) : 0u;
                acc[2 * q] += bf2f(lo);
                acc[2 * q + 1] += bf2f(hi);
                o[q] = lo | (hi << 16);
            }
            dx[v] = make_uint4(o[0], o[1], o[2], o[3]);
        }
    }
    for (int q = 0; q < RD_VEC; ++q) red[threadIdx.x][q] = acc[q];
    __syncthreads();
    if (rr == 0) {
        for (int k = 1; k < rows_per_pass; ++k)
            for (int q = 0; q < RD_VEC; ++q) acc[q] += red[threadIdx.x + k * cols_v][q];
        float *dst = partial + (int64_t)blockIdx.x * F + RD_VEC * cv;
        reinterpret_cast<float4 *>(dst)[0] = make_float4(acc[0], acc[1], acc[2], acc[3]);
        reinterpret_cast<float4 *>(dst)[1] = make_float4(acc[4], acc[5], acc[6], acc[7]);
    }
}

inline bool rd_shape_ok(int64_t T, int F) { return T > 0 && F >= RD_VEC && F % RD_VEC == 0 && F / RD_VEC <= RD_THREADS; }

}  // namespace

extern "C" int g2048_relu_dropout_fwd(const void *x, void *y, int64_t T, int F, float p_drop, uint64_t seed,
                                      const uint64_t *seed_state, void *stream) {
    if (!x || !y || !rd_shape_ok(T, F) || !(p_drop >= 0.f && p_drop < 1.f) || (((uintptr_t)x | (uintptr_t)y) & 15)) return G2048_EINVAL;
    const int64_t n_vec = T * (F / RD_VEC);
    int64_t blocks = (n_vec + RD_THREADS - 1) / RD_THREADS;
    if (blocks > 16384) blocks = 16384;
    hipLaunchKernelGGL(k_relu_dropout_fwd, dim3((unsigned)blocks), dim3(RD_THREADS), 0, (hipStream_t)stream, (const uint4 *)x, (uint4 *)y,
                       n_vec, 1.0f / (1.0f - p_drop), (uint32_t)(p_drop * 65536.0f + 0.5f), (uint32_t)seed, (uint32_t)(seed >> 32),
                       seed_state);
    return done();
}

extern "C" int64_t g2048_relu_dropout_bwd_workspace_floats(int64_t T, int F) {
    return rd_shape_ok(T, F) ? ((T + rd_rows_per_block(T) - 1) / rd_rows_per_block(T)) * F : 0;
}

extern "C" int g2048_relu_dropout_bwd(const void *dy, const void *y, void *dx, float *dbias, float *workspace, int64_t T, int F,
                                      float p_drop, void *stream) {
    if (!dy || !y || !dx || !workspace || !rd_shape_ok(T, F) || !(p_drop >= 0.f && p_drop < 1.f) ||
        (((uintptr_t)dy | (uintptr_t)y | (uintptr_t)dx | (uintptr_t)workspace) & 15))
        return G2048_EINVAL;
    const int64_t blocks = (T + rd_rows_per_block(T) - 1) / rd_rows_per_block(T);
    hipLaunchKernelGGL(k_relu_dropout_bwd, dim3((unsigned)blocks), dim3(RD_THREADS), 0, (hipStream_t)stream, (const uint4 *)dy,
                       (const uint4 *)y, (uint4 *)dx, workspace, T, F, 1.0f / (1.0f - p_drop));
    if (dbias)
        hipLaunchKernelGGL(k_colsum_final, dim3((unsigned)((F + CF_COLS - 1) / CF_COLS)), dim3(CF_COLS * CF_SLICES), 0, (hipStream_t)stream,
                           workspace, (int)blocks, F, dbias);
    return done();
}

// ---- token embedding of packed boards (update) ---------------------------------------------------------------------------
// x0[m][0] = cls;  x0[m][1 + c] = dropout(Wt[boards[m][c]] + pe[c])   (f32 [M][17][256]).
// Reference: PPOAgent.forward's input_embedding (a bias-free Linear over the one-hot cell, src/ppo/ppo_agent.py:59-66,
// 103-106) + PositionalEncoding2D + CLS concat (src/ppo/transformer_encoder.py:150-190).  In PyTorch that is a one-hot
// expansion, a K = 31 GEMM, an add and a cat forward, and backward a [256 x 32768] x [32768 x 31] GEMM (146 us: the
// reduction runs over every token of the minibatch), slice copies and a row sum for the CLS token.  Here the forward is
// a gather and the backward a segmented sum: each wave adds its token rows into its own [32 classes][256] f32 image in
// LDS (class 31 = the CLS token; plain read-modify-write, nobody else touches that image, so the order is fixed), the
// four images of a workgroup are summed, and k_colsum_final adds the workgroups' partials.
namespace {

constexpr int EMB_D = 256, EMB_CLASSES = 32, EMB_SEQ = 17, EMB_BLOCKS = 256;

// LN: also h = bf16(LayerNorm(x0 row)) and the row's mean / rstd - the first LayerNorm of the encoder (layers[0].norm1) on the row the
// wave still holds, in k_add_ln_fwd's arithmetic (that launch read the 36 MB back for it: 13.7 us per minibatch)
struct EmbedLN {
    const float *gamma, *beta;
    uint16_t *h;
    float *mean, *rstd;
    float eps;
};

template <bool LN>
__global__ void __launch_bounds__(256)
k_embed_fwd(const uint8_t *__restrict__ boards, const float *__restrict__ wt, int w_ld, const float *__restrict__ pe,
            const float *__restrict__ cls, float *__restrict__ x0, int64_t n_rows, float inv_keep, uint32_t thr, uint32_t s0,
            uint32_t s1, const uint64_t *seed_state, EmbedLN L) {
    // the nn.Linear weight itself ([256][w_ld], w_ld >= 31) is turned into the class-major table [31][256] in LDS once per workgroup:
    // read per row it was four strided 4-byte loads per lane (19 us per launch for 36 MB of output)
    __shared__ __attribute__((aligned(16))) float table[31 * EMB_D];
    mix_seed_state(seed_state, s0, s1);
    const int lane = threadIdx.x & 63;
    float4 g = make_float4(0.f, 0.f, 0.f, 0.f), bt = g;
    if (LN) g = reinterpret_cast<const float4 *>(L.gamma)[lane], bt = reinterpret_cast<const float4 *>(L.beta)[lane];
    if (w_ld != 0) {
        const float *w = wt + (size_t)threadIdx.x * w_ld;  // thread d copies row d of the weight into column d of the table
        for (int e = 0; e < 31; ++e) table[e * EMB_D + threadIdx.x] = w[e];
        __syncthreads();
    }
    for (int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6); row < n_rows; row += (int64_t)gridDim.x * 4) {
        const int64_t m = row / EMB_SEQ;
        const int c = (int)(row - m * EMB_SEQ);
        float4 v;
        if (c == 0) {
            v = reinterpret_cast<const float4 *>(cls)[lane];
        } else {
            const int e = min((int)boards[m * 16 + c - 1], 30);  // wt has 31 rows; the env never exceeds 17
            const float4 a = w_ld == 0 ? reinterpret_cast<const float4 *>(wt + (size_t)e * EMB_D)[lane]  // class-major table [31][256]
                                       : reinterpret_cast<const float4 *>(table + e * EMB_D)[lane];
            const float4 b = reinterpret_cast<const float4 *>(pe + (size_t)(c - 1) * EMB_D)[lane];
            v = make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w);
            if (thr) {
                const uint64_t base = (uint64_t)row * EMB_D + 4 * lane;
                v.x = keep_elem(s0, s1, thr, base + 0) ? v.x * inv_keep : 0.f;
                v.y = keep_elem(s0, s1, thr, base + 1) ? v.y * inv_keep : 0.f;
                v.z = keep_elem(s0, s1, thr, base + 2) ? v.z * inv_keep : 0.f;
                v.w = keep_elem(s0, s1, thr, base + 3) ? v.w * inv_keep : 0.f;
            }
        }
        reinterpret_cast<float4 *>(x0 + row * EMB_D)[lane] = v;
        if (LN) {
            const float mean = wave_sum(v.x + v.y + v.z + v.w) * (1.0f / EMB_D);
            const float dx = v.x - mean, dy = v.y - mean, dz = v.z - mean, dw = v.w - mean;
            const float rstd = rsqrtf(wave_sum(dx * dx + dy * dy + dz * dz + dw * dw) * (1.0f / EMB_D) + L.eps);
            const uint32_t lo = f2bf(dx * rstd * g.x + bt.x) | (f2bf(dy * rstd * g.y + bt.y) << 16);
            const uint32_t hi = f2bf(dz * rstd * g.z + bt.z) | (f2bf(dw * rstd * g.w + bt.w) << 16);
            reinterpret_cast<uint2 *>(L.h + row * EMB_D)[lane] = make_uint2(lo, hi);
            if (lane == 0) {
                L.mean[row] = mean;
                L.rstd[row] = rstd;
            }
        }
    }
}

// partial[blockIdx][32][256]
__global__ void __launch_bounds__(256)
k_embed_bwd(const uint8_t *__restrict__ boards, const float *__restrict__ dx0, float *__restrict__ partial, int64_t n_rows,
            float inv_keep, uint32_t thr, uint32_t s0, uint32_t s1, const uint64_t *seed_state) {
    mix_seed_state(seed_state, s0, s1);
    extern __shared__ __attribute__((aligned(16))) char emb_smem[];
    float4 *img = reinterpret_cast<float4 *>(emb_smem);  // [4 waves][32 classes][64 lanes] float4
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    float4 *mine = img + (size_t)w * EMB_CLASSES * 64;
    for (int k = 0; k < EMB_CLASSES; ++k) mine[k * 64 + lane] = make_float4(0.f, 0.f, 0.f, 0.f);
    const int64_t per_block = (n_rows + gridDim.x - 1) / gridDim.x;
    const int64_t r0 = (int64_t)blockIdx.x * per_block, r1 = (r0 + per_block < n_rows) ? r0 + per_block : n_rows;
    // four rows of this wave in flight: the read-modify-write of the LDS image is a dependent chain per row, and one 16-byte
    // load per lane and iteration left the kernel at 1.4 TB/s
    constexpr int PF = 4;
    for (int64_t row = r0 + w; row < r1; row += 4 * PF) {
        float4 g[PF];
        int cls[PF], col[PF];
        for (int u = 0; u < PF; ++u) {
            const int64_t rw = row + 4 * u;
            cls[u] = -1;
            if (rw < r1) {
                const int64_t m = rw / EMB_SEQ;
                col[u] = (int)(rw - m * EMB_SEQ);
                cls[u] = col[u] == 0 ? EMB_CLASSES - 1 : min((int)boards[m * 16 + col[u] - 1], 30);
                g[u] = reinterpret_cast<const float4 *>(dx0 + rw * EMB_D)[lane];
            }
        }
        for (int u = 0; u < PF; ++u) {
            if (cls[u] < 0) break;
            float4 gg = g[u];
            if (thr && col[u] != 0) {
                const uint64_t base = (uint64_t)(row + 4 * u) * EMB_D + 4 * lane;
                gg.x = keep_elem(s0, s1, thr, base + 0) ? gg.x * inv_keep : 0.f;
                gg.y = keep_elem(s0, s1, thr, base + 1) ? gg.y * inv_keep : 0.f;
                gg.z = keep_elem(s0, s1, thr, base + 2) ? gg.z * inv_keep : 0.f;
                gg.w = keep_elem(s0, s1, thr, base + 3) ? gg.w * inv_keep : 0.f;
            }
            float4 a = mine[cls[u] * 64 + lane];
            a.x += gg.x; a.y += gg.y; a.z += gg.z; a.w += gg.w;
            mine[cls[u] * 64 + lane] = a;
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < EMB_CLASSES * 64; i += 256) {
        float4 s = img[i];
        for (int ww = 1; ww < 4; ++ww) {
            const float4 t = img[(size_t)ww * EMB_CLASSES * 64 + i];
            s.x += t.x; s.y += t.y; s.z += t.z; s.w += t.w;
        }
        reinterpret_cast<float4 *>(partial + (size_t)blockIdx.x * EMB_CLASSES * EMB_D)[i] = s;
    }
}

}  // namespace

extern "C" int g2048_embed_fwd(const uint8_t *boards, const float *wt, int w_ld, const float *pe, const float *cls, float *x0, int64_t M,
                               float p_drop, uint64_t seed, const uint64_t *seed_state, void *stream) {
    if (!boards || !wt || !pe || !cls || !x0 || M <= 0 || !(p_drop >= 0.f && p_drop < 1.f) || (w_ld != 0 && w_ld < 31) ||
        (((uintptr_t)pe | (uintptr_t)cls | (uintptr_t)x0) & 15) || ((uintptr_t)wt & (w_ld ? 3 : 15)))
        return G2048_EINVAL;
    const int64_t n_rows = M * EMB_SEQ;
    int64_t blocks = (n_rows + 3) / 4;
    if (blocks > 1024) blocks = 1024;  // four workgroups per CU: the table is staged 1 024 times (32 MB of L2 reads)
    hipLaunchKernelGGL(k_embed_fwd<false>, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, boards, wt, w_ld, pe, cls, x0, n_rows,
                       1.0f / (1.0f - p_drop), (uint32_t)(p_drop * 16777216.0f), (uint32_t)seed, (uint32_t)(seed >> 32), seed_state,
                       EmbedLN{});
    return done();
}

extern "C" int g2048_embed_ln_fwd(const uint8_t *boards, const float *wt, int w_ld, const float *pe, const float *cls, float *x0, int64_t M,
                                  float p_drop, uint64_t seed, const uint64_t *seed_state, const float *gamma, const float *beta, float eps,
                                  void *h, float *mean, float *rstd, void *stream) {
    if (!boards || !wt || !pe || !cls || !x0 || M <= 0 || !(p_drop >= 0.f && p_drop < 1.f) || (w_ld != 0 && w_ld < 31) || !gamma || !beta ||
        !h || !mean || !rstd || (((uintptr_t)pe | (uintptr_t)cls | (uintptr_t)x0 | (uintptr_t)gamma | (uintptr_t)beta) & 15) ||
        ((uintptr_t)h & 7) || ((uintptr_t)wt & (w_ld ? 3 : 15)))
        return G2048_EINVAL;
    const int64_t n_rows = M * EMB_SEQ;
    int64_t blocks = (n_rows + 3) / 4;
    if (blocks > 1024) blocks = 1024;
    EmbedLN L;
    L.gamma = gamma, L.beta = beta, L.h = (uint16_t *)h, L.mean = mean, L.rstd = rstd, L.eps = eps;
    hipLaunchKernelGGL(k_embed_fwd<true>, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, boards, wt, w_ld, pe, cls, x0, n_rows,
                       1.0f / (1.0f - p_drop), (uint32_t)(p_drop * 16777216.0f), (uint32_t)seed, (uint32_t)(seed >> 32), seed_state, L);
    return done();
}

extern "C" int64_t g2048_embed_bwd_workspace_floats(int64_t M) { return M <= 0 ? 0 : (int64_t)EMB_BLOCKS * EMB_CLASSES * EMB_D; }

extern "C" int g2048_embed_bwd(const uint8_t *boards, const float *dx0, float *dwt_dcls, float *workspace, int64_t M, float p_drop,
                               uint64_t seed, const uint64_t *seed_state, void *stream) {
    if (!boards || !dx0 || !workspace || M <= 0 || !(p_drop >= 0.f && p_drop < 1.f) ||
        (((uintptr_t)dx0 | (uintptr_t)dwt_dcls | (uintptr_t)workspace) & 15))
        return G2048_EINVAL;
    // per call, not latched: the attribute is per device, and a latch would be the library's only global state
    const int lds = 4 * EMB_CLASSES * EMB_D * (int)sizeof(float);
    if (hipFuncSetAttribute(reinterpret_cast<const void *>(k_embed_bwd), hipFuncAttributeMaxDynamicSharedMemorySize, lds) != hipSuccess)
        return -(1000 + (int)hipGetLastError());
    const int64_t n_rows = M * EMB_SEQ;
    hipLaunchKernelGGL(k_embed_bwd, dim3(EMB_BLOCKS), dim3(256), lds, (hipStream_t)stream, boards, dx0, workspace, n_rows,
                       1.0f / (1.0f - p_drop), (uint32_t)(p_drop * 16777216.0f), (uint32_t)seed, (uint32_t)(seed >> 32), seed_state);
    if (dwt_dcls)
        hipLaunchKernelGGL(k_colsum_final, dim3(EMB_CLASSES * EMB_D / CF_COLS), dim3(CF_COLS * CF_SLICES), 0, (hipStream_t)stream,
                           workspace, EMB_BLOCKS, EMB_CLASSES * EMB_D, dwt_dcls);
    return done();
}
