// Second stage of every gradient reduction of the PPO update in ONE launch.
//
// The backward of the policy produces ~50 small reductions per minibatch: 16 split-K slices of every weight gradient
// (bf16 [16][out][in] -> f32 [out][in]), per-workgroup partial column sums of every bias / LayerNorm gradient
// (f32 [G][N] -> f32 [N]), bf16 -> f32 conversions of the small weight gradients.  As separate launches (at::sum,
// k_colsum_final, copy kernels) each costs 4-6 us of launch floor for microseconds of work: 0.25 ms of a 3 ms minibatch.
// Here the producers only leave their partials behind and register a job; g2048_reduce_jobs runs all jobs of a backward
// pass at once: out[c] = sum over parts p of src[p * part_stride + c], f32 accumulation in a fixed order
// (bit-reproducible), straight into the optimiser's flat gradient buffer.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/g2048.h"

namespace {

// Two workgroup shapes, chosen per job on the host:
//   few parts (the split-K slices of a weight gradient, n up to 786 432): 256 threads x 4 columns, every thread walks the
//     parts of its columns itself, four loads in flight - coalesced 512-byte rows per wave, no LDS;
//   many parts (per-workgroup partial column sums, n <= 1024): 16 column groups x 16 part-slices, combined through LDS.
constexpr int RJ_TX = 16, RJ_TY = 16, RJ_VEC = 4, RJ_COLS = RJ_TX * RJ_VEC;  // 64 columns x 16 part-slices per workgroup
constexpr int RJ_THREADS = RJ_TX * RJ_TY, RJ_WIDE_COLS = RJ_THREADS * RJ_VEC;   // 1024 columns per workgroup
constexpr int RJ_WIDE_MAX_PARTS = 32;
constexpr int RJ_MAX = G2048_REDUCE_MAX_JOBS;

struct JobTable {
    g2048_reduce_job job[RJ_MAX];
    int32_t first_block[RJ_MAX + 1];
    int32_t n_jobs;
};

__device__ __forceinline__ float bf2f(uint32_t hi16) { return __uint_as_float(hi16 << 16); }

constexpr int RJ_WIDE8_COLS = RJ_THREADS * 8;  // 2048 columns per workgroup
// the all-vector case of the wide shape (host and device must agree: it decides the number of workgroups of the job)
__host__ __device__ inline bool wide8(const g2048_reduce_job &J) {
    return !J.transpose_rows && J.src_bf16 && J.parts <= RJ_WIDE_MAX_PARTS && J.n % 8 == 0 && J.part_stride % 8 == 0 && !((uintptr_t)J.src & 15) &&
           !((uintptr_t)J.dst & 15);
}

__global__ void __launch_bounds__(RJ_TX * RJ_TY)
k_reduce_jobs(JobTable T) {
    __shared__ float red[RJ_TY][RJ_COLS + 4];
    static_assert(RJ_THREADS == 256, "");
    int j = 0;
    while (j + 1 < T.n_jobs && (int)blockIdx.x >= T.first_block[j + 1]) ++j;  // <= 64 entries, uniform
    const g2048_reduce_job J = T.job[j];
    if (wide8(J)) {  // bf16 slices of a weight gradient: 8 columns per thread, 16-byte loads, eight parts in flight
        const int64_t c0 = ((int64_t)blockIdx.x - T.first_block[j]) * RJ_WIDE8_COLS + (int64_t)threadIdx.x * 8;
        if (c0 >= J.n) return;
        const uint16_t *src = reinterpret_cast<const uint16_t *>(J.src) + c0;
        float a[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        auto add = [&](const uint4 &v) {
            a[0] += bf2f(v.x & 0xFFFFu); a[1] += bf2f(v.x >> 16); a[2] += bf2f(v.y & 0xFFFFu); a[3] += bf2f(v.y >> 16);
            a[4] += bf2f(v.z & 0xFFFFu); a[5] += bf2f(v.z >> 16); a[6] += bf2f(v.w & 0xFFFFu); a[7] += bf2f(v.w >> 16);
        };
        int p = 0;
        for (; p + 8 <= J.parts; p += 8) {
            uint4 v[8];
            for (int k = 0; k < 8; ++k) v[k] = *reinterpret_cast<const uint4 *>(src + (int64_t)(p + k) * J.part_stride);
            for (int k = 0; k < 8; ++k) add(v[k]);
        }
        for (; p < J.parts; ++p) add(*reinterpret_cast<const uint4 *>(src + (int64_t)p * J.part_stride));
        float4 *dst = reinterpret_cast<float4 *>(J.dst + c0);
        dst[0] = make_float4(a[0], a[1], a[2], a[3]);
        dst[1] = make_float4(a[4], a[5], a[6], a[7]);
        return;
    }
    const bool wide = J.parts <= RJ_WIDE_MAX_PARTS && !J.transpose_rows;
    const int tx = wide ? (int)threadIdx.x : (int)threadIdx.x % RJ_TX, ty = wide ? 0 : (int)threadIdx.x / RJ_TX;
    const int p_step = wide ? 1 : RJ_TY;
    const int c0 = ((int)blockIdx.x - T.first_block[j]) * (wide ? RJ_WIDE_COLS : RJ_COLS) + tx * RJ_VEC;
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
    if (c0 < J.n) {
        const bool vec = c0 + RJ_VEC <= J.n && !(J.part_stride & 3) && !((uintptr_t)J.src & (J.src_bf16 ? 7 : 15));
        auto load = [&](int p) -> float4 {
            const int64_t off = (int64_t)p * J.part_stride + c0;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (J.src_bf16) {
                const uint16_t *s = reinterpret_cast<const uint16_t *>(J.src) + off;
                if (vec) {
                    const uint2 w = *reinterpret_cast<const uint2 *>(s);
                    v = make_float4(bf2f(w.x & 0xFFFFu), bf2f(w.x >> 16), bf2f(w.y & 0xFFFFu), bf2f(w.y >> 16));
                } else {
                    v.x = bf2f(s[0]);
                    if (c0 + 1 < J.n) v.y = bf2f(s[1]);
                    if (c0 + 2 < J.n) v.z = bf2f(s[2]);
                    if (c0 + 3 < J.n) v.w = bf2f(s[3]);
                }
            } else {
                const float *s = reinterpret_cast<const float *>(J.src) + off;
                if (vec) {
                    v = *reinterpret_cast<const float4 *>(s);
                } else {
                    v.x = s[0];
                    if (c0 + 1 < J.n) v.y = s[1];
                    if (c0 + 2 < J.n) v.z = s[2];
                    if (c0 + 3 < J.n) v.w = s[3];
                }
            }
            return v;
        };
        int p = ty;
        for (; p + 3 * p_step < J.parts; p += 4 * p_step) {  // four independent loads in flight, added in a fixed order
            const float4 u0 = load(p), u1 = load(p + p_step), u2 = load(p + 2 * p_step), u3 = load(p + 3 * p_step);
            a0 += (u0.x + u1.x) + (u2.x + u3.x); a1 += (u0.y + u1.y) + (u2.y + u3.y);
            a2 += (u0.z + u1.z) + (u2.z + u3.z); a3 += (u0.w + u1.w) + (u2.w + u3.w);
        }
        for (; p < J.parts; p += p_step) {
            const float4 u = load(p);
            a0 += u.x; a1 += u.y; a2 += u.z; a3 += u.w;
        }
    }
    if (wide) {  // (uniform per workgroup)
        if (c0 + RJ_VEC <= J.n && !((uintptr_t)J.dst & 15)) {
            *reinterpret_cast<float4 *>(J.dst + c0) = make_float4(a0, a1, a2, a3);
        } else if (c0 < J.n) {
            const float s[4] = {a0, a1, a2, a3};
            for (int q = 0; q < 4 && c0 + q < J.n; ++q) J.dst[c0 + q] = s[q];
        }
        return;
    }
    float *r = &red[ty][tx * RJ_VEC];
    r[0] = a0; r[1] = a1; r[2] = a2; r[3] = a3;
    __syncthreads();
    if (ty == 0 && c0 < J.n) {
        float s[4] = {0.f, 0.f, 0.f, 0.f};
        for (int k = 0; k < RJ_TY; ++k)
            for (int q = 0; q < 4; ++q) s[q] += red[k][tx * RJ_VEC + q];
        // transpose_rows = R: the n columns are a row-major [R][n / R] matrix whose sum is stored transposed ([n / R][R]): the
        // embedding gradient is accumulated per class ([31][256]) and belongs to an nn.Linear weight ([256][31])
        const int R = J.transpose_rows, cols = R ? J.n / R : 0;
        for (int q = 0; q < 4; ++q) {
            const int c = c0 + q;
            if (c < J.n) J.dst[R ? (int64_t)(c % cols) * R + c / cols : c] = s[q];
        }
    }
}

}  // namespace

extern "C" int g2048_reduce_jobs(const g2048_reduce_job *jobs, int n_jobs, void *stream) {
    if (n_jobs < 0 || (n_jobs > 0 && !jobs)) return G2048_EINVAL;
    if (n_jobs == 0) return 0;
    for (int i = 0; i < n_jobs; ++i)
        if (!jobs[i].src || !jobs[i].dst || jobs[i].n <= 0 || jobs[i].parts <= 0 || jobs[i].transpose_rows < 0 ||
            (jobs[i].transpose_rows && jobs[i].n % jobs[i].transpose_rows) || (jobs[i].parts > 1 && jobs[i].part_stride < jobs[i].n) ||
            ((uintptr_t)jobs[i].src & (jobs[i].src_bf16 ? 1 : 3)) || ((uintptr_t)jobs[i].dst & 3))
            return G2048_EINVAL;
    for (int base = 0; base < n_jobs; base += RJ_MAX) {
        JobTable T;
        T.n_jobs = n_jobs - base < RJ_MAX ? n_jobs - base : RJ_MAX;
        int blocks = 0;
        for (int i = 0; i < T.n_jobs; ++i) {
            T.job[i] = jobs[base + i];
            T.first_block[i] = blocks;
            const int cols = wide8(jobs[base + i]) ? RJ_WIDE8_COLS
                             : (jobs[base + i].parts <= RJ_WIDE_MAX_PARTS && !jobs[base + i].transpose_rows) ? RJ_WIDE_COLS : RJ_COLS;
            blocks += (jobs[base + i].n + cols - 1) / cols;
        }
        for (int i = T.n_jobs; i <= RJ_MAX; ++i) T.first_block[i] = blocks;
        hipLaunchKernelGGL(k_reduce_jobs, dim3((unsigned)blocks), dim3(RJ_TX * RJ_TY), 0, (hipStream_t)stream, T);
    }
    const hipError_t e = hipGetLastError();
    return e == hipSuccess ? 0 : -(1000 + (int)e);
}
