// Second stage of every column-sum in the update: out[c] = sum over the G partial rows of partial[g][c], fixed order.
// Shared by g2048_layernorm.hip (add+LN backward, activation backward, embedding backward, g2048_colsum) and
// g2048_linear.hip (the fused feed-forward backward); each translation unit gets its own copy in its anonymous namespace.
#ifndef G2048_COLSUM_FINAL_H
#define G2048_COLSUM_FINAL_H
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace {

// 32 columns per workgroup, 32 interleaved slices of the G partial rows per column (4 independent loads in flight per
// thread), combined through LDS in a fixed order
constexpr int CF_COLS = 32, CF_SLICES = 32;
__global__ void __launch_bounds__(CF_COLS * CF_SLICES)
k_colsum_final(const float *__restrict__ partial, int G, int N, float *__restrict__ out) {
    __shared__ float red[CF_SLICES][CF_COLS];
    const int cl = threadIdx.x % CF_COLS, slice = threadIdx.x / CF_COLS, c = blockIdx.x * CF_COLS + cl;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    if (c < N) {
        const float *p = partial + c;
        int g = slice;
        for (; g + 3 * CF_SLICES < G; g += 4 * CF_SLICES) {
            s0 += p[(int64_t)g * N];
            s1 += p[(int64_t)(g + CF_SLICES) * N];
            s2 += p[(int64_t)(g + 2 * CF_SLICES) * N];
            s3 += p[(int64_t)(g + 3 * CF_SLICES) * N];
        }
        for (; g < G; g += CF_SLICES) s0 += p[(int64_t)g * N];
    }
    red[slice][cl] = (s0 + s1) + (s2 + s3);
    __syncthreads();
    if (slice == 0 && c < N) {
        float t = 0.f;
        for (int k = 0; k < CF_SLICES; ++k) t += red[k][cl];
        out[c] = t;
    }
}


}  // namespace
#endif
