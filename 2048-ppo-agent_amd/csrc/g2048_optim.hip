// The optimiser step of the PPO update in two launches: gradient-norm clipping, GradScaler bookkeeping and AdamW over all
// parameters at once (reference: src/ppo/ppo_trainer.py:413-434 - scaler.unscale_, clip_grad_norm_, scaler.step(AdamW),
// scaler.update - which PyTorch runs as ~12 multi-tensor launches per minibatch, 0.25 ms of a 3.2 ms minibatch).
//
// Layout: gradients and both moments live in flat f32 buffers (the gradient buffer is the all-reduce bucket of a
// multi-GPU run); the parameters stay where the module owns them.  A chunk table cuts every parameter tensor into pieces of
// at most OPT_CHUNK elements; workgroup b owns chunk b in both kernels, so the summation order of the norm is fixed.
//   k_opt_sqnorm : partial[b] = sum over chunk b of (g * inv_scale)^2 (the UNSCALED gradients, as clip_grad_norm_ sees them after
//                  scaler.unscale_: summing the squares of the still loss-scaled values overflowed f32 once the scale passed ~2^60,
//                  long before torch's own element-wise inf check would skip a step)
//   k_opt_adamw  : every workgroup adds the partials in the same order -> total norm, found_inf, clip factor; then
//                  g' = (g * inv_scale) * clip;  p -= lr*wd*p;  m = lerp(m, g', 1-b1);  v = b2*v + (1-b2)*g'^2;
//                  p -= (lr / bc1) * m / (sqrt(v) / sqrt(bc2) + eps)                      (torch's fused AdamW arithmetic)
//                  unless found_inf
//                  workgroup 0 also advances the step counts and the scaler (scale *= backoff on inf, *= growth after
//                  growth_interval clean steps, as scaler.update() does): finish()
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/g2048.h"

namespace {

constexpr int OPT_THREADS = 256;
constexpr int OPT_VEC = 4;
constexpr int OPT_CHUNK = G2048_OPT_CHUNK;  // elements per workgroup
static_assert(OPT_CHUNK % (OPT_THREADS * OPT_VEC) == 0, "chunk = whole float4 passes");

__device__ __forceinline__ float block_sum(float v, float *lds) {
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
    const int w = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) lds[w] = v;
    __syncthreads();
    float s = 0.f;
    for (int i = 0; i < OPT_THREADS / 64; ++i) s += lds[i];  // same order in every thread
    __syncthreads();
    return s;
}

struct StepArgs {
    g2048_opt_group groups[G2048_OPT_MAX_GROUPS];
    float max_grad_norm;  // <= 0: no clipping
    float growth, backoff;
    int growth_interval;
};

// per-group constants of one step, derived once (by workgroup 0 of k_opt_sqnorm) from the f64 hyper-parameters and the step
// count: the two f64 pow calls of the bias corrections cost microseconds and must not sit in every workgroup of the update
struct Derived { float step_size, inv_bc2_sqrt, lr_wd, w1, b2, w2, eps, inv_scale; };  // inv_scale: of the scaler BEFORE this step's update

__global__ void __launch_bounds__(OPT_THREADS)
k_opt_sqnorm(const g2048_opt_chunk *__restrict__ chunks, const float *__restrict__ grads, float *__restrict__ partial, StepArgs A,
             int n_groups, const float *__restrict__ steps, Derived *__restrict__ derived, const float *__restrict__ scale) {
    __shared__ float lds[OPT_THREADS / 64];
    if (blockIdx.x == 0 && threadIdx.x < n_groups) {
        const g2048_opt_group G = A.groups[threadIdx.x];
        const double t = (double)steps[0] + 1.0;  // (all step counts are equal: every parameter steps on every non-skipped call;
                                                    //  FlatAdamWStep.adopt_state asserts it for loaded states)
        const double bc1 = 1.0 - pow(G.beta1, t), bc2 = 1.0 - pow(G.beta2, t);
        Derived d;
        // hyper-parameters arrive as f64 (what torch.optim holds) and are rounded once, after the f64 arithmetic on them
        d.step_size = (float)(G.lr / bc1);
        d.inv_bc2_sqrt = (float)(1.0 / sqrt(bc2));
        d.lr_wd = (float)(G.lr * G.weight_decay);
        d.w1 = (float)(1.0 - G.beta1);
        d.b2 = (float)G.beta2;
        d.w2 = (float)(1.0 - G.beta2);
        d.eps = (float)G.eps;
        d.inv_scale = scale ? (float)(1.0 / (double)*scale) : 1.f;  // scaler.unscale_: grads *= scale.double().reciprocal().float()
        derived[threadIdx.x] = d;
    }
    const g2048_opt_chunk c = chunks[blockIdx.x];
    const float *g = grads + c.offset;
    const float inv_scale = scale ? (float)(1.0 / (double)*scale) : 1.f;  // scaler.unscale_: grads *= scale.double().reciprocal().float()
    float s = 0.f;
    for (int i = threadIdx.x * OPT_VEC; i < c.n; i += OPT_THREADS * OPT_VEC) {
        if (i + OPT_VEC <= c.n) {
            float4 v = *reinterpret_cast<const float4 *>(g + i);
            v.x *= inv_scale; v.y *= inv_scale; v.z *= inv_scale; v.w *= inv_scale;
            s += v.x * v.x + v.y * v.y + v.z * v.z + v.w * v.w;
        } else {
            for (int k = i; k < c.n; ++k) s += (g[k] * inv_scale) * (g[k] * inv_scale);
        }
    }
    s = block_sum(s, lds);
    if (threadIdx.x == 0) partial[blockIdx.x] = s;
}

__device__ __forceinline__ void adamw1(float &p, float g, float &m, float &v, float lr_wd, float w1, float b2, float w2,
                                       float step_size, float inv_bc2_sqrt, float eps) {
    p -= lr_wd * p;
    const float d = g - m;
    m = (w1 < 0.5f) ? m + w1 * d : g - d * (1.f - w1);  // at::lerp
    v = b2 * v + w2 * g * g;
    const float denom = sqrtf(v) * inv_bc2_sqrt + eps;
    p -= step_size * m / denom;
}

// bf16 shadows of the parameters the update path multiplies with (dense copy and, for the weights whose input gradient is
// computed by our own GEMM, a transposed copy): rewritten here, in the kernel that changes the parameter, instead of by cast
// and transpose kernels in front of every forward
__device__ __forceinline__ uint16_t to_bf16(float f) {
    const __bf16 b = (__bf16)f;
    return *reinterpret_cast<const uint16_t *>(&b);
}
// offset of element (row, col) of a [rows][cols] matrix in the fragment-packed layout (include/g2048.h)
__device__ __forceinline__ int64_t packed_off(int64_t row, int64_t col, int64_t cols) {
    return ((((row >> 5) * (cols >> 4) + (col >> 4)) * 2 + ((col >> 3) & 1)) * 32 + (row & 31)) * 8 + (col & 7);
}
// `tiled`: the transposed copies of this chunk go through LDS (see k_opt_adamw) and is not written here
__device__ __forceinline__ void refresh_shadow(const g2048_opt_chunk &c, int i, float a, float b, float cc, float d, int n, bool tiled,
                                               uint16_t *stage) {
    const float vals[4] = {a, b, cc, d};
    const int64_t e = (int64_t)c.e0 + i;
    if (tiled) {
        for (int q = 0; q < n; ++q) stage[i + q] = to_bf16(vals[q]);
    }
    if (c.shadow_p) {  // fragment-packed copy: 4 consecutive columns of one row are 8 contiguous bytes there as well
        uint16_t *s = reinterpret_cast<uint16_t *>(c.shadow_p);
        const int64_t row = e / c.cols, col = e - row * c.cols;
        if (n == 4 && !(col & 3)) {
            *reinterpret_cast<uint2 *>(s + packed_off(row, col, c.cols)) =
                make_uint2((uint32_t)to_bf16(a) | ((uint32_t)to_bf16(b) << 16), (uint32_t)to_bf16(cc) | ((uint32_t)to_bf16(d) << 16));
        } else {
            for (int q = 0; q < n; ++q) {
                const int64_t eq = e + q, rq = eq / c.cols;
                s[packed_off(rq, eq - rq * c.cols, c.cols)] = to_bf16(vals[q]);
            }
        }
    }
    if (c.shadow_tp && !tiled) {
        uint16_t *t = reinterpret_cast<uint16_t *>(c.shadow_tp);
        for (int q = 0; q < n; ++q) {
            const int64_t eq = e + q, rq = eq / c.cols;
            t[packed_off(eq - rq * c.cols, rq, c.rows)] = to_bf16(vals[q]);
        }
    }
    if (c.shadow) {
        uint16_t *s = reinterpret_cast<uint16_t *>(c.shadow) + e;
        if (n == 4 && !((uintptr_t)s & 7)) {
            *reinterpret_cast<uint2 *>(s) = make_uint2((uint32_t)to_bf16(a) | ((uint32_t)to_bf16(b) << 16),
                                                       (uint32_t)to_bf16(cc) | ((uint32_t)to_bf16(d) << 16));
        } else {
            for (int q = 0; q < n; ++q) s[q] = to_bf16(vals[q]);
        }
    }
    if (c.shadow_t && !tiled) {
        uint16_t *t = reinterpret_cast<uint16_t *>(c.shadow_t);
        for (int q = 0; q < n; ++q) {
            const int64_t eq = e + q, r = eq / c.cols, col = eq - r * c.cols;
            t[col * c.rows + r] = to_bf16(vals[q]);
        }
    }
}

// Bookkeeping of one step, by workgroup 0 of k_opt_adamw: step counts, scaler state (scale *= backoff on inf, *= growth after
// growth_interval clean steps, as scaler.update() does), info.  Nothing of the same launch reads what it writes: the step counts are
// read by k_opt_sqnorm only, the scale reaches the other workgroups through Derived.inv_scale.  (Rounds 2-3: a launch of its own,
// k_opt_finish, ~4.5 us of every minibatch.  A "last workgroup done" counter is NOT what this is: that needed a device-scope release
// fence per workgroup, each writing back its XCD's L2 - measured 114 us for the update kernel instead of 20.)
struct FinishArgs {
    float growth, backoff;
    int growth_interval, n_steps;
    float *steps;
    int32_t *growth_tracker;
    float *info;
};
__device__ __forceinline__ void finish(const FinishArgs &F, float total, bool found_inf, float *scale) {
    if (!found_inf)
        for (int i = threadIdx.x; i < F.n_steps; i += OPT_THREADS) F.steps[i] += 1.f;  // one count per parameter, as torch keeps them
    if (threadIdx.x == 0) {
        if (F.info) {
            F.info[0] = sqrtf(total);  // gradient norm (unscaled) before clipping: what clip_grad_norm_ returns
            F.info[1] = found_inf ? 1.f : 0.f;
        }
        if (scale) {
            if (found_inf) {
                *scale *= F.backoff;
                *F.growth_tracker = 0;
            } else {
                const int32_t ok = *F.growth_tracker + 1;
                if (ok == F.growth_interval) {
                    *scale *= F.growth;
                    *F.growth_tracker = 0;
                } else {
                    *F.growth_tracker = ok;
                }
            }
        }
    }
}

__global__ void __launch_bounds__(OPT_THREADS)
k_opt_adamw(const g2048_opt_chunk *__restrict__ chunks, int n_chunks, const float *__restrict__ grads, float *__restrict__ exp_avg,
            float *__restrict__ exp_avg_sq, const float *__restrict__ partial, const Derived *__restrict__ derived, float max_grad_norm,
            float *__restrict__ scale, FinishArgs F) {
    __shared__ float lds[OPT_THREADS / 64];
    __shared__ float sh[2];
    __shared__ uint16_t stage[OPT_CHUNK];
    // total of the partials, identical in every workgroup (fixed order: strided per thread, then the block tree)
    float s = 0.f;
    for (int i = threadIdx.x; i < n_chunks; i += OPT_THREADS) s += partial[i];
    const float total = block_sum(s, lds);
    const g2048_opt_chunk c = chunks[blockIdx.x];
    const Derived G = derived[c.group];
    // Transposed shadow of a full chunk that covers R = 2, 4 or 8 whole rows of a [rows][cols] weight: the chunk's bf16 values
    // are staged in LDS and written as one 2R-byte run per column (R consecutive rows of the [cols][rows] copy) instead of one
    // scattered 2-byte store per element (round 2: 22 -> 30 us for the step with one transposed weight per layer; round 3 adds
    // the transposed copies the fused CLS tail's backward multiplies with)
    const bool any_t = c.shadow_t || c.shadow_tp;
    const int rows_in_chunk = (any_t && c.cols > 0) ? OPT_CHUNK / c.cols : 0;
    const bool tiled = any_t && c.n == OPT_CHUNK && c.cols > 0 && OPT_CHUNK % c.cols == 0 && c.e0 % c.cols == 0 &&
                       (rows_in_chunk == 2 || rows_in_chunk == 4 || rows_in_chunk == 8) && c.rows % rows_in_chunk == 0 &&
                       !(((uintptr_t)c.shadow_t | (uintptr_t)c.shadow_tp) & 15);
    if (threadIdx.x == 0) {
        const float inv_scale = G.inv_scale;  // (of k_opt_sqnorm's making: workgroup 0 below rewrites *scale while others still start)
        // the norm of the unscaled gradients (k_opt_sqnorm squared g * inv_scale); non-finite anywhere makes the total non-finite
        const float norm = sqrtf(total);
        float clip = 1.f;
        if (max_grad_norm > 0.f) {
            clip = max_grad_norm / (norm + 1e-6f);  // torch.nn.utils.clip_grad_norm_
            if (clip > 1.f) clip = 1.f;
        }
        sh[0] = inv_scale;
        sh[1] = clip;
    }
    __syncthreads();
    const bool found_inf = scale != nullptr && !(fabsf(total) <= 3.4028234664e38f);  // inf or nan; without a scaler torch steps anyway
    if (blockIdx.x == 0) finish(F, total, found_inf, scale);
    if (!found_inf) {
        const float inv_scale = sh[0], clip = sh[1], step_size = G.step_size, inv_bc2_sqrt = G.inv_bc2_sqrt;
        const float lr_wd = G.lr_wd, w1 = G.w1, b2 = G.b2, w2 = G.w2, eps = G.eps;
        float *p = c.param;
        const float *g = grads + c.offset;
        float *m = exp_avg + c.offset, *v = exp_avg_sq + c.offset;
        for (int i = threadIdx.x * OPT_VEC; i < c.n; i += OPT_THREADS * OPT_VEC) {
            if (i + OPT_VEC <= c.n) {
                float4 pv = *reinterpret_cast<float4 *>(p + i), mv = *reinterpret_cast<float4 *>(m + i),
                       vv = *reinterpret_cast<float4 *>(v + i);
                const float4 gv = *reinterpret_cast<const float4 *>(g + i);
                adamw1(pv.x, (gv.x * inv_scale) * clip, mv.x, vv.x, lr_wd, w1, b2, w2, step_size, inv_bc2_sqrt, eps);
                adamw1(pv.y, (gv.y * inv_scale) * clip, mv.y, vv.y, lr_wd, w1, b2, w2, step_size, inv_bc2_sqrt, eps);
                adamw1(pv.z, (gv.z * inv_scale) * clip, mv.z, vv.z, lr_wd, w1, b2, w2, step_size, inv_bc2_sqrt, eps);
                adamw1(pv.w, (gv.w * inv_scale) * clip, mv.w, vv.w, lr_wd, w1, b2, w2, step_size, inv_bc2_sqrt, eps);
                *reinterpret_cast<float4 *>(p + i) = pv;
                *reinterpret_cast<float4 *>(m + i) = mv;
                *reinterpret_cast<float4 *>(v + i) = vv;
                if (c.shadow || c.shadow_p || any_t) refresh_shadow(c, i, pv.x, pv.y, pv.z, pv.w, 4, tiled, stage);
            } else {
                for (int k = i; k < c.n; ++k) {
                    float pk = p[k], mk = m[k], vk = v[k];
                    adamw1(pk, (g[k] * inv_scale) * clip, mk, vk, lr_wd, w1, b2, w2, step_size, inv_bc2_sqrt, eps);
                    p[k] = pk; m[k] = mk; v[k] = vk;
                    if (c.shadow || c.shadow_p || any_t) refresh_shadow(c, k, pk, 0.f, 0.f, 0.f, 1, tiled, stage);
                }
            }
        }
        if (tiled) {  // (uniform per workgroup)
            __syncthreads();
            const int R = rows_in_chunk, r0 = c.e0 / c.cols;
            // the run of R consecutive rows of one column: 2R contiguous bytes of the [cols][rows] copy, and (R <= 8, r0 % R == 0)
            // of its fragment-packed form too
            for (int col = threadIdx.x; col < c.cols; col += OPT_THREADS) {
                const uint16_t *src = stage + col;
                const int cs = c.cols;
                uint32_t w[4] = {0u, 0u, 0u, 0u};
                if (R == 8) {
                    for (int q = 0; q < 4; ++q) w[q] = src[2 * q * cs] | ((uint32_t)src[(2 * q + 1) * cs] << 16);
                } else if (R == 4) {
                    for (int q = 0; q < 2; ++q) w[q] = src[2 * q * cs] | ((uint32_t)src[(2 * q + 1) * cs] << 16);
                } else {
                    w[0] = src[0] | ((uint32_t)src[cs] << 16);
                }
                for (int which = 0; which < 2; ++which) {
                    uint16_t *base = reinterpret_cast<uint16_t *>(which ? c.shadow_tp : c.shadow_t);
                    if (!base) continue;
                    uint16_t *dst = base + (which ? packed_off(col, r0, c.rows) : (int64_t)col * c.rows + r0);
                    if (R == 8) *reinterpret_cast<uint4 *>(dst) = make_uint4(w[0], w[1], w[2], w[3]);
                    else if (R == 4) *reinterpret_cast<uint2 *>(dst) = make_uint2(w[0], w[1]);
                    else *reinterpret_cast<uint32_t *>(dst) = w[0];
                }
            }
        }
    }
}

}  // namespace

extern "C" int g2048_opt_step(const g2048_opt_chunk *chunks, int n_chunks, const float *grads, float *exp_avg, float *exp_avg_sq,
                              const g2048_opt_group *groups, int n_groups, float max_grad_norm, float *steps, int n_steps, float *scale,
                              int32_t *growth_tracker, float growth, float backoff, int growth_interval, float *workspace,
                              float *info, void *stream) {
    if (!chunks || n_chunks <= 0 || !grads || !exp_avg || !exp_avg_sq || !groups || n_groups <= 0 ||
        n_groups > G2048_OPT_MAX_GROUPS || !steps || n_steps <= 0 || !workspace || (scale && !growth_tracker) ||
        (((uintptr_t)grads | (uintptr_t)exp_avg | (uintptr_t)exp_avg_sq | (uintptr_t)workspace) & 15))
        return G2048_EINVAL;
    StepArgs A;
    for (int i = 0; i < G2048_OPT_MAX_GROUPS; ++i) A.groups[i] = groups[i < n_groups ? i : 0];
    // workspace: [n_chunks] partial sums, then the per-group constants
    float *partial = workspace;
    Derived *derived = reinterpret_cast<Derived *>(workspace + ((n_chunks + 3) & ~3));
    hipLaunchKernelGGL(k_opt_sqnorm, dim3((unsigned)n_chunks), dim3(OPT_THREADS), 0, (hipStream_t)stream, chunks, grads, partial, A,
                       n_groups, steps, derived, scale);
    FinishArgs F;
    F.growth = growth, F.backoff = backoff, F.growth_interval = growth_interval, F.n_steps = n_steps;
    F.steps = steps, F.growth_tracker = growth_tracker, F.info = info;
    hipLaunchKernelGGL(k_opt_adamw, dim3((unsigned)n_chunks), dim3(OPT_THREADS), 0, (hipStream_t)stream, chunks, n_chunks, grads,
                       exp_avg, exp_avg_sq, partial, derived, max_grad_norm, scale, F);
    const hipError_t e = hipGetLastError();
    return e == hipSuccess ? 0 : -(1000 + (int)e);
}

extern "C" int64_t g2048_opt_workspace_floats(int n_chunks) {
    return n_chunks <= 0 ? 0 : (int64_t)((n_chunks + 3) & ~3) + G2048_OPT_MAX_GROUPS * (int64_t)(sizeof(Derived) / sizeof(float));
}
