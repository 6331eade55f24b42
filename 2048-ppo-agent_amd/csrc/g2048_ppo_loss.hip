// Clipped-surrogate PPO loss of one minibatch: forward, the five logged means and the gradient w.r.t. the policy logits
// and the value estimates in ONE launch.
//
// Reference: PPOTrainer._compute_ppo_loss (src/ppo/ppo_trainer.py:251-314) on top of PPOAgent.evaluate_actions
// (src/ppo/ppo_agent.py:159-191): logits - 1e8 * (1 - mask) -> Categorical(logits) -> log_prob(action), entropy;
// ratio = exp(new - old); policy = -min(ratio * A, clamp(ratio, 1 - eps, 1 + eps) * A); value = (V - R)^2;
// total = mean(policy + c_v * value - c_e * entropy).  In PyTorch that is ~35 elementwise/reduction kernels forward and
// as many backward over 2048 x 4 numbers: pure launch latency.  Here: one workgroup, one sample per thread per pass,
// sums combined in a fixed order (bit-reproducible).  The gradients follow autograd's conventions exactly: torch.min
// splits ties evenly between its arguments and clamp passes the gradient on the closed interval [1 - eps, 1 + eps].
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/g2048.h"

namespace {

constexpr int THREADS = 1024, NSUM = 5;

__device__ __forceinline__ float ld(const void *p, int64_t i, int bf16) {
    if (bf16) return __uint_as_float((uint32_t)((const uint16_t *)p)[i] << 16);
    return ((const float *)p)[i];
}
__device__ __forceinline__ void st(void *p, int64_t i, int bf16, float v) {
    if (bf16) {
        const __bf16 b = (__bf16)v;
        ((uint16_t *)p)[i] = *reinterpret_cast<const uint16_t *>(&b);
    } else {
        ((float *)p)[i] = v;
    }
}

__global__ void __launch_bounds__(THREADS)
k_ppo_loss(const void *__restrict__ logits, int logits_bf16, const void *__restrict__ values, int values_bf16,
           const uint8_t *__restrict__ actions, const uint8_t *__restrict__ mask_bits, const float *__restrict__ old_logp,
           const float *__restrict__ adv, const float *__restrict__ ret, int64_t M, float clip_eps, float c_value,
           float c_entropy, float *__restrict__ new_logp, float *__restrict__ sums, void *__restrict__ dlogits,
           void *__restrict__ dvalues, const float *__restrict__ grad_scale, double *__restrict__ running) {
    __shared__ float red[NSUM][THREADS / 64];
    float acc[NSUM] = {0.f, 0.f, 0.f, 0.f, 0.f};  // policy, value, entropy loss, total, old - new log-prob
    const float inv_m = 1.0f / (float)M, lo = 1.0f - clip_eps, hi = 1.0f + clip_eps;
    // gradients come out multiplied by the loss scale (GradScaler's device scalar) when one is given: what backward would
    // otherwise do with two more launches
    const float g_m = grad_scale ? inv_m * *grad_scale : inv_m;
    for (int64_t i = threadIdx.x; i < M; i += THREADS) {
        const uint32_t mb = mask_bits ? mask_bits[i] : 0xFu;
        float z[4], zmax = -INFINITY;
        for (int j = 0; j < 4; ++j) {
            const float l = ld(logits, 4 * i + j, logits_bf16);
            z[j] = ((mb >> j) & 1u) ? l : l - 1e8f;
            zmax = fmaxf(zmax, z[j]);
        }
        float se = 0.f;
        for (int j = 0; j < 4; ++j) se += expf(z[j] - zmax);
        const float lse = zmax + logf(se);
        float lp[4], p[4], ent = 0.f;
        for (int j = 0; j < 4; ++j) {
            lp[j] = z[j] - lse;
            p[j] = expf(lp[j]);
            ent -= lp[j] * p[j];
        }
        const int a = actions[i] & 3;
        const float nlp = lp[a], olp = old_logp[i], A = adv[i];
        const float ratio = expf(nlp - olp);
        const float s1 = ratio * A, s2 = fminf(fmaxf(ratio, lo), hi) * A;
        const float pl = -fminf(s1, s2);
        const float v = ld(values, i, values_bf16), dv = v - ret[i];
        const float vl = dv * dv, el = -ent;
        const float tot = pl + c_value * vl + c_entropy * el;
        new_logp[i] = nlp;
        acc[0] += pl; acc[1] += vl; acc[2] += el; acc[3] += tot; acc[4] += olp - nlp;
        // d(policy loss)/d(new log-prob)
        const float in_range = (ratio >= lo && ratio <= hi) ? 1.0f : 0.0f;
        const float w1 = s1 < s2 ? 1.0f : (s1 == s2 ? 0.5f : 0.0f), w2 = 1.0f - w1;
        const float g_lp = -(w1 + w2 * in_range) * A * ratio;
        for (int j = 0; j < 4; ++j) {
            const float dz = g_lp * ((j == a ? 1.0f : 0.0f) - p[j]) + c_entropy * p[j] * (lp[j] + ent);
            st(dlogits, 4 * i + j, logits_bf16, dz * g_m);
        }
        st(dvalues, i, values_bf16, 2.0f * c_value * dv * g_m);
    }
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    for (int k = 0; k < NSUM; ++k) {
        float s = acc[k];
        for (int m = 32; m >= 1; m >>= 1) s += __shfl_xor(s, m);
        if (lane == 0) red[k][w] = s;
    }
    __syncthreads();
    if (threadIdx.x < NSUM) {
        float s = 0.f;
        for (int ww = 0; ww < THREADS / 64; ++ww) s += red[threadIdx.x][ww];
        sums[threadIdx.x] = s * inv_m;
        if (running) running[threadIdx.x] += (double)(s * inv_m);
    }
}

}  // namespace

extern "C" int g2048_ppo_loss(const void *logits, int logits_bf16, const void *values, int values_bf16, const uint8_t *actions,
                              const uint8_t *mask_bits, const float *old_logp, const float *adv, const float *ret, int64_t M,
                              float clip_eps, float c_value, float c_entropy, float *new_logp, float *sums, void *dlogits,
                              void *dvalues, const float *grad_scale, double *running, void *stream) {
    if (!logits || !values || !actions || !old_logp || !adv || !ret || !new_logp || !sums || !dlogits || !dvalues || M <= 0 ||
        M > G2048_PPO_LOSS_MAX_BATCH)
        return G2048_EINVAL;
    hipLaunchKernelGGL(k_ppo_loss, dim3(1), dim3(THREADS), 0, (hipStream_t)stream, logits, logits_bf16, values, values_bf16, actions,
                       mask_bits, old_logp, adv, ret, M, clip_eps, c_value, c_entropy, new_logp, sums, dlogits, dvalues, grad_scale, running);
    const hipError_t e = hipGetLastError();
    return e == hipSuccess ? 0 : -(1000 + (int)e);
}

// ---- minibatch gather ------------------------------------------------------------------------------------------------
// One launch instead of six index_select kernels (+ six copies into the hipGraph's static inputs): sample idx[i] of the
// rollout buffer -> row i of the minibatch.  Reference: the DataLoader collation over PPODataset.__getitem__
// (src/ppo/data_loader.py) - here the buffer never leaves the device.
namespace {

__global__ void __launch_bounds__(256)
k_gather_minibatch(const int64_t *__restrict__ idx, int64_t M, int64_t N, const uint4 *__restrict__ boards,
                   const uint8_t *__restrict__ actions, const uint8_t *__restrict__ masks, const float *__restrict__ logp,
                   const float *__restrict__ adv, const float *__restrict__ ret, uint4 *__restrict__ o_boards,
                   uint8_t *__restrict__ o_actions, uint8_t *__restrict__ o_masks, float *__restrict__ o_logp,
                   float *__restrict__ o_adv, float *__restrict__ o_ret) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= M) return;
    int64_t s = idx[i];
    s = s < 0 ? 0 : (s >= N ? N - 1 : s);  // never read out of bounds on a bad index
    o_boards[i] = boards[s];
    o_actions[i] = actions[s];
    o_masks[i] = masks[s];
    o_logp[i] = logp[s];
    o_adv[i] = adv[s];
    o_ret[i] = ret[s];
}

}  // namespace

extern "C" int g2048_gather_minibatch(const int64_t *idx, int64_t M, int64_t N, const uint8_t *boards, const uint8_t *actions,
                                      const uint8_t *masks, const float *logp, const float *adv, const float *ret,
                                      uint8_t *o_boards, uint8_t *o_actions, uint8_t *o_masks, float *o_logp, float *o_adv,
                                      float *o_ret, void *stream) {
    if (!idx || !boards || !actions || !masks || !logp || !adv || !ret || !o_boards || !o_actions || !o_masks || !o_logp ||
        !o_adv || !o_ret || M <= 0 || N <= 0 || (((uintptr_t)boards | (uintptr_t)o_boards) & 15))
        return G2048_EINVAL;
    hipLaunchKernelGGL(k_gather_minibatch, dim3((unsigned)((M + 255) / 256)), dim3(256), 0, (hipStream_t)stream, idx, M, N,
                       (const uint4 *)boards, actions, masks, logp, adv, ret, (uint4 *)o_boards, o_actions, o_masks, o_logp, o_adv,
                       o_ret);
    const hipError_t e = hipGetLastError();
    return e == hipSuccess ? 0 : -(1000 + (int)e);
}
