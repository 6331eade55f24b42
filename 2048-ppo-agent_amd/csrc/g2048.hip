// libg2048.so -- HIP kernels (gfx950) and the C ABI of include/g2048.h.
//
// Layout in HBM: boards are 16 B per env, so one `global_load_dwordx4` per lane and 1 KiB per
// wave-instruction, perfectly coalesced; byte/float side arrays are SoA so a wave touches 64 or 256
// contiguous bytes per instruction.  Trajectories are step-major ([t][env]) for the same reason.
// Everything is integer/byte work per lane -- no LDS, no MFMA; the bound is HBM (and, close behind it,
// VALU issue for the five threefry blocks per step).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/g2048.h"
#include "g2048_device.h"

using namespace g2048;

namespace {

constexpr int kBlock = 256;

struct StepKeyTable {
    u32 k[G2048_MAX_FUSED_STEPS][4];  // act sub-key, step sub-key per lock-step (wave-uniform -> SGPRs)
};

__device__ __forceinline__ Board load_board(const uint8_t *p, int64_t i) {
    const uint4 v = reinterpret_cast<const uint4 *>(p)[i];
    Board b;
    b.r[0] = v.x; b.r[1] = v.y; b.r[2] = v.z; b.r[3] = v.w;
    return b;
}
__device__ __forceinline__ void store_board(uint8_t *p, int64_t i, const Board &b) {
    reinterpret_cast<uint4 *>(p)[i] = make_uint4(b.r[0], b.r[1], b.r[2], b.r[3]);
}

// ---------------------------------------------------------------------------------------- RNG
template <int MODE>
__global__ void __launch_bounds__(kBlock) k_split(u32 k0, u32 k1, u32 *out, int64_t n) {
    const int64_t j = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (j >= n) return;
    u32 a, b;
    split_at<MODE>(k0, k1, (u32)n, (u32)j, a, b);
    reinterpret_cast<uint2 *>(out)[j] = make_uint2(a, b);
}

// ---------------------------------------------------------------------------------------- env
template <int MODE>
__global__ void __launch_bounds__(kBlock) k_init(const u32 *keys, uint8_t *boards, uint8_t *masks, uint8_t *done,
                                                 int64_t B) {
    const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (i >= B) return;
    const uint2 k = reinterpret_cast<const uint2 *>(keys)[i];
    Board bd;
    u32 m;
    env_init<MODE>(bd, m, k.x, k.y);
    store_board(boards, i, bd);
    masks[i] = (uint8_t)m;
    done[i] = 0;
}

template <int MODE>
__global__ void __launch_bounds__(kBlock) k_step(uint8_t *boards, uint8_t *masks, uint8_t *done,
                                                 const int32_t *actions, const u32 *keys, float *rewards, int64_t B) {
    const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (i >= B) return;
    Board bd = load_board(boards, i);
    const uint2 k = reinterpret_cast<const uint2 *>(keys)[i];
    const u32 a = (u32)actions[i] & 3u;
    u32 m = masks[i], d = done[i];
    const float r = env_step<MODE>(bd, m, d, a, k.x, k.y);
    store_board(boards, i, bd);
    rewards[i] = r;
    masks[i] = (uint8_t)m;
    done[i] = (uint8_t)d;
}

// obs[e][cell][k] = (board[e][cell] == k), 31 bytes per cell, 496 per board; one thread per output dword
__global__ void __launch_bounds__(kBlock) k_observe(const uint8_t *boards, uint8_t *obs, int64_t B) {
    const int64_t w = (int64_t)blockIdx.x * kBlock + threadIdx.x;  // dword index, 124 per board
    if (w >= B * 124) return;
    const int64_t e = w / 124;
    const int byte0 = (int)(w - e * 124) * 4;
    u32 out = 0;
    for (int q = 0; q < 4; ++q) {
        const int idx = byte0 + q, cell = idx / 31, k = idx - cell * 31;
        out |= (u32)(boards[e * 16 + cell] == k) << (8 * q);
    }
    reinterpret_cast<u32 *>(obs)[w] = out;
}

// ---------------------------------------------------------------------------------------- policies
__global__ void __launch_bounds__(kBlock) k_act_drul(const uint8_t *masks, int32_t *actions, int64_t B) {
    const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (i >= B) return;
    actions[i] = (int32_t)policy_drul(masks[i]);
}

template <int MODE>
__global__ void __launch_bounds__(kBlock) k_act_random(const u32 *keys, const uint8_t *masks, int32_t *actions,
                                                       float *logp, int64_t B) {
    const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (i >= B) return;
    const uint2 k = reinterpret_cast<const uint2 *>(keys)[i];
    float lp;
    actions[i] = (int32_t)policy_random<MODE>(k.x, k.y, masks[i], lp);
    logp[i] = lp;
}

template <int MODE>
__global__ void __launch_bounds__(kBlock) k_act_logits(const u32 *keys, const float *logits, const uint8_t *masks,
                                                       int use_mask, int sample, int32_t *actions, float *logp,
                                                       int64_t B) {
    const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (i >= B) return;
    const uint2 k = reinterpret_cast<const uint2 *>(keys)[i];
    const float4 lg = reinterpret_cast<const float4 *>(logits)[i];
    const float raw[4] = {lg.x, lg.y, lg.z, lg.w};
    float lp;
    actions[i] = (int32_t)policy_logits<MODE>(k.x, k.y, raw, masks[i], use_mask != 0, sample != 0, lp);
    logp[i] = lp;
}

// ---------------------------------------------------------------------------------------- fused engine
template <int MODE>
__global__ void __launch_bounds__(kBlock) k_reset_fused(u32 s0, u32 s1, uint8_t *boards, uint8_t *masks,
                                                        uint8_t *done, int32_t *ep_len, int64_t B, u32 B_total,
                                                        u32 env0) {
    const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (i >= B) return;
    u32 k0, k1, m;
    split_at<MODE>(s0, s1, B_total, env0 + (u32)i, k0, k1);
    Board bd;
    env_init<MODE>(bd, m, k0, k1);
    store_board(boards, i, bd);
    masks[i] = (uint8_t)m;
    done[i] = 0;
    ep_len[i] = 0;
}

// *live_count += lanes of this workgroup that are still running; live_count NULL: nobody asked (the host polls every few
// lock-steps only).  ONE atomic per workgroup, none for a workgroup without live lanes: adds to one address serialise at
// ~12 ns each at the memory side, so one per WAVE (rounds 1-3) made the counter, not the arithmetic, the length of a launch
// from ~2^20 boards on (757 us at 2^22 boards = 65 536 atomics).  (All lanes of the workgroup reach this call.)
__device__ __forceinline__ void count_live(bool lane_live, u32 *live_count) {
    if (!live_count) return;  // (uniform)
    __shared__ u32 wave_live[kBlock / 64];
    const unsigned long long b = __ballot(lane_live);
    if ((threadIdx.x & 63) == 0) wave_live[threadIdx.x >> 6] = (u32)__popcll(b);
    __syncthreads();
    if (threadIdx.x == 0) {
        u32 n = 0;
        for (int w = 0; w < kBlock / 64; ++w) n += wave_live[w];
        if (n) atomicAdd(live_count, n);
    }
}

// Persistent multi-step rollout with a fused naive policy: the board lives in four VGPRs across all
// n_steps; per live env-step the only HBM traffic is the trajectory write (16 B board + 1 B meta + 4 B
// reward [+ 4 B log-prob]).
template <int MODE, int POLICY>
__global__ void __launch_bounds__(kBlock) k_rollout_fused(const StepKeyTable keys, int n_steps, int64_t t0,
                                                          uint8_t *boards, uint8_t *masks, uint8_t *done,
                                                          int32_t *ep_len, uint8_t *tr_boards, uint8_t *tr_meta,
                                                          float *tr_rewards, float *tr_logp, int64_t B, u32 B_total,
                                                          u32 env0, int fill_frozen, u32 *live_count) {
    const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    const bool in_range = i < B;
    Board bd;
    u32 m = 0xF, d = 1, len = 0;
    if (in_range) {
        bd = load_board(boards, i);
        m = masks[i];
        d = done[i];
        len = (u32)ep_len[i];
    } else {
        bd.r[0] = bd.r[1] = bd.r[2] = bd.r[3] = 0;
    }
    const u32 g = env0 + (u32)i;
    for (int s = 0; s < n_steps; ++s) {
        const bool work = in_range && (fill_frozen || d == 0);
        if (!fill_frozen && __all(d != 0)) break;  // every env of this wave has finished
        if (work) {
            u32 k0, k1, a;
            float lp = 0.0f;
            if (POLICY == G2048_POLICY_RANDOM) {
                split_at<MODE>(keys.k[s][0], keys.k[s][1], B_total, g, k0, k1);
                a = policy_random<MODE>(k0, k1, m, lp);
            } else {
                a = policy_drul(m);
            }
            split_at<MODE>(keys.k[s][2], keys.k[s][3], B_total, g, k0, k1);
            const int64_t o = (t0 + s) * B + i;
            store_board(tr_boards, o, bd);
            const u32 m_before = m;
            len += (d == 0);
            const float r = env_step<MODE>(bd, m, d, a, k0, k1);
            tr_meta[o] = (uint8_t)(a | (m_before << 2) | (d << 6));
            tr_rewards[o] = r;
            if (POLICY == G2048_POLICY_RANDOM && tr_logp) tr_logp[o] = lp;
        }
    }
    if (in_range) {
        store_board(boards, i, bd);
        masks[i] = (uint8_t)m;
        done[i] = (uint8_t)d;
        ep_len[i] = (int32_t)len;
    }
    count_live(in_range && d == 0, live_count);
}

// jax.random.fold_in(key, 0xFFFFFFFF): the per-step "reset" sub-key of the auto-reset mode, derived from the step
// sub-key without advancing the host's key chain (fold_in is one threefry block over the counter (0, data))
__device__ __forceinline__ void reset_subkey(u32 ss0, u32 ss1, u32 &r0, u32 &r1) { tf2x32(ss0, ss1, 0u, 0xFFFFFFFFu, r0, r1); }

// AUTO = 0: the reference's lock-step semantics (finished envs freeze; fill_frozen writes their frozen frames).
// AUTO = 1: fixed-horizon throughput mode (replaces the lock-step of src/runs/batch_runner.py:117): every
//   lane steps at every call; a lane whose step terminates is re-initialised at once from its own key
//   split(fold_in(step_sub, 0xFFFFFFFF), B_total)[g], so row t+1 of the trajectory starts its next episode.  The row of
//   the terminal step keeps done_after = 1 (the episode boundary for GAE); ep_len counts the steps of the running episode.
template <int MODE, int AUTO>
__global__ void __launch_bounds__(kBlock) k_policy_step(u32 as0, u32 as1, u32 ss0, u32 ss1, const float *logits,
                                                        const float *values, int use_mask, int sample, int64_t t,
                                                        uint8_t *boards, uint8_t *masks, uint8_t *done,
                                                        int32_t *ep_len, uint8_t *tr_boards, uint8_t *tr_meta,
                                                        float *tr_rewards, float *tr_logp, float *tr_values,
                                                        int64_t B, u32 B_total, u32 env0, int fill_frozen,
                                                        u32 *live_count) {
    const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    const bool in_range = i < B;
    u32 d = 1;
    if (in_range) {
        d = AUTO ? 0u : done[i];
        if (AUTO || fill_frozen || d == 0) {
            Board bd = load_board(boards, i);
            u32 m = masks[i];
            const u32 g = env0 + (u32)i;
            u32 k0, k1;
            split_at<MODE>(as0, as1, B_total, g, k0, k1);
            const float4 lg = reinterpret_cast<const float4 *>(logits)[i];
            const float raw[4] = {lg.x, lg.y, lg.z, lg.w};
            float lp;
            const u32 a = policy_logits<MODE>(k0, k1, raw, m, use_mask != 0, sample != 0, lp);
            split_at<MODE>(ss0, ss1, B_total, g, k0, k1);
            const int64_t o = t * B + i;
            store_board(tr_boards, o, bd);
            const u32 m_before = m;
            const bool was_live = d == 0;
            const float r = env_step<MODE>(bd, m, d, a, k0, k1);
            tr_meta[o] = (uint8_t)(a | (m_before << 2) | (d << 6));
            tr_rewards[o] = r;
            tr_logp[o] = lp;
            tr_values[o] = values[i];
            if (AUTO) {
                int32_t len = ep_len[i] + 1;
                if (d) {  // terminal step: the lane starts its next episode now
                    u32 r0, r1;
                    reset_subkey(ss0, ss1, r0, r1);
                    split_at<MODE>(r0, r1, B_total, g, k0, k1);
                    env_init<MODE>(bd, m, k0, k1);
                    len = 0;
                }
                store_board(boards, i, bd);
                masks[i] = (uint8_t)m;
                ep_len[i] = len;
            } else if (was_live) {
                store_board(boards, i, bd);
                masks[i] = (uint8_t)m;
                done[i] = (uint8_t)d;
                ep_len[i] += 1;
            }
        }
    }
    if (!AUTO) count_live(in_range && d == 0, live_count);
}

// ---------------------------------------------------------------------------------------- GAE / buffer
// One lane per env walks its column of the [T][B] trajectory backwards: every wave-instruction reads or
// writes 256 contiguous bytes.  Float32 ops in the reference's order, no contraction.
__global__ void __launch_bounds__(kBlock) k_gae_tb(const float *rew, const float *val, const int32_t *ep_len,
                                                   float *adv, float *ret, int64_t T, int64_t B, float g, float gl) {
    const int64_t e = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (e >= B) return;
    int64_t n = ep_len[e];
    n = n < T ? n : T;
    float last_gae = 0.0f, last_v = 0.0f;
    for (int64_t t = n - 1; t >= 0; --t) {
        const int64_t o = t * B + e;
        const float r = rew[o], v = val[o];
        const float delta = __fsub_rn(__fadd_rn(r, __fmul_rn(g, last_v)), v);
        last_gae = __fadd_rn(delta, __fmul_rn(gl, last_gae));
        adv[o] = last_gae;
        ret[o] = __fadd_rn(last_gae, v);
        last_v = v;
    }
}

// The same scan for fixed-horizon rollouts: every lane has T steps, episode boundaries are the done_after bits of tr_meta
// (the reference's reset at terminations[step]), and the scan starts from V(s_T) = last_val[e] -- the value of the state the
// horizon cut off (ignored when step T-1 was terminal).
__global__ void __launch_bounds__(kBlock) k_gae_tb_boot(const float *rew, const float *val, const uint8_t *meta,
                                                        const float *last_val, float *adv, float *ret, int64_t T,
                                                        int64_t B, float g, float gl) {
    const int64_t e = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (e >= B) return;
    float last_gae = 0.0f, last_v = last_val[e];
    for (int64_t t = T - 1; t >= 0; --t) {
        const int64_t o = t * B + e;
        if ((meta[o] >> 6) & 1u) {
            last_v = 0.0f;
            last_gae = 0.0f;
        }
        const float r = rew[o], v = val[o];
        const float delta = __fsub_rn(__fadd_rn(r, __fmul_rn(g, last_v)), v);
        last_gae = __fadd_rn(delta, __fmul_rn(gl, last_gae));
        adv[o] = last_gae;
        ret[o] = __fadd_rn(last_gae, v);
        last_v = v;
    }
}

// Flat buffer: each episode segment (ending at a termination flag, or at N-1) is scanned by the lane
// that owns its last element.
__global__ void __launch_bounds__(kBlock) k_gae_flat(const float *rew, const float *val, const uint8_t *term,
                                                     float *adv, float *ret, int64_t N, float g, float gl) {
    const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (i >= N) return;
    if (!(term[i] || i == N - 1)) return;
    float last_gae = 0.0f, last_v = 0.0f;
    int64_t t = i;
    do {
        const float r = rew[t], v = val[t];
        const float delta = __fsub_rn(__fadd_rn(r, __fmul_rn(g, last_v)), v);
        last_gae = __fadd_rn(delta, __fmul_rn(gl, last_gae));
        adv[t] = last_gae;
        ret[t] = __fadd_rn(last_gae, v);
        last_v = v;
        --t;
    } while (t >= 0 && !term[t]);
}

// keep-through-first-termination, env-major: one wave per env, lanes stride over its steps so the writes
// are contiguous (the strided reads are the cost of the reference's env-major order).
__global__ void __launch_bounds__(kBlock) k_compact(const uint8_t *tr_boards, const uint8_t *tr_meta,
                                                    const float *tr_rewards, const float *tr_logp,
                                                    const float *tr_values, const float *tr_adv, const float *tr_ret,
                                                    const int32_t *ep_len, const int64_t *offsets, uint8_t *out_boards,
                                                    uint8_t *out_actions, uint8_t *out_masks, float *out_rewards,
                                                    float *out_logp, float *out_values, float *out_adv, float *out_ret,
                                                    uint8_t *out_terms, int64_t T, int64_t B) {
    const int64_t e = ((int64_t)blockIdx.x * kBlock + threadIdx.x) >> 6;
    const int lane = threadIdx.x & 63;
    if (e >= B) return;
    int64_t n = ep_len[e];
    n = n < T ? n : T;
    const int64_t off = offsets[e];
    for (int64_t t = lane; t < n; t += 64) {
        const int64_t src = t * B + e, dst = off + t;
        reinterpret_cast<uint4 *>(out_boards)[dst] = reinterpret_cast<const uint4 *>(tr_boards)[src];
        const u32 meta = tr_meta[src];
        out_actions[dst] = (uint8_t)(meta & 3u);
        out_masks[dst] = (uint8_t)((meta >> 2) & 0xFu);
        out_terms[dst] = (uint8_t)((meta >> 6) & 1u);
        out_rewards[dst] = tr_rewards[src];
        if (out_logp) out_logp[dst] = tr_logp[src];
        if (out_values) out_values[dst] = tr_values[src];
        if (out_adv) out_adv[dst] = tr_adv[src];
        if (out_ret) out_ret[dst] = tr_ret[src];
    }
}

inline int finish() {
    const hipError_t e = hipGetLastError();
    return e == hipSuccess ? 0 : -(1000 + (int)e);
}
inline unsigned blocks_for(int64_t n) { return (unsigned)((n + kBlock - 1) / kBlock); }
inline bool bad_mode(int m) { return m != G2048_RNG_LEGACY && m != G2048_RNG_PARTITIONABLE; }
// global env indices and 2*j+1 must fit in 32 bits for the legacy split
constexpr int64_t kMaxEnvs = (int64_t)1 << 30;

}  // namespace

#define G2048_LAUNCH(kern, n, stream, ...) \
    hipLaunchKernelGGL(kern, dim3(blocks_for(n)), dim3(kBlock), 0, (hipStream_t)(stream), __VA_ARGS__)

extern "C" {

int g2048_abi_version(void) { return G2048_ABI_VERSION; }

int g2048_split(uint32_t key0, uint32_t key1, uint32_t *out_keys, int64_t n, int rng_mode, void *stream) {
    if (!out_keys || n <= 0 || n > kMaxEnvs || bad_mode(rng_mode)) return G2048_EINVAL;
    if (rng_mode) G2048_LAUNCH(k_split<1>, n, stream, key0, key1, out_keys, n);
    else G2048_LAUNCH(k_split<0>, n, stream, key0, key1, out_keys, n);
    return finish();
}

int g2048_chain_keys(uint32_t *key, uint32_t *subs, int64_t n, int rng_mode) {
    if (!key || !subs || n < 0 || bad_mode(rng_mode)) return G2048_EINVAL;
    for (int64_t i = 0; i < n; ++i) {
        u32 a0, a1, b0, b1;
        if (rng_mode) {
            tf2x32(key[0], key[1], 0u, 0u, a0, a1);
            tf2x32(key[0], key[1], 0u, 1u, b0, b1);
        } else {
            u32 p0, p1, q0, q1;
            tf2x32(key[0], key[1], 0u, 2u, p0, p1);
            tf2x32(key[0], key[1], 1u, 3u, q0, q1);
            a0 = p0; a1 = q0; b0 = p1; b1 = q1;
        }
        key[0] = a0; key[1] = a1;
        subs[2 * i] = b0; subs[2 * i + 1] = b1;
    }
    return 0;
}

int g2048_init(const uint32_t *keys, uint8_t *boards, uint8_t *masks, uint8_t *done, int64_t B, int rng_mode,
               void *stream) {
    if (!keys || !boards || !masks || !done || B <= 0 || B > kMaxEnvs || bad_mode(rng_mode)) return G2048_EINVAL;
    if (((uintptr_t)boards & 15) || ((uintptr_t)keys & 7)) return G2048_EINVAL;
    if (rng_mode) G2048_LAUNCH(k_init<1>, B, stream, keys, boards, masks, done, B);
    else G2048_LAUNCH(k_init<0>, B, stream, keys, boards, masks, done, B);
    return finish();
}

int g2048_step(uint8_t *boards, uint8_t *masks, uint8_t *done, const int32_t *actions, const uint32_t *keys,
               float *rewards, int64_t B, int rng_mode, void *stream) {
    if (!boards || !masks || !done || !actions || !keys || !rewards || B <= 0 || B > kMaxEnvs || bad_mode(rng_mode))
        return G2048_EINVAL;
    if (((uintptr_t)boards & 15) || ((uintptr_t)keys & 7)) return G2048_EINVAL;
    if (rng_mode) G2048_LAUNCH(k_step<1>, B, stream, boards, masks, done, actions, keys, rewards, B);
    else G2048_LAUNCH(k_step<0>, B, stream, boards, masks, done, actions, keys, rewards, B);
    return finish();
}

int g2048_observe(const uint8_t *boards, uint8_t *obs, int64_t B, void *stream) {
    if (!boards || !obs || B <= 0 || B > kMaxEnvs || ((uintptr_t)obs & 3)) return G2048_EINVAL;
    G2048_LAUNCH(k_observe, B * 124, stream, boards, obs, B);
    return finish();
}

int g2048_act_drul(const uint8_t *masks, int32_t *actions, int64_t B, void *stream) {
    if (!masks || !actions || B <= 0 || B > kMaxEnvs) return G2048_EINVAL;
    G2048_LAUNCH(k_act_drul, B, stream, masks, actions, B);
    return finish();
}

int g2048_act_random(const uint32_t *keys, const uint8_t *masks, int32_t *actions, float *log_probs, int64_t B,
                     int rng_mode, void *stream) {
    if (!keys || !masks || !actions || !log_probs || B <= 0 || B > kMaxEnvs || bad_mode(rng_mode)) return G2048_EINVAL;
    if ((uintptr_t)keys & 7) return G2048_EINVAL;
    if (rng_mode) G2048_LAUNCH(k_act_random<1>, B, stream, keys, masks, actions, log_probs, B);
    else G2048_LAUNCH(k_act_random<0>, B, stream, keys, masks, actions, log_probs, B);
    return finish();
}

int g2048_act_logits(const uint32_t *keys, const float *logits, const uint8_t *masks, int use_mask, int sample,
                     int32_t *actions, float *log_probs, int64_t B, int rng_mode, void *stream) {
    if (!keys || !logits || !masks || !actions || !log_probs || B <= 0 || B > kMaxEnvs || bad_mode(rng_mode))
        return G2048_EINVAL;
    if (((uintptr_t)keys & 7) || ((uintptr_t)logits & 15)) return G2048_EINVAL;
    if (rng_mode)
        G2048_LAUNCH(k_act_logits<1>, B, stream, keys, logits, masks, use_mask, sample, actions, log_probs, B);
    else
        G2048_LAUNCH(k_act_logits<0>, B, stream, keys, logits, masks, use_mask, sample, actions, log_probs, B);
    return finish();
}

int g2048_reset_fused(uint32_t sub0, uint32_t sub1, uint8_t *boards, uint8_t *masks, uint8_t *done,
                      int32_t *ep_len, int64_t B, int64_t B_total, int64_t env0, int rng_mode, void *stream) {
    if (!boards || !masks || !done || !ep_len || B <= 0 || env0 < 0 || B_total > kMaxEnvs || env0 + B > B_total ||
        bad_mode(rng_mode) || ((uintptr_t)boards & 15))
        return G2048_EINVAL;
    if (rng_mode)
        G2048_LAUNCH(k_reset_fused<1>, B, stream, sub0, sub1, boards, masks, done, ep_len, B, (u32)B_total, (u32)env0);
    else
        G2048_LAUNCH(k_reset_fused<0>, B, stream, sub0, sub1, boards, masks, done, ep_len, B, (u32)B_total, (u32)env0);
    return finish();
}

int g2048_rollout_fused(const uint32_t *step_subs, int n_steps, int64_t t0, uint8_t *boards, uint8_t *masks,
                        uint8_t *done, int32_t *ep_len, uint8_t *tr_boards, uint8_t *tr_meta, float *tr_rewards,
                        float *tr_logp, int64_t B, int64_t B_total, int64_t env0, int policy, int fill_frozen,
                        int rng_mode, uint32_t *live_count, void *stream) {
    if (!step_subs || n_steps <= 0 || n_steps > G2048_MAX_FUSED_STEPS || t0 < 0 || !boards || !masks || !done ||
        !ep_len || !tr_boards || !tr_meta || !tr_rewards || !live_count || B <= 0 || env0 < 0 ||
        B_total > kMaxEnvs || env0 + B > B_total || bad_mode(rng_mode) ||
        (policy != G2048_POLICY_DRUL && policy != G2048_POLICY_RANDOM))
        return G2048_EINVAL;
    if (((uintptr_t)boards & 15) || ((uintptr_t)tr_boards & 15)) return G2048_EINVAL;
    StepKeyTable tab;
    for (int s = 0; s < n_steps; ++s)
        for (int q = 0; q < 4; ++q) tab.k[s][q] = step_subs[4 * s + q];
    const u32 bt = (u32)B_total, e0 = (u32)env0;
#define G2048_RF(M, P)                                                                                            \
    G2048_LAUNCH((k_rollout_fused<M, P>), B, stream, tab, n_steps, t0, boards, masks, done, ep_len, tr_boards,   \
                 tr_meta, tr_rewards, tr_logp, B, bt, e0, fill_frozen, live_count)
    if (rng_mode) {
        if (policy == G2048_POLICY_RANDOM) G2048_RF(1, G2048_POLICY_RANDOM);
        else G2048_RF(1, G2048_POLICY_DRUL);
    } else {
        if (policy == G2048_POLICY_RANDOM) G2048_RF(0, G2048_POLICY_RANDOM);
        else G2048_RF(0, G2048_POLICY_DRUL);
    }
#undef G2048_RF
    return finish();
}

static int policy_step_impl(int autoreset, uint32_t act_sub0, uint32_t act_sub1, uint32_t step_sub0, uint32_t step_sub1,
                            const float *logits, const float *values, int use_mask, int sample, int64_t t, uint8_t *boards,
                            uint8_t *masks, uint8_t *done, int32_t *ep_len, uint8_t *tr_boards, uint8_t *tr_meta,
                            float *tr_rewards, float *tr_logp, float *tr_values, int64_t B, int64_t B_total, int64_t env0,
                            int fill_frozen, int rng_mode, uint32_t *live_count, void *stream) {
    if (!logits || !values || t < 0 || !boards || !masks || !ep_len || !tr_boards || !tr_meta || !tr_rewards || !tr_logp ||
        !tr_values || B <= 0 || env0 < 0 || B_total > kMaxEnvs || env0 + B > B_total || bad_mode(rng_mode))
        return G2048_EINVAL;
    if (!autoreset && !done) return G2048_EINVAL;  // (live_count may be NULL: no poll after this lock-step)
    if (((uintptr_t)boards & 15) || ((uintptr_t)tr_boards & 15) || ((uintptr_t)logits & 15)) return G2048_EINVAL;
    const u32 bt = (u32)B_total, e0 = (u32)env0;
#define G2048_PS(M, A)                                                                                                   \
    G2048_LAUNCH((k_policy_step<M, A>), B, stream, act_sub0, act_sub1, step_sub0, step_sub1, logits, values, use_mask,  \
                 sample, t, boards, masks, done, ep_len, tr_boards, tr_meta, tr_rewards, tr_logp, tr_values, B, bt, e0, \
                 fill_frozen, live_count)
    if (rng_mode) {
        if (autoreset) G2048_PS(1, 1);
        else G2048_PS(1, 0);
    } else {
        if (autoreset) G2048_PS(0, 1);
        else G2048_PS(0, 0);
    }
#undef G2048_PS
    return finish();
}

int g2048_policy_step(uint32_t act_sub0, uint32_t act_sub1, uint32_t step_sub0, uint32_t step_sub1,
                      const float *logits, const float *values, int use_mask, int sample, int64_t t,
                      uint8_t *boards, uint8_t *masks, uint8_t *done, int32_t *ep_len, uint8_t *tr_boards,
                      uint8_t *tr_meta, float *tr_rewards, float *tr_logp, float *tr_values, int64_t B,
                      int64_t B_total, int64_t env0, int fill_frozen, int rng_mode, uint32_t *live_count,
                      void *stream) {
    return policy_step_impl(0, act_sub0, act_sub1, step_sub0, step_sub1, logits, values, use_mask, sample, t, boards, masks,
                            done, ep_len, tr_boards, tr_meta, tr_rewards, tr_logp, tr_values, B, B_total, env0, fill_frozen,
                            rng_mode, live_count, stream);
}

int g2048_policy_step_autoreset(uint32_t act_sub0, uint32_t act_sub1, uint32_t step_sub0, uint32_t step_sub1,
                                const float *logits, const float *values, int use_mask, int sample, int64_t t,
                                uint8_t *boards, uint8_t *masks, int32_t *ep_len, uint8_t *tr_boards, uint8_t *tr_meta,
                                float *tr_rewards, float *tr_logp, float *tr_values, int64_t B, int64_t B_total,
                                int64_t env0, int rng_mode, void *stream) {
    return policy_step_impl(1, act_sub0, act_sub1, step_sub0, step_sub1, logits, values, use_mask, sample, t, boards, masks,
                            nullptr, ep_len, tr_boards, tr_meta, tr_rewards, tr_logp, tr_values, B, B_total, env0, 0,
                            rng_mode, nullptr, stream);
}

int g2048_reset_key(uint32_t step_sub0, uint32_t step_sub1, uint32_t *out /*host [2]*/) {
    if (!out) return G2048_EINVAL;
    tf2x32(step_sub0, step_sub1, 0u, 0xFFFFFFFFu, out[0], out[1]);
    return 0;
}

int g2048_gae_tb(const float *tr_rewards, const float *tr_values, const int32_t *ep_len, float *tr_adv,
                 float *tr_ret, int64_t T, int64_t B, double gamma, double lam, void *stream) {
    if (!tr_rewards || !tr_values || !ep_len || !tr_adv || !tr_ret || T <= 0 || B <= 0) return G2048_EINVAL;
    G2048_LAUNCH(k_gae_tb, B, stream, tr_rewards, tr_values, ep_len, tr_adv, tr_ret, T, B, (float)gamma,
                 (float)(gamma * lam));
    return finish();
}

int g2048_gae_tb_boot(const float *tr_rewards, const float *tr_values, const uint8_t *tr_meta, const float *last_values,
                      float *tr_adv, float *tr_ret, int64_t T, int64_t B, double gamma, double lam, void *stream) {
    if (!tr_rewards || !tr_values || !tr_meta || !last_values || !tr_adv || !tr_ret || T <= 0 || B <= 0) return G2048_EINVAL;
    G2048_LAUNCH(k_gae_tb_boot, B, stream, tr_rewards, tr_values, tr_meta, last_values, tr_adv, tr_ret, T, B, (float)gamma,
                 (float)(gamma * lam));
    return finish();
}

int g2048_gae_flat(const float *rewards, const float *values, const uint8_t *terms, float *adv, float *ret,
                   int64_t N, double gamma, double lam, void *stream) {
    if (!rewards || !values || !terms || !adv || !ret || N <= 0) return G2048_EINVAL;
    G2048_LAUNCH(k_gae_flat, N, stream, rewards, values, terms, adv, ret, N, (float)gamma, (float)(gamma * lam));
    return finish();
}

int g2048_compact(const uint8_t *tr_boards, const uint8_t *tr_meta, const float *tr_rewards, const float *tr_logp,
                  const float *tr_values, const float *tr_adv, const float *tr_ret, const int32_t *ep_len,
                  const int64_t *offsets, uint8_t *out_boards, uint8_t *out_actions, uint8_t *out_masks, float *out_rewards,
                  float *out_logp, float *out_values, float *out_adv, float *out_ret, uint8_t *out_terms, int64_t T,
                  int64_t B, void *stream) {
    if (!tr_boards || !tr_meta || !tr_rewards || !ep_len || !offsets || !out_boards || !out_actions || !out_masks ||
        !out_rewards || !out_terms || T <= 0 || B <= 0)
        return G2048_EINVAL;
    if ((out_logp && !tr_logp) || (out_values && !tr_values) || (out_adv && !tr_adv) || (out_ret && !tr_ret)) return G2048_EINVAL;
    if (((uintptr_t)tr_boards & 15) || ((uintptr_t)out_boards & 15)) return G2048_EINVAL;
    G2048_LAUNCH(k_compact, B * 64, stream, tr_boards, tr_meta, tr_rewards, tr_logp, tr_values, tr_adv, tr_ret, ep_len,
                 offsets, out_boards, out_actions, out_masks, out_rewards, out_logp, out_values, out_adv, out_ret, out_terms,
                 T, B);
    return finish();
}

}  // extern "C"
