/*
 * g2048.h -- C ABI of libg2048.so, the MI355X (gfx950) 2048 rollout engine.
 *
 * This is the drop-in boundary of the hot path.  The reference (michaelriedl/2048-ppo-agent) is pure
 * Python and has no FFI of its own: its hot path calls jitted JAX/Pgx functions.  Each entry point
 * below names the reference call it replaces (paths relative to the reference repo); the ctypes
 * binding a maintainer would add is shown in INTEGRATION.md and lives in
 * 2048-ppo-agent_amd/src/g2048/native.py.
 *
 * Conventions
 *   - All array pointers are DEVICE pointers owned by the caller (e.g. torch tensors); the library
 *     never allocates, frees, copies to the host or synchronises.  Kernels are enqueued on `stream`
 *     (a hipStream_t passed as void*; NULL = the null stream).  Host-side arguments are marked (host).
 *   - Return value: 0 on success, G2048_EINVAL (-1) for a bad argument, -(1000 + hipError_t) when the
 *     launch itself failed.  Nothing throws.  Re-entrant: no global state (kernels that need more than the default
 *     dynamic-LDS limit set that per-device function attribute on every call instead of latching it).
 *   - boards   u8[B][16]  log2(tile) per cell, row-major, 0 = empty, values <= 30; 16-byte aligned
 *   - masks    u8[B]      bit a = legal_action_mask[a]  (0 left, 1 up, 2 right, 3 down)
 *   - done     u8[B]      terminated flag (0/1)
 *   - keys     u32[B][2]  one JAX threefry key per env
 *   - rng_mode 0 = legacy threefry stream (jax_threefry_partitionable=False),
 *              1 = partitionable stream (default of the reference's pinned jax 0.5.3)
 *   - Trajectories are step-major SoA with row stride B:  x[t][e] at index t*B + e
 *       tr_boards u8[T][B][16]  board BEFORE step t (the observation the policy saw)
 *       tr_meta   u8[T][B]      action | mask_before << 2 | done_after << 6
 *       tr_rewards/tr_logp/tr_values f32[T][B]
 */
#ifndef G2048_H
#define G2048_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define G2048_ABI_VERSION 4
#define G2048_EINVAL (-1)
#define G2048_RNG_LEGACY 0
#define G2048_RNG_PARTITIONABLE 1
#define G2048_POLICY_DRUL 0
#define G2048_POLICY_RANDOM 1
/* most lock-steps one g2048_rollout_fused launch can advance (its key table rides in the kernarg) */
#define G2048_MAX_FUSED_STEPS 128

int g2048_abi_version(void);

/* ---- RNG ---------------------------------------------------------------------------------------- */

/* out[j] = jax.random.split(key, n)[j].  Replaces `subkey = jax.random.split(subkey, batch_size)`
 * (src/runs/batch_runner.py:106,119,127; src/runs/run_actions_batch.py:43,50,53). */
int g2048_split(uint32_t key0, uint32_t key1, uint32_t *out_keys, int64_t n, int rng_mode, void *stream);

/* (host, no device work) n times `key, sub = jax.random.split(key)`: key[2] is advanced in place,
 * subs[i] = i-th sub-key.  Replaces the host-side chain at src/runs/batch_runner.py:105,118,126. */
int g2048_chain_keys(uint32_t *key /*host [2]*/, uint32_t *subs /*host [n][2]*/, int64_t n, int rng_mode);

/* ---- env, explicit per-env keys (the shape of the reference's vmapped calls) ----------------------- */

/* jit(vmap(env.init))(keys)           src/runs/batch_runner.py:34,107 */
int g2048_init(const uint32_t *keys, uint8_t *boards, uint8_t *masks, uint8_t *done, int64_t B,
               int rng_mode, void *stream);

/* jit(vmap(env.step))(state, action, keys), state updated in place.   src/runs/batch_runner.py:35,128
 * Algorithmic traffic: 50 B per env (board 16 r + 16 w, action 4, key 8, reward 4, mask 1, done 1). */
int g2048_step(uint8_t *boards, uint8_t *masks, uint8_t *done, const int32_t *actions,
               const uint32_t *keys, float *rewards, int64_t B, int rng_mode, void *stream);

/* State.observation: obs u8(bool)[B][4][4][31], obs[e][r][c][k] = (board == k).  Only for callers that
 * want the reference's one-hot array (src/runs/batch_runner.py:121,138); the engine never needs it. */
int g2048_observe(const uint8_t *boards, uint8_t *obs, int64_t B, void *stream);

/* ---- act_fn plug-ins, batched ------------------------------------------------------------------- */

/* src/actions/act_drul.py:40-44 */
int g2048_act_drul(const uint8_t *masks, int32_t *actions, int64_t B, void *stream);
/* src/actions/act_randomly.py:40-54 */
int g2048_act_random(const uint32_t *keys, const uint8_t *masks, int32_t *actions, float *log_probs,
                     int64_t B, int rng_mode, void *stream);
/* tail of TorchActionFunction.__call__ (src/ppo/torch_action_wrapper.py:85-102) given the agent's raw
 * actor logits f32[B][4]; use_mask applies src/ppo/ppo_agent.py:117-121 first. */
int g2048_act_logits(const uint32_t *keys, const float *logits, const uint8_t *masks, int use_mask,
                     int sample, int32_t *actions, float *log_probs, int64_t B, int rng_mode, void *stream);

/* ---- fused rollout engine (keys derived in-kernel from the batch-wide sub-key) ---------------------
 * env i of this call is global env (env0 + i) of a batch of B_total envs, so a shard reproduces exactly
 * the slice of the single-device run (keys = split(sub, B_total)[env0 + i]). */

/* init + zero ep_len.  sub = the sub-key of batch_runner.py:105-106. */
int g2048_reset_fused(uint32_t sub0, uint32_t sub1, uint8_t *boards, uint8_t *masks, uint8_t *done,
                      int32_t *ep_len, int64_t B, int64_t B_total, int64_t env0, int rng_mode, void *stream);

/* n_steps lock-steps of BatchRunner's loop body (batch_runner.py:117-136) with a fused naive policy,
 * boards resident in registers across steps.  step_subs (host) = [n_steps][4]: act sub-key, step sub-key.
 * Writes steps t0..t0+n_steps-1 of the trajectory (tr_logp may be NULL; unused for DRUL).
 * fill_frozen != 0 also writes the frozen frames of already-finished envs (what the reference's [B,T]
 * arrays contain); 0 skips them and lets finished waves retire.  ep_len[e] counts steps through the first
 * termination.  *live_count (device u32, zeroed by the caller) += envs still running afterwards. */
int g2048_rollout_fused(const uint32_t *step_subs /*host*/, int n_steps, int64_t t0, uint8_t *boards,
                        uint8_t *masks, uint8_t *done, int32_t *ep_len, uint8_t *tr_boards,
                        uint8_t *tr_meta, float *tr_rewards, float *tr_logp, int64_t B, int64_t B_total,
                        int64_t env0, int policy, int fill_frozen, int rng_mode, uint32_t *live_count,
                        void *stream);

/* One lock-step with the policy's outputs already on the device: sample (or argmax) from logits,
 * log-prob, env step, trajectory write at step t -- the fusion of batch_runner.py:123-136 with
 * torch_action_wrapper.py:85-102.  logits f32[B][4], values f32[B] come from the agent forward.
 * live_count (device u32, zeroed by the caller; may be NULL when nobody polls after this step): += envs still running
 * afterwards, one atomic per workgroup that has any. */
int g2048_policy_step(uint32_t act_sub0, uint32_t act_sub1, uint32_t step_sub0, uint32_t step_sub1,
                      const float *logits, const float *values, int use_mask, int sample, int64_t t,
                      uint8_t *boards, uint8_t *masks, uint8_t *done, int32_t *ep_len, uint8_t *tr_boards,
                      uint8_t *tr_meta, float *tr_rewards, float *tr_logp, float *tr_values, int64_t B,
                      int64_t B_total, int64_t env0, int fill_frozen, int rng_mode, uint32_t *live_count,
                      void *stream);

/* Fixed-horizon throughput mode (no reference counterpart; it replaces the lock-step `while not all terminated` of
 * src/runs/batch_runner.py:117): g2048_policy_step for EVERY lane at every call, and a lane whose step
 * terminates starts its next episode at once: state = env.init(split(fold_in(step_sub, 0xFFFFFFFF), B_total)[env0 + i]).
 * Row t of the trajectory is written as by g2048_policy_step (done_after = 1 marks the episode boundary); boards/masks hold
 * the state for step t+1 (a fresh board after a terminal step), ep_len[i] the steps of the running episode.  The key chain
 * advances exactly as in the reference (two splits per lock-step), so until a lane's first termination its rows are those
 * of the lock-step mode under the same policy outputs. */
int g2048_policy_step_autoreset(uint32_t act_sub0, uint32_t act_sub1, uint32_t step_sub0, uint32_t step_sub1,
                                const float *logits, const float *values, int use_mask, int sample, int64_t t,
                                uint8_t *boards, uint8_t *masks, int32_t *ep_len, uint8_t *tr_boards, uint8_t *tr_meta,
                                float *tr_rewards, float *tr_logp, float *tr_values, int64_t B, int64_t B_total,
                                int64_t env0, int rng_mode, void *stream);
/* (host, no device work) out[2] = jax.random.fold_in(step_sub, 0xFFFFFFFF): the per-step reset sub-key of that mode. */
int g2048_reset_key(uint32_t step_sub0, uint32_t step_sub1, uint32_t *out /*host [2]*/);

/* ---- rollout buffer + GAE ----------------------------------------------------------------------- */

/* GAE over the step-major trajectory: env e uses steps 0..ep_len[e]-1 (its last kept step is terminal).
 * Same float32 operation order as PPODataset._compute_gae_returns (src/ppo/data_loader.py:103-130),
 * so results are bit-identical to the reference's scan over the compacted buffer. 17 B per env-step. */
int g2048_gae_tb(const float *tr_rewards, const float *tr_values, const int32_t *ep_len, float *tr_adv,
                 float *tr_ret, int64_t T, int64_t B, double gamma, double lam, void *stream);

/* GAE over a fixed-horizon trajectory: all T steps of every lane, episode boundaries = the done_after bit of tr_meta (the
 * reset at `terminations[step]` of src/ppo/data_loader.py:103-130), bootstrapped from last_values[e] = V(state after step
 * T-1) where the horizon cut an episode.  Same float32 operation order as g2048_gae_tb. */
int g2048_gae_tb_boot(const float *tr_rewards, const float *tr_values, const uint8_t *tr_meta, const float *last_values,
                      float *tr_adv, float *tr_ret, int64_t T, int64_t B, double gamma, double lam, void *stream);

/* GAE over a flat buffer with termination flags (the reference's layout, data_loader.py:103-130). */
int g2048_gae_flat(const float *rewards, const float *values, const uint8_t *terms, float *adv, float *ret,
                   int64_t N, double gamma, double lam, void *stream);

/* RolloutBuffer.store_batch (src/ppo/rollout_buffer.py:164-187): keep steps 0..ep_len[e]-1 of every env,
 * env-major, at offsets[e] (exclusive prefix sum of ep_len, int64).  The optional f32 columns (log-probs, values, and
 * the advantages / returns of g2048_gae_tb computed on the coalesced [T][B] layout) may be NULL together with their
 * outputs.  out_terms[n] = 1 on each env's last kept step. */
int g2048_compact(const uint8_t *tr_boards, const uint8_t *tr_meta, const float *tr_rewards, const float *tr_logp,
                  const float *tr_values, const float *tr_adv, const float *tr_ret, const int32_t *ep_len,
                  const int64_t *offsets, uint8_t *out_boards, uint8_t *out_actions, uint8_t *out_masks, float *out_rewards,
                  float *out_logp, float *out_values, float *out_adv, float *out_ret, uint8_t *out_terms, int64_t T,
                  int64_t B, void *stream);

/* ---- policy network (inference) ------------------------------------------------------------------- */

/* Fused bf16 forward of the reference's Transformer encoder at its default shape (d_model 256, 8 heads, ff 1024,
 * 17 tokens, "cls" reduction): boards u8[B][16] -> features f32[B][256], i.e. PPOAgent's
 * `self.transformer(self.input_embedding(obs), reduction="cls")` (src/ppo/ppo_agent.py:103-106,
 * src/ppo/transformer_encoder.py:150-190) in eval mode under bf16 autocast numerics.
 *   embed_table f32[16][31][256] = input_embedding.weight^T[e] + positional code of cell c   (16-byte aligned)
 *   cls_token   f32[256]                                                                     (16-byte aligned)
 * The per-layer parameters are passed PACKED (the caller packs once per policy update; src/ppo/fused_policy.py):
 *   folding, in f32:  LayerNorm's affine goes into the Linear behind it (W' = W diag(gamma), b' = b + W beta, for
 *     norm1 -> in_proj and norm2 -> linear1); the key bias is dropped (it shifts every score of a query by the same
 *     amount, softmax does not see it); the value bias moves into out_proj's: bo' = bo + Wo bv' (softmax rows sum to 1);
 *   column order: the kernel feeds accumulator tiles straight back in as MFMA operands, so the input columns of every
 *     group of 16 of all four matrices are stored in the order [0 1 2 3 8 9 10 11 4 5 6 7 12 13 14 15];
 *   weights_bf16, per layer: in_proj_weight'[768][256] | out_proj.weight[256][256] | linear1.weight'[1024][256] |
 *                            linear2.weight[256][1024]                                     (bf16, 16-byte aligned)
 *   params_f32,  per layer: ones[256] | zeros[256] | bq'[256], zeros[512] | bo'[256] | ones[256] | zeros[256] |
 *                            b1'[1024] | linear2.bias[256]   (the slots of the folded LayerNorm affines are unused)
 *
 * workspace: NULL = one kernel carries every token through every layer.  Otherwise
 * g2048_policy_encoder_workspace_bytes(B) bytes (16-byte aligned): the last layer is split - a first kernel parks that
 * layer's K/V and the CLS residual row there, a second one batches the CLS tokens of 128 boards per workgroup through the
 * rest of it (only the CLS row of the last layer is read by the "cls" reduction).  Same result up to summation order. */
int64_t g2048_policy_encoder_workspace_bytes(int64_t B);
int g2048_policy_encoder(const uint8_t *boards, const float *embed_table, const float *cls_token,
                         const void *weights_bf16, const float *params_f32, int n_layers, float *features,
                         int64_t B, void *workspace, void *stream);

/* ---- policy network (update): attention for 17-token sequences ------------------------------------- */

/* softmax(q k^T * scale) v with attention dropout, head_dim 32, Sk = 17 keys, Sq = 17 queries (or 1: the CLS row
 * of the last layer); replaces F.scaled_dot_product_attention inside the encoder layers of the update
 * (reference: nn.TransformerEncoderLayer built at src/ppo/transformer_encoder.py:138-148).
 * q/k/v (and dq/dk/dv) are bf16 with element strides (batch, token) and heads contiguous inside a token, so they may
 * point into the packed in_proj output [B][S][3][H][32] and its gradient; o/dout bf16 [B][Sq][H][32]; lse f32
 * [B][H][Sq].  The dropout mask is a function of (seed, *seed_state, element index): pass the same pair to the
 * backward.  seed_state: NULL, or a device-resident 64-bit word read by the kernel at run time, so that a launch
 * captured in a hipGraph draws a new mask on every replay once the owner advances the word between replays. */
int g2048_attn_fwd(const void *q, const void *k, const void *v, void *o, float *lse, int64_t B, int H, int Sq,
                   int64_t q_sb, int64_t q_ss, int64_t k_sb, int64_t k_ss, int64_t v_sb, int64_t v_ss, float scale,
                   float p_drop, uint64_t seed, const uint64_t *seed_state, void *stream);
int g2048_attn_bwd(const void *q, const void *k, const void *v, const void *dout, const float *lse, void *dq, void *dk,
                   void *dv, int64_t B, int H, int Sq, int64_t q_sb, int64_t q_ss, int64_t k_sb, int64_t k_ss,
                   int64_t v_sb, int64_t v_ss, float scale, float p_drop, uint64_t seed, const uint64_t *seed_state,
                   void *stream);

/* ---- policy network (update): residual add + dropout + LayerNorm, d_model 256 ------------------------ */

/* x_new = x + dropout(a);  h = bf16(LayerNorm(x_new) * gamma + beta): the tail of one pre-norm sub-layer fused with
 * the head of the next (reference: nn.TransformerEncoderLayer(norm_first=True), src/ppo/transformer_encoder.py:138-148;
 * `x = x + dropout1(sa(norm1(x)))`, `x = x + dropout2(ff(norm2(x)))`).
 * x f32 rows of 256 with element stride x_row_stride (a strided view of the residual stream is fine); a bf16 [T][256]
 * or NULL (then x_new is not written and h = LayerNorm(x)); x_new f32 [T][256]; h bf16 [T][256]; mean, rstd f32 [T]
 * (saved for the backward).  seed, seed_state: as for g2048_attn_fwd; pass the same pair to the backward.
 * gamma NULL: no LayerNorm, h = bf16(x + dropout(a)) (the output of the LAST sub-layer, which the heads read in bf16);
 * beta, mean, rstd and x_new may then be NULL too. */
int g2048_add_ln_fwd(const float *x, int64_t x_row_stride, const void *a, const float *gamma, const float *beta,
                     float *x_new, void *h, float *mean, float *rstd, int64_t T, float eps, float p_drop,
                     uint64_t seed, const uint64_t *seed_state, void *stream);
/* x_norm = the tensor that was normalised (x_new, or x when a was NULL); g_x f32 or NULL = gradient arriving on x_new from
 * the residual stream: [T][256] with g_x_period 1, or only for every g_x_period-th token row ([T / g_x_period][256], e.g.
 * period 17 = the CLS rows of [B][17][256], all the last encoder layer hands back); g_h bf16 [T][256].  dx f32 [T][256] = g_x + dLayerNorm (gradient for x);
 * da bf16 [T][256] or NULL = dropout-masked dx (gradient for a); dparams f32 [3][256] = dgamma, dbeta and the column
 * sums of da (= the bias gradient of the Linear that produced a; zeros when da is NULL), summed in a fixed order;
 * workspace: g2048_add_ln_bwd_workspace_floats(T) floats of scratch.  dparams NULL: first stage only, the workspace then
 * holds f32 [workspace floats / 768][3 * 256] partial sums (dgamma | dbeta | bias gradient) for g2048_reduce_jobs.
 * gamma NULL (the forward ran without LayerNorm): dx = g_x + g_h, dgamma = dbeta = 0; x_norm, mean, rstd are not read. */
int64_t g2048_add_ln_bwd_workspace_floats(int64_t T);
int g2048_add_ln_bwd(const float *x_norm, int64_t x_row_stride, const float *g_x, const void *g_h, const float *mean,
                     const float *rstd, const float *gamma, float *dx, void *da, float *dparams, float *workspace,
                     int64_t T, float p_drop, uint64_t seed, const uint64_t *seed_state, int g_x_period, void *stream);

/* ---- policy network (update): Linear (256 outputs) fused with the add + LayerNorm kernel behind it -------------- */

/* x_new = x + dropout(u . W^T + bias);  h = bf16(LayerNorm(x_new) * gamma + beta) in ONE launch: g2048_linear_bf16 (N = 256)
 * followed by g2048_add_ln_fwd, without the bf16 [T][256] tensor between them (reference: `x = x + dropout1(self_attn(...))` /
 * `x = x + dropout2(linear2(...))` + the next sub-layer's norm, nn.TransformerEncoderLayer(norm_first=True) built at
 * src/ppo/transformer_encoder.py:138-148).  u bf16 [T][K], leading dimension ldu (elements, multiple of 8); w_packed: the Linear's
 * weight [256][K] as bf16 in the FRAGMENT-PACKED layout above (g2048_opt_step maintains such copies); K = 256, 512, 768 or 1024; bias f32 [256]
 * or NULL; the rest as g2048_add_ln_fwd (same dropout hash on the same element index: the two paths draw the same mask for the same
 * seed).  The Linear's output is rounded to bf16 before dropout and the add, as the unfused pair does. */
int g2048_linear_add_ln_fwd(const void *u, int64_t ldu, const void *w_packed, const float *bias, int K, const float *x,
                            int64_t x_row_stride, const float *gamma, const float *beta, float *x_new, void *h, float *mean,
                            float *rstd, int64_t T, float eps, float p_drop, uint64_t seed, const uint64_t *seed_state, void *stream);
/* g_h = bf16(dy . Wt^T);  then g2048_add_ln_bwd on it: the input-gradient GEMM of the Linear that CONSUMED h (linear1: K = 1024,
 * in_proj: K = 768) fused with the backward of the LayerNorm that produced h (reference: autograd of the same modules).
 * dy bf16 [T][K]; wt_packed: the TRANSPOSE of that Linear's weight, [256][K], fragment-packed - or K consecutive columns of a wider
 * packed matrix [256][K'] (point at the first k-step and pass wt_tile_stride = (K' / 16) * 512, the distance in elements between its
 * 32-row tiles; 0 = dense): the K/V rows of a packed in_proj^T.  g_h_extra (bf16 [T / extra_period][256] or NULL) is added to g_h on
 * the rows tok % extra_period == 0 (rounded to bf16 again, as an addmm into the bf16 gradient would): the CLS rows' share of the
 * last layer's query projection.  da may be NULL (the LayerNorm had no branch); partial: f32
 * [g2048_linear_add_ln_bwd_partial_rows(T)][3][256] first-stage sums (dgamma | dbeta | column sums of da) for g2048_reduce_jobs. */
int64_t g2048_linear_add_ln_bwd_partial_rows(int64_t T);
int g2048_linear_add_ln_bwd(const void *dy, int64_t lddy, const void *wt_packed, int64_t wt_tile_stride, int K, const float *x_norm,
                            int64_t x_row_stride, const float *g_x, int g_x_period, const void *g_h_extra, int extra_period,
                            const float *mean, const float *rstd, const float *gamma, float *dx, void *da, float *partial, int64_t T,
                            float p_drop, uint64_t seed, const uint64_t *seed_state, void *stream);

/* ---- MLP policy (BASELINE configs[1]): the update as a handful of launches ---------------------------------------------- */

/* trunk_in on packed boards: y[m] = relu(bias + sum over the 16 cells c of wt[31 c + boards[m][c]]) (bf16 [M][512]) - a one-hot input
 * times W^T is a sum of weight columns (reference: the MLP policy has no counterpart in the reference; its input convention is the
 * one-hot observation of src/runs/batch_runner.py:130-136).  wt: bf16 [496][512], the TRANSPOSE of the Linear's weight [512][496];
 * bias f32 [512]; onehot (bf16 [M][512], columns 496.. zero; may be NULL): the one-hot matrix itself, which the weight-gradient launch
 * multiplies with. */
int g2048_mlp_embed_fwd(const uint8_t *boards, const void *wt, const float *bias, void *y, void *onehot, int64_t M, void *stream);

/* A table of small GEMMs in ONE launch: for every job y[M][N] (bf16) = epi(sum over its K-segments s of x[s][M][k[s]] . w[s][N][k[s]]^T),
 * bf16 operands, f32 accumulation; epi: + bias (f32 [N], or NULL), ReLU when relu != 0, multiplied by (act > 0) when act (bf16 [M][N],
 * leading dimension ldact) is given - the ReLU backward on a saved activation.  Two K-segments: two Linears back-propagating into the
 * same input.  Replaces nn.Linear + ReLU and their autograd for 2048-row activations, where a hipGraph node costs more than its
 * arithmetic.  N, k[s] multiples of 64 (k[1] may be 0); leading dimensions multiples of 8 (ldact: 4); bases 16-byte aligned (act: 8).
 * jobs: host array, read during the call. */
#define G2048_GEMM_MAX_JOBS 8
typedef struct {
    const void *x[2]; int64_t ldx[2];
    const void *w[2]; int64_t ldw[2];
    int32_t k[2];
    const float *bias;
    const void *act; int64_t ldact;
    void *y; int64_t ldy;
    int32_t N, relu;
} g2048_gemm_job;
int g2048_gemm_jobs(const g2048_gemm_job *jobs, int n_jobs, int64_t M, void *stream);

/* Both heads' output layers: logits[m][0..3] = h2[m][:512] . w3[0..3]^T, values[m] = h2[m][512:] . w3[4]  (h2 bf16 [M][1024] = the
 * actor's | the critic's last hidden layer; w3 bf16 [5][512] = actor.4.weight, then critic.4.weight: src/ppo/ppo_agent.py:72-87). */
int g2048_mlp_out_fwd(const void *h2, const void *w3, float *logits, float *values, int64_t M, void *stream);
/* Their backward: dh2 (bf16 [M][1024]) = [w3[0..3]^T dlogits | w3[4] dvalues] where h2 > 0; partial: f32
 * [g2048_mlp_out_bwd_partial_rows(M)][5][512] first-stage sums of the output layers' weight gradients (for g2048_reduce_jobs). */
int64_t g2048_mlp_out_bwd_partial_rows(int64_t M);
int g2048_mlp_out_bwd(const float *dlogits, const float *dvalues, const void *h2, const void *w3, void *dh2, float *partial, int64_t M,
                      void *stream);

/* ---- policy network (update): bias gradients ------------------------------------------------------------ */

/* out[c] = sum_r x[r][c] for x bf16 (is_bf16 != 0) or f32 [T][N] with element stride row_stride between rows; f32
 * accumulation in a fixed order (bit-reproducible, safe to replay from a hipGraph).  N a multiple of 4 (matrices wider
 * than 1024 columns are summed in column tiles of 1024), row_stride >= N.
 * The bias gradient of every Linear in the update, and the CLS-token gradient (reference: nn.Linear inside
 * src/ppo/transformer_encoder.py:138-148 and src/ppo/ppo_agent.py:59-86; PyTorch computes it with at::sum).
 * workspace: g2048_colsum_workspace_floats(T, N) floats of scratch.  out NULL (N <= 1024 only): first stage only, the
 * workspace then holds f32 [g2048_colsum_partial_rows(T, N)][N] partial sums for g2048_reduce_jobs. */
#define G2048_COLSUM_MAX_GROUPS 512
int64_t g2048_colsum_workspace_floats(int64_t T, int N);
int64_t g2048_colsum_partial_rows(int64_t T, int N);
int g2048_colsum(const void *x, int is_bf16, int64_t row_stride, int64_t T, int N, float *workspace, float *out,
                 void *stream);

/* ---- PPO update: loss ------------------------------------------------------------------------------------- */

/* Clipped-surrogate PPO loss of one minibatch, forward + gradient in one launch (reference:
 * PPOTrainer._compute_ppo_loss src/ppo/ppo_trainer.py:251-314 over PPOAgent.evaluate_actions src/ppo/ppo_agent.py:159-191).
 * logits [M][4] and values [M]: f32, or bf16 when the *_bf16 flag is set; actions u8 [M] (0..3); mask_bits u8 [M]
 * (bit a = action a legal) or NULL for no masking; old_logp, adv, ret f32 [M].
 * Out: new_logp f32 [M]; sums f32 [5] = mean policy loss, mean value loss, mean entropy loss (-H), mean total loss,
 * mean(old_logp - new_logp); dlogits [M][4] and dvalues [M] = d(mean total loss)/d(input) in the input's dtype.
 * total = policy + c_value * value + c_entropy * entropy_loss.  One workgroup, fixed summation order.
 * grad_scale (optional device f32 scalar, e.g. GradScaler's scale): dlogits and dvalues come out multiplied by it, i.e. as
 * the gradients of grad_scale * total (scaler.scale(loss).backward() of the reference, src/ppo/ppo_trainer.py:411-413).
 * running (optional device f64 [5]): running[k] += (double)sums[k] - the per-update accumulation of the logged means
 * (src/ppo/ppo_trainer.py:424-437 of the reference sums Python floats) without a launch of its own per minibatch. */
#define G2048_PPO_LOSS_MAX_BATCH 1048576
int g2048_ppo_loss(const void *logits, int logits_bf16, const void *values, int values_bf16, const uint8_t *actions,
                   const uint8_t *mask_bits, const float *old_logp, const float *adv, const float *ret, int64_t M,
                   float clip_eps, float c_value, float c_entropy, float *new_logp, float *sums, void *dlogits,
                   void *dvalues, const float *grad_scale, double *running, void *stream);

/* ---- policy network (update): feed-forward activation ---------------------------------------------------- */

/* y = dropout(relu(x)), x and y bf16 [T][F] (F a multiple of 8, F <= 2048): `dropout(activation(linear1(x)))` of the
 * encoder layer's feed-forward block (reference: nn.TransformerEncoderLayer, src/ppo/transformer_encoder.py:138-148).
 * seed, seed_state: as for g2048_attn_fwd. */
int g2048_relu_dropout_fwd(const void *x, void *y, int64_t T, int F, float p_drop, uint64_t seed,
                           const uint64_t *seed_state, void *stream);
/* dx = dy / (1 - p_drop) where y != 0, else 0 (y is non-zero exactly where the unit was active and kept: neither x nor
 * a mask is needed); dbias f32 [F] = column sums of dx in a fixed order = the bias gradient of the Linear in front.
 * dx may alias dy.  workspace: g2048_relu_dropout_bwd_workspace_floats(T, F) floats.  dbias NULL: first stage only, the
 * workspace then holds f32 [workspace floats / F][F] partial sums for g2048_reduce_jobs. */
int64_t g2048_relu_dropout_bwd_workspace_floats(int64_t T, int F);
int g2048_relu_dropout_bwd(const void *dy, const void *y, void *dx, float *dbias, float *workspace, int64_t T, int F,
                           float p_drop, void *stream);

/* ---- policy network (update): weight gradient of a Linear over the whole minibatch ------------------------------ */

/* parts[s][n][k] = sum over the tokens t of slice s (T / slices consecutive rows) of dy[t][n] * x[t][k]: the first stage of
 * dW = dY^T X for an nn.Linear applied to T token rows (reference: autograd of the Linears of nn.TransformerEncoderLayer,
 * src/ppo/transformer_encoder.py:138-148, inside PPOTrainer.update_policy, src/ppo/ppo_trainer.py:409-437); the sum over s is
 * left to g2048_reduce_jobs.  dy bf16 [T][N] and x bf16 [T][K] with leading dimensions lddy / ldx (elements, multiples of 8),
 * parts bf16 [slices][N][K] contiguous; N and K multiples of 128, T a multiple of 64 * slices, slices 1..7 or a multiple of
 * 8; base pointers 16-byte aligned.  f32 accumulation over a slice, one rounding to bf16 per partial.  block_rows: rows of the
 * gradient per workgroup (x 128 columns), 128 or 256 (N a multiple of 256), 0 = the kernel's choice; slices x blocks workgroups.
 * colsum (optional): f32 [slices][N], colsum[s][n] = sum over the slice's tokens of dy[t][n] (f32 sum of the bf16 values) - the
 * first stage of the Linear's bias gradient `dy.sum(0)`, read off the operand tiles the product stages anyway. */
int g2048_dweight_bf16(const void *dy, int64_t lddy, const void *x, int64_t ldx, void *parts, float *colsum, int64_t T, int N,
                       int K, int slices, int block_rows, void *stream);

/* The same for several Linears in ONE launch (the weight gradients of a whole backward pass, deferred to its end by the caller):
 * job j is g2048_dweight_bf16(dy, lddy, x, ldx, parts, colsum, T, N, K, slices, 128) with slices a multiple of 8.  jobs: host array,
 * read during the call.  parts_f32 != 0: the job's partials are stored as f32 [slices][N][K] instead of bf16 (twice the bytes into
 * g2048_reduce_jobs; the switch behind profiles/round4_dweight_slices_seeds.txt). */
#define G2048_DWG_MAX_JOBS 16
typedef struct {
    const void *dy; const void *x; void *parts; float *colsum;
    int64_t lddy, ldx, T;
    int32_t N, K, slices, parts_f32;
} g2048_dwg_job;
int g2048_dweight_jobs(const g2048_dwg_job *jobs, int n_jobs, void *stream);

/* ---- policy network (update): Linear for tall-skinny activations ------------------------------------------ */

/* y[T][N] = x[T][K] . weight[N][K]^T (+ bias[N]) - nn.Linear in bf16 with f32 accumulation (reference: every nn.Linear
 * of nn.TransformerEncoderLayer, src/ppo/transformer_encoder.py:138-148, and the same product with weight^T for the
 * input gradient).  x, weight, y bf16 with leading dimensions ldx, ldw, ldy (elements, multiples of 8); bias f32 or
 * NULL; K and N multiples of 128; all base pointers 16-byte aligned. */
int g2048_linear_bf16(const void *x, int64_t ldx, const void *weight, int64_t ldw, const float *bias, void *y, int64_t ldy,
                      int64_t T, int K, int N, void *stream);

/* y = dropout(relu(x . weight^T + bias)) in one launch: linear1 + activation + dropout of the encoder layer's feed-forward
 * block (reference: nn.TransformerEncoderLayer._ff_block, built at src/ppo/transformer_encoder.py:138-148).  Operands as
 * for g2048_linear_bf16 with K <= 256 and bias required; the pre-activation is never rounded to bf16.  seed / seed_state
 * as for g2048_attn_fwd.  The output is non-zero exactly where the unit was active and kept; mask_bits (optional,
 * g2048_ffn_mask_bytes(T, N) bytes, 8-byte aligned) receives one bit per output saying so, in an opaque layout that only
 * g2048_linear_mask_bwd_bf16 with the same T and N reads. */
int64_t g2048_ffn_mask_bytes(int64_t T, int N);
int g2048_linear_relu_dropout_bf16(const void *x, int64_t ldx, const void *weight, int64_t ldw, const float *bias, void *y,
                                   int64_t ldy, int64_t T, int K, int N, float p_drop, uint64_t seed,
                                   const uint64_t *seed_state, void *mask_bits, void *stream);
/* The backward of `y = dropout(relu(.)); out = y . W2^T` with respect to the pre-activation, in one GEMM:
 * dz[T][N] = (dy[T][K] . weight_t[N][K]^T) / (1 - p_drop) where y != 0 (mask_bits of the forward call), else 0
 * (weight_t = W2^T, i.e. linear2's weight [K][N] transposed to [N][K]; dy = the gradient of linear2's output), and
 * dbias f32 [N] = column sums of the bf16 dz in a fixed order = linear1's bias gradient.  Replaces g2048_linear_bf16 +
 * g2048_relu_dropout_bwd (which re-reads the [T][N] activation; the bit mask is 1/16 of it).  K <= 256;
 * workspace: g2048_linear_mask_bwd_workspace_floats(T, N) floats; dbias NULL: first stage only, the workspace then holds
 * f32 [g2048_linear_mask_bwd_partial_rows(T, N)][N] partial sums for g2048_reduce_jobs. */
int64_t g2048_linear_mask_bwd_workspace_floats(int64_t T, int N);
int64_t g2048_linear_mask_bwd_partial_rows(int64_t T, int N);
int g2048_linear_mask_bwd_bf16(const void *dy, int64_t lddy, const void *weight_t, int64_t ldw, const void *mask_bits, void *dz,
                               int64_t lddz, float *dbias, float *workspace, int64_t T, int K, int N, float p_drop,
                               void *stream);

/* ---- policy network (update): token embedding of packed boards ---------------------------------------------- */

/* x0[m][0] = cls, x0[m][1 + c] = dropout(wt[boards[m][c]] + pe[c]): the bias-free input Linear over the one-hot cell (w_ld
 * = 0: wt = input_embedding.weight^T, f32 [31][256], 16-byte aligned; w_ld >= 31: wt = the nn.Linear weight itself, f32
 * [256][w_ld], read in place), the 2-D positional code (pe f32 [16][256]) and the CLS concat
 * (reference: src/ppo/ppo_agent.py:59-66,103-106; src/ppo/transformer_encoder.py:150-190).  boards u8 [M][16] with
 * cells <= 30, x0 f32 [M][17][256].  The positional encoding's dropout (p_drop; seed, seed_state as for g2048_attn_fwd)
 * applies to the 16 board tokens, not to the CLS row. */
int g2048_embed_fwd(const uint8_t *boards, const float *wt, int w_ld, const float *pe, const float *cls, float *x0, int64_t M,
                    float p_drop, uint64_t seed, const uint64_t *seed_state, void *stream);
/* dwt_dcls f32 [32][256]: rows 0..30 = gradient of wt (sum of dx0 rows by cell value), row 31 = gradient of cls; fixed
 * summation order.  dx0 f32 [M][17][256]; workspace: g2048_embed_bwd_workspace_floats(M) floats.  dwt_dcls NULL: first
 * stage only, the workspace then holds f32 [workspace floats / 8192][32 * 256] partial sums for g2048_reduce_jobs (whose
 * transpose_rows = 31 stores the first 31 * 256 columns as the [256][31] gradient of the nn.Linear weight). */
/* The same, and on the row it still holds h[m][t] = bf16(LayerNorm(x0[m][t]) * gamma + beta) with mean / rstd f32 [M * 17]: the first
 * LayerNorm of the encoder (layers[0].norm1, src/ppo/transformer_encoder.py:138-148 norm_first) in g2048_add_ln_fwd's arithmetic, without
 * reading x0 back.  gamma, beta f32 [256]; h bf16 [M][17][256]. */
int g2048_embed_ln_fwd(const uint8_t *boards, const float *wt, int w_ld, const float *pe, const float *cls, float *x0, int64_t M,
                       float p_drop, uint64_t seed, const uint64_t *seed_state, const float *gamma, const float *beta, float eps,
                       void *h, float *mean, float *rstd, void *stream);
int64_t g2048_embed_bwd_workspace_floats(int64_t M);
int g2048_embed_bwd(const uint8_t *boards, const float *dx0, float *dwt_dcls, float *workspace, int64_t M, float p_drop,
                    uint64_t seed, const uint64_t *seed_state, void *stream);

/* ---- PPO update: minibatch assembly ------------------------------------------------------------------------ */

/* ---- gradient reductions of the update, second stage ----------------------------------------------------------- */

/* All second-stage reductions of one backward pass in one launch (per 64 jobs): for every job,
 * dst[c] = sum over p < parts of src[p * part_stride + c], c < n, in f32 and in a fixed order.  Jobs: the 16 split-K
 * slices of a weight gradient (bf16 [16][out * in]), the per-workgroup partial column sums the backward kernels leave in
 * their workspace when called with a NULL result pointer (g2048_add_ln_bwd, g2048_relu_dropout_bwd,
 * g2048_linear_mask_bwd_bf16, g2048_embed_bwd, g2048_colsum: f32 [rows][N], rows = workspace floats / N), a bf16 -> f32
 * conversion (parts 1).  Replaces what PyTorch runs as one at::sum / copy kernel per parameter
 * (reference: loss.backward() at src/ppo/ppo_trainer.py:410-414).  jobs: host array, read during the call; src 2- (bf16) or
 * 4-byte aligned device memory, vector loads when 8-/16-byte aligned with part_stride % 4 == 0.
 * transpose_rows = R > 0 (n a multiple of R): the n columns are a row-major [R][n / R] matrix and the sum is stored
 * transposed, dst[(c % (n / R)) * R + c / (n / R)] - the embedding gradient is accumulated per class ([31][256]) and belongs
 * to an nn.Linear weight ([256][31]); 0: dst[c]. */
#define G2048_REDUCE_MAX_JOBS 64
typedef struct {
    const void *src; float *dst; int64_t part_stride; int32_t n, parts, src_bf16, transpose_rows;
} g2048_reduce_job;
int g2048_reduce_jobs(const g2048_reduce_job *jobs, int n_jobs, void *stream);

/* ---- optimiser step (update) ---------------------------------------------------------------------------- */

/* One optimiser step for every parameter of the agent in two launches: GradScaler unscale + inf check, gradient-norm
 * clipping, AdamW, GradScaler update (reference: src/ppo/ppo_trainer.py:413-434 = scaler.unscale_(opt);
 * clip_grad_norm_(params, max_grad_norm); scaler.step(opt); scaler.update(), with opt = torch.optim.AdamW built by
 * src/optim/configure_optimizers.py:16-127).
 * grads / exp_avg / exp_avg_sq: flat f32 device buffers (16-byte aligned) sharing one element layout; chunks[n_chunks]
 * (device memory): piece `n` <= G2048_OPT_CHUNK elements of one parameter tensor starting at `param` (16-byte aligned),
 * whose gradient and moments start at element `offset` (a multiple of 4) of the flat buffers, hyper-parameters
 * groups[group].  groups: host array, read during the call (the LR schedule changes lr every step).
 * A chunk may name bf16 shadows of its tensor (what the bf16 update path multiplies with): they are rewritten together with
 * the parameter, so that no cast kernels run per step.
 * max_grad_norm <= 0: no clipping.  steps: device f32 [n_steps], the number of steps taken so far, one copy per parameter
 * tensor as torch.optim keeps them (all equal; every entry +1 per non-skipped call).
 * scale / growth_tracker: the GradScaler's device scalars, or NULL/NULL when no scaler is used (then gradients are taken
 * as they are and nothing is skipped); with a scaler, a step whose gradients hold an inf/nan changes neither parameters
 * nor moments nor steps, and the scale is multiplied by backoff; after growth_interval consecutive clean steps by growth.
 * workspace: g2048_opt_workspace_floats(n_chunks) floats.  info (optional, device f32[2]): total gradient norm before clipping, found_inf.
 * Arithmetic is torch's fused AdamW (bias corrections in f64 from the step count); the summation order of the norm is
 * fixed by the chunk table. */
#define G2048_OPT_CHUNK 2048
#define G2048_OPT_MAX_GROUPS 4
typedef struct {
    float *param; int64_t offset; int32_t n; int32_t group;
    void *shadow;      /* optional bf16 copy of the tensor (dense, same element order): the chunk's piece is refreshed in place */
    void *shadow_t;    /* optional bf16 copy of the TRANSPOSED 2-D tensor [cols][rows] */
    void *shadow_p;    /* optional bf16 copy of the 2-D tensor in the fragment-packed layout (see g2048_tail_saved); rows % 32 == 0, cols % 16 == 0 */
    void *shadow_tp;   /* optional fragment-packed bf16 copy of the TRANSPOSED tensor; cols % 32 == 0, rows % 16 == 0 */
    int32_t e0;        /* element index of the chunk's first element inside its tensor */
    int32_t rows, cols; /* shape of the 2-D tensor (only read when shadow_t is set) */
    int32_t reserved;
} g2048_opt_chunk;
typedef struct { double lr, beta1, beta2, eps, weight_decay; } g2048_opt_group; /* f64, as torch.optim holds them */
int64_t g2048_opt_workspace_floats(int n_chunks);
int g2048_opt_step(const g2048_opt_chunk *chunks, int n_chunks, const float *grads, float *exp_avg, float *exp_avg_sq,
                   const g2048_opt_group *groups, int n_groups, float max_grad_norm, float *steps, int n_steps, float *scale,
                   int32_t *growth_tracker, float growth, float backoff, int growth_interval, float *workspace,
                   float *info, void *stream);

/* ---- policy network (update): the 2048-row tail as two kernels -------------------------------------------------- */

/* Everything after the last encoder layer's attention works on ONE row per board (the "cls" reduction reads only the CLS row
 * of the last layer): out_proj + dropout + residual + LayerNorm(norm2), linear1 + ReLU + dropout, linear2 + dropout +
 * residual, then the actor head (256 -> 512 -> 512 -> 4) and the critic head (256 -> 512 -> 512 -> 1)
 * (reference: nn.TransformerEncoderLayer(norm_first=True), src/ppo/transformer_encoder.py:138-148 and :150-190 for the CLS
 * read-out; the heads of src/ppo/ppo_agent.py:62-92, evaluated by evaluate_actions :159-191).  d_model 256, feed-forward
 * 1024, head width 512.  g2048_cls_tail_fwd runs that chain for M rows in one launch, g2048_cls_tail_bwd its backward, and
 * g2048_dweight_t all weight / bias gradients from the transposed operands the two leave behind.
 * Weights: bf16 copies in nn.Linear's [out][in] layout (16-byte aligned), biases and LayerNorm parameters f32. */
typedef struct {
    const void *wo, *w1, *w2;           /* out_proj [256][256], linear1 [1024][256], linear2 [256][1024]: packed */
    const void *a1, *a2, *a3;           /* actor  [512][256], [512][512] packed; [4][512] row-major (no bias) */
    const void *c1, *c2, *c3;           /* critic [512][256], [512][512] packed; [1][512] row-major (no bias) */
    const float *bo, *b1, *b2, *ab1, *ab2, *cb1, *cb2;
    const float *ln_g, *ln_b;           /* norm2 */
} g2048_tail_weights;
/* the backward multiplies with the TRANSPOSED weights ([in][out], bf16, packed; g2048_opt_step keeps such shadows current) */
typedef struct {
    const void *woT, *w1T, *w2T;        /* [256][256], [256][1024], [1024][256] */
    const void *a1T, *a2T, *a3;         /* [256][512], [512][512]; a3 as it is, [4][512] */
    const void *c1T, *c2T, *c3;         /* [256][512], [512][512]; c3 as it is, [1][512] */
    const float *ln_g;
} g2048_tail_weights_t;
/* Fragment-packed layout ("packed" below) of a bf16 matrix X[rows][cols], rows % 32 == 0, cols % 16 == 0 -- the order in
 * which a wavefront consumes X as an MFMA operand, one contiguous KB per wave-instruction:
 *   offset(row, col) = ((((row / 32) * (cols / 16) + col / 16) * 2 + (col / 8) % 2) * 32 + row % 32) * 8 + col % 8.
 * The weights of g2048_tail_weights / g2048_tail_weights_t (except a3, c3: row-major) and every transposed activation buffer
 * below are stored that way; g2048_opt_step maintains packed shadows of parameters (g2048_opt_chunk.shadow_p / shadow_tp). */
/* What the forward leaves for the backward and for the weight gradients.  ld = number of columns of every transposed
 * buffer, a multiple of 32 and >= 32 * ceil(M / 32); columns M..ld-1 are written as zero.
 * masks: u16 [ceil(M/32)][G2048_TAIL_MASK_TILES][64], one bit per element of an MFMA accumulator tile (private layout). */
#define G2048_TAIL_MASK_TILES 96
typedef struct {
    float *x_mid, *mean, *rstd;         /* f32 [M][256] residual after the attention block; LayerNorm statistics [M] */
    void *masks;
    void *oT, *h2T, *uT, *featsT;       /* bf16 [256][ld], [256][ld], [1024][ld], [256][ld]: inputs of out_proj, linear1, linear2, heads */
    void *a1T, *a2T, *c1T, *c2T;        /* bf16 [512][ld] each: hidden activations of the heads */
    int64_t ld;
} g2048_tail_saved;
/* dY^T of every Linear (bf16, [out][ld]; dlT / dvT have 32 rows of which 4 / 1 are written: allocate them zero-filled) and
 * the LayerNorm gradient partials f32 [ceil(M/32)][2][256] (dgamma | dbeta per workgroup, for g2048_reduce_jobs). */
typedef struct {
    void *daoT, *dzT, *df2T;            /* out_proj [256][ld], linear1 [1024][ld], linear2 [256][ld] */
    void *da1T, *da2T, *dlT;            /* actor [512][ld], [512][ld], [32][ld] */
    void *dc1T, *dc2T, *dvT;            /* critic [512][ld], [512][ld], [32][ld] */
    float *ln_partial;
} g2048_tail_grads;
/* o: bf16 [M][256] attention output of the CLS rows; x_cls: f32 rows of 256 with element stride x_row_stride (the CLS rows
 * of the residual stream, read in place); logits f32 [M][4], values f32 [M].  Dropout as everywhere in the update: the mask
 * is a function of (seed, *seed_state, site, element); pass the same triple to the backward. */
int g2048_cls_tail_fwd(const void *o, const float *x_cls, int64_t x_row_stride, const g2048_tail_weights *W,
                       const g2048_tail_saved *S, float *logits, float *values, int64_t M, float eps, float p_drop, uint64_t seed,
                       const uint64_t *seed_state, void *stream);
/* dlogits f32 [M][4], dvalues f32 [M] -> d_o bf16 [M][256] (gradient of the attention output), dx_cls f32 [M][256]
 * (gradient of the residual CLS rows), G (see above). */
int g2048_cls_tail_bwd(const float *dlogits, const float *dvalues, const g2048_tail_weights_t *WT, const g2048_tail_saved *S,
                       const g2048_tail_grads *G, void *d_o, float *dx_cls, int64_t M, float p_drop, uint64_t seed,
                       const uint64_t *seed_state, void *stream);

/* Weight gradients from transposed operands, all jobs in one launch: for job j,
 * dw[s][n][k] = sum over the s-th of `slices` equal pieces of the row axis of dyT[n][m] * xT[k][m]   (f32 [slices][N][K]),
 * db[s][n]    = the same sum of dyT[n][m]                                                        (f32 [slices][N], or NULL);
 * dyT bf16 [N][ld], xT bf16 [K][ld], both packed; N a multiple of 32, K of 64, m (rows used, <= ld) a multiple of 16 * slices
 * (8 slices put each slice on one XCD).  The slices are
 * summed by g2048_reduce_jobs.  Replaces dY^T X (torch: `dy.t() @ x`) and the bias column sums in the backward of every
 * nn.Linear of the tail (reference: loss.backward() at src/ppo/ppo_trainer.py:410-414).  jobs: host array, read during the call. */
#define G2048_DW_MAX_JOBS 16
typedef struct { const void *dyT; const void *xT; float *dw; float *db; int32_t N, K; } g2048_dw_job;
int g2048_dweight_t(const g2048_dw_job *jobs, int n_jobs, int64_t ld, int64_t m, int slices, void *stream);

/* Row i of the minibatch = sample idx[i] (int64, clamped to [0, N)) of the device-resident rollout buffer: boards
 * u8 [N][16], actions u8 [N], masks u8 [N], logp / adv / ret f32 [N] -> the o_* arrays of M rows.  One launch for what
 * the reference's DataLoader collation does per field (PPODataset.__getitem__, src/ppo/data_loader.py). */
int g2048_gather_minibatch(const int64_t *idx, int64_t M, int64_t N, const uint8_t *boards, const uint8_t *actions,
                           const uint8_t *masks, const float *logp, const float *adv, const float *ret, uint8_t *o_boards,
                           uint8_t *o_actions, uint8_t *o_masks, float *o_logp, float *o_adv, float *o_ret, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* G2048_H */
