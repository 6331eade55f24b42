"""ctypes binding of libg2048.so (C ABI: include/g2048.h).

This is the only place the Python host side touches native code.  There is no CPU fallback: if the
library is missing, or a tensor is not on a HIP device, the call raises.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np
import torch

RNG_LEGACY = 0
RNG_PARTITIONABLE = 1
POLICY_DRUL = 0
POLICY_RANDOM = 1
MAX_FUSED_STEPS = 128

_PKG_ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), "..", ".."))
LIB_PATH = os.environ.get("G2048_LIB", os.path.join(_PKG_ROOT, "lib", "libg2048.so"))

_u32, _i32, _i64, _vp, _dbl = C.c_uint32, C.c_int, C.c_int64, C.c_void_p, C.c_double

# name -> argtypes; every entry point declared in include/g2048.h
SIGNATURES = {
    "g2048_abi_version": [],
    "g2048_split": [_u32, _u32, _vp, _i64, _i32, _vp],
    "g2048_chain_keys": [_vp, _vp, _i64, _i32],
    "g2048_init": [_vp, _vp, _vp, _vp, _i64, _i32, _vp],
    "g2048_step": [_vp, _vp, _vp, _vp, _vp, _vp, _i64, _i32, _vp],
    "g2048_observe": [_vp, _vp, _i64, _vp],
    "g2048_act_drul": [_vp, _vp, _i64, _vp],
    "g2048_act_random": [_vp, _vp, _vp, _vp, _i64, _i32, _vp],
    "g2048_act_logits": [_vp, _vp, _vp, _i32, _i32, _vp, _vp, _i64, _i32, _vp],
    "g2048_reset_fused": [_u32, _u32, _vp, _vp, _vp, _vp, _i64, _i64, _i64, _i32, _vp],
    "g2048_rollout_fused": [_vp, _i32, _i64, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _i64, _i64,
                            _i32, _i32, _i32, _vp, _vp],
    "g2048_policy_step": [_u32, _u32, _u32, _u32, _vp, _vp, _i32, _i32, _i64, _vp, _vp, _vp, _vp, _vp, _vp,
                          _vp, _vp, _vp, _i64, _i64, _i64, _i32, _i32, _vp, _vp],
    "g2048_policy_step_autoreset": [_u32, _u32, _u32, _u32, _vp, _vp, _i32, _i32, _i64, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp,
                                    _i64, _i64, _i64, _i32, _vp],
    "g2048_reset_key": [_u32, _u32, _vp],
    "g2048_gae_tb": [_vp, _vp, _vp, _vp, _vp, _i64, _i64, _dbl, _dbl, _vp],
    "g2048_gae_tb_boot": [_vp, _vp, _vp, _vp, _vp, _vp, _i64, _i64, _dbl, _dbl, _vp],
    "g2048_gae_flat": [_vp, _vp, _vp, _vp, _vp, _i64, _dbl, _dbl, _vp],
    "g2048_compact": [_vp] * 18 + [_i64, _i64, _vp],
    "g2048_policy_encoder_workspace_bytes": [_i64],
    "g2048_policy_encoder": [_vp, _vp, _vp, _vp, _vp, _i32, _vp, _i64, _vp, _vp],
    "g2048_attn_fwd": [_vp, _vp, _vp, _vp, _vp, _i64, _i32, _i32, _i64, _i64, _i64, _i64, _i64, _i64, C.c_float,
                       C.c_float, C.c_uint64, _vp, _vp],
    "g2048_attn_bwd": [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _i32, _i32, _i64, _i64, _i64, _i64, _i64, _i64,
                       C.c_float, C.c_float, C.c_uint64, _vp, _vp],
    "g2048_add_ln_fwd": [_vp, _i64, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i64, C.c_float, C.c_float, C.c_uint64, _vp, _vp],
    "g2048_ppo_loss": [_vp, _i32, _vp, _i32, _vp, _vp, _vp, _vp, _vp, _i64, C.c_float, C.c_float, C.c_float, _vp, _vp, _vp,
                       _vp, _vp, _vp, _vp],
    "g2048_linear_bf16": [_vp, _i64, _vp, _i64, _vp, _vp, _i64, _i64, _i32, _i32, _vp],
    "g2048_ffn_mask_bytes": [_i64, _i32],
    "g2048_linear_relu_dropout_bf16": [_vp, _i64, _vp, _i64, _vp, _vp, _i64, _i64, _i32, _i32, C.c_float, C.c_uint64, _vp, _vp, _vp],
    "g2048_linear_mask_bwd_workspace_floats": [_i64, _i32],
    "g2048_linear_mask_bwd_bf16": [_vp, _i64, _vp, _i64, _vp, _vp, _i64, _vp, _vp, _i64, _i32, _i32, C.c_float, _vp],
    "g2048_embed_fwd": [_vp, _vp, C.c_int, _vp, _vp, _vp, _i64, C.c_float, C.c_uint64, _vp, _vp],
    "g2048_embed_ln_fwd": [_vp, _vp, C.c_int, _vp, _vp, _vp, _i64, C.c_float, C.c_uint64, _vp, _vp, _vp, C.c_float, _vp, _vp, _vp, _vp],
    "g2048_embed_bwd_workspace_floats": [_i64],
    "g2048_embed_bwd": [_vp, _vp, _vp, _vp, _i64, C.c_float, C.c_uint64, _vp, _vp],
    "g2048_gather_minibatch": [_vp, _i64, _i64] + [_vp] * 13,
    "g2048_colsum_workspace_floats": [_i64, _i32],
    "g2048_colsum": [_vp, _i32, _i64, _i64, _i32, _vp, _vp, _vp],
    "g2048_colsum_partial_rows": [_i64, _i32],
    "g2048_linear_mask_bwd_partial_rows": [_i64, _i32],
    "g2048_reduce_jobs": [_vp, _i32, _vp],
    "g2048_linear_add_ln_fwd": [_vp, _i64, _vp, _vp, _i32, _vp, _i64, _vp, _vp, _vp, _vp, _vp, _vp, _i64, C.c_float, C.c_float,
                                C.c_uint64, _vp, _vp],
    "g2048_linear_add_ln_bwd_partial_rows": [_i64],
    "g2048_linear_add_ln_bwd": [_vp, _i64, _vp, _i64, _i32, _vp, _i64, _vp, _i32, _vp, _i32, _vp, _vp, _vp, _vp, _vp, _vp, _i64,
                                C.c_float, C.c_uint64, _vp, _vp],
    "g2048_mlp_embed_fwd": [_vp, _vp, _vp, _vp, _vp, _i64, _vp],
    "g2048_gemm_jobs": [_vp, _i32, _i64, _vp],
    "g2048_mlp_out_fwd": [_vp, _vp, _vp, _vp, _i64, _vp],
    "g2048_mlp_out_bwd_partial_rows": [_i64],
    "g2048_mlp_out_bwd": [_vp, _vp, _vp, _vp, _vp, _vp, _i64, _vp],
    "g2048_add_ln_bwd_workspace_floats": [_i64],
    "g2048_add_ln_bwd": [_vp, _i64, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i64, C.c_float, C.c_uint64, _vp,
                         _i32, _vp],
    "g2048_relu_dropout_fwd": [_vp, _vp, _i64, _i32, C.c_float, C.c_uint64, _vp, _vp],
    "g2048_relu_dropout_bwd_workspace_floats": [_i64, _i32],
    "g2048_relu_dropout_bwd": [_vp, _vp, _vp, _vp, _vp, _i64, _i32, C.c_float, _vp],
    "g2048_cls_tail_fwd": [_vp, _vp, _i64, _vp, _vp, _vp, _vp, _i64, C.c_float, C.c_float, C.c_uint64, _vp, _vp],
    "g2048_cls_tail_bwd": [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i64, C.c_float, C.c_uint64, _vp, _vp],
    "g2048_dweight_t": [_vp, _i32, _i64, _i64, _i32, _vp],
    "g2048_dweight_bf16": [_vp, _i64, _vp, _i64, _vp, _vp, _i64, _i32, _i32, _i32, _i32, _vp],
    "g2048_dweight_jobs": [_vp, _i32, _vp],
    "g2048_opt_workspace_floats": [_i32],
    "g2048_opt_step": [_vp, _i32, _vp, _vp, _vp, _vp, _i32, C.c_float, _vp, _i32, _vp, _vp, C.c_float, C.c_float, _i32, _vp, _vp, _vp],
}

_lib = None


class NativeError(RuntimeError):
    pass


def load() -> C.CDLL:
    """Load libg2048.so or raise; never falls back to anything else."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise NativeError(
                f"libg2048.so not found at {LIB_PATH}: build it with "
                f"`python 2048-ppo-agent_amd/build.py` (hipcc --offload-arch=gfx950). "
                "There is no CPU fallback for the 2048 rollout engine."
            )
        lib = C.CDLL(LIB_PATH)
        for name, argtypes in SIGNATURES.items():
            fn = getattr(lib, name)  # AttributeError if the symbol is missing
            fn.argtypes = argtypes
            fn.restype = C.c_int64 if name.endswith(("_workspace_floats", "_workspace_bytes", "_partial_rows", "_mask_bytes")) else C.c_int
        if lib.g2048_abi_version() != 4:
            raise NativeError("libg2048.so ABI version mismatch")
        _lib = lib
    return _lib


def _check(rc: int, name: str):
    if rc != 0:
        raise NativeError(f"{name} failed with code {rc}"
                          + (" (invalid argument)" if rc == -1 else f" (hipError {-rc - 1000})"))


def _dev(t: torch.Tensor | None, dtype, numel: int | None, name: str, optional=False):
    if t is None:
        if optional:
            return None
        raise NativeError(f"{name}: tensor required")
    if not t.is_cuda:
        raise NativeError(f"{name}: expected a HIP device tensor, got {t.device} (no CPU path exists)")
    if t.dtype != dtype or not t.is_contiguous():
        raise NativeError(f"{name}: expected contiguous {dtype}, got {t.dtype} contiguous={t.is_contiguous()}")
    if numel is not None and t.numel() < numel:
        raise NativeError(f"{name}: needs {numel} elements, has {t.numel()}")
    return t.data_ptr()


def _stream():
    return torch.cuda.current_stream().cuda_stream


u8, i32, i64, f32 = torch.uint8, torch.int32, torch.int64, torch.float32
# uint32 keys travel as int32 tensors (same bits); helpers below convert
KEY_DTYPE = torch.int32


def keys_from_numpy(a: np.ndarray, device) -> torch.Tensor:
    return torch.from_numpy(np.ascontiguousarray(a, np.uint32).view(np.int32)).to(device)


def keys_to_numpy(t: torch.Tensor) -> np.ndarray:
    return t.detach().cpu().numpy().view(np.uint32)


# ------------------------------------------------------------------------------------------ host-only
def chain_keys(key: np.ndarray, n: int, rng_mode: int):
    """n x (key, sub = split(key)) on the host. Returns (new_key u32[2], subs u32[n,2])."""
    k = np.array(key, dtype=np.uint32).reshape(2).copy()
    subs = np.empty((n, 2), np.uint32)
    _check(load().g2048_chain_keys(k.ctypes.data, subs.ctypes.data, n, rng_mode), "g2048_chain_keys")
    return k, subs


# ------------------------------------------------------------------------------------------ device calls
def split(key, n: int, rng_mode: int, device) -> torch.Tensor:
    out = torch.empty((n, 2), dtype=KEY_DTYPE, device=device)
    _check(load().g2048_split(int(key[0]), int(key[1]), _dev(out, KEY_DTYPE, 2 * n, "out"), n, rng_mode,
                              _stream()), "g2048_split")
    return out


def init(keys, boards, masks, done, rng_mode: int):
    B = masks.numel()
    _check(load().g2048_init(_dev(keys, KEY_DTYPE, 2 * B, "keys"), _dev(boards, u8, 16 * B, "boards"),
                             _dev(masks, u8, B, "masks"), _dev(done, u8, B, "done"), B, rng_mode, _stream()),
           "g2048_init")


def step(boards, masks, done, actions, keys, rewards, rng_mode: int):
    B = masks.numel()
    _check(load().g2048_step(_dev(boards, u8, 16 * B, "boards"), _dev(masks, u8, B, "masks"),
                             _dev(done, u8, B, "done"), _dev(actions, i32, B, "actions"),
                             _dev(keys, KEY_DTYPE, 2 * B, "keys"), _dev(rewards, f32, B, "rewards"), B, rng_mode,
                             _stream()), "g2048_step")


def observe(boards, obs):
    B = boards.numel() // 16
    _check(load().g2048_observe(_dev(boards, u8, 16 * B, "boards"), _dev(obs, u8, 496 * B, "obs"), B, _stream()),
           "g2048_observe")


def act_drul(masks, actions):
    B = masks.numel()
    _check(load().g2048_act_drul(_dev(masks, u8, B, "masks"), _dev(actions, i32, B, "actions"), B, _stream()),
           "g2048_act_drul")


def act_random(keys, masks, actions, log_probs, rng_mode: int):
    B = masks.numel()
    _check(load().g2048_act_random(_dev(keys, KEY_DTYPE, 2 * B, "keys"), _dev(masks, u8, B, "masks"),
                                   _dev(actions, i32, B, "actions"), _dev(log_probs, f32, B, "log_probs"), B,
                                   rng_mode, _stream()), "g2048_act_random")


def act_logits(keys, logits, masks, use_mask, sample, actions, log_probs, rng_mode: int):
    B = masks.numel()
    _check(load().g2048_act_logits(_dev(keys, KEY_DTYPE, 2 * B, "keys"), _dev(logits, f32, 4 * B, "logits"),
                                   _dev(masks, u8, B, "masks"), int(bool(use_mask)), int(bool(sample)),
                                   _dev(actions, i32, B, "actions"), _dev(log_probs, f32, B, "log_probs"), B,
                                   rng_mode, _stream()), "g2048_act_logits")


def reset_fused(sub, boards, masks, done, ep_len, B_total: int, env0: int, rng_mode: int):
    B = masks.numel()
    _check(load().g2048_reset_fused(int(sub[0]), int(sub[1]), _dev(boards, u8, 16 * B, "boards"),
                                    _dev(masks, u8, B, "masks"), _dev(done, u8, B, "done"),
                                    _dev(ep_len, i32, B, "ep_len"), B, B_total, env0, rng_mode, _stream()),
           "g2048_reset_fused")


def rollout_fused(step_subs: np.ndarray, t0: int, boards, masks, done, ep_len, tr_boards, tr_meta, tr_rewards,
                  tr_logp, B_total: int, env0: int, policy: int, fill_frozen: bool, rng_mode: int, live_count):
    B = masks.numel()
    subs = np.ascontiguousarray(step_subs, np.uint32).reshape(-1, 4)
    n = subs.shape[0]
    need = (t0 + n) * B
    _check(load().g2048_rollout_fused(subs.ctypes.data, n, t0, _dev(boards, u8, 16 * B, "boards"),
                                      _dev(masks, u8, B, "masks"), _dev(done, u8, B, "done"),
                                      _dev(ep_len, i32, B, "ep_len"), _dev(tr_boards, u8, 16 * need, "tr_boards"),
                                      _dev(tr_meta, u8, need, "tr_meta"), _dev(tr_rewards, f32, need, "tr_rewards"),
                                      _dev(tr_logp, f32, need, "tr_logp", optional=True), B, B_total, env0, policy,
                                      int(bool(fill_frozen)), rng_mode, _dev(live_count, i32, 1, "live_count"),
                                      _stream()), "g2048_rollout_fused")


def policy_step(act_sub, step_sub, logits, values, use_mask, sample, t: int, boards, masks, done, ep_len, tr_boards,
                tr_meta, tr_rewards, tr_logp, tr_values, B_total: int, env0: int, fill_frozen: bool, rng_mode: int,
                live_count):
    B = masks.numel()
    need = (t + 1) * B
    _check(load().g2048_policy_step(int(act_sub[0]), int(act_sub[1]), int(step_sub[0]), int(step_sub[1]),
                                    _dev(logits, f32, 4 * B, "logits"), _dev(values, f32, B, "values"),
                                    int(bool(use_mask)), int(bool(sample)), t, _dev(boards, u8, 16 * B, "boards"),
                                    _dev(masks, u8, B, "masks"), _dev(done, u8, B, "done"),
                                    _dev(ep_len, i32, B, "ep_len"), _dev(tr_boards, u8, 16 * need, "tr_boards"),
                                    _dev(tr_meta, u8, need, "tr_meta"), _dev(tr_rewards, f32, need, "tr_rewards"),
                                    _dev(tr_logp, f32, need, "tr_logp"), _dev(tr_values, f32, need, "tr_values"),
                                    B, B_total, env0, int(bool(fill_frozen)), rng_mode,
                                    _dev(live_count, i32, 1, "live_count", optional=True), _stream()), "g2048_policy_step")


def policy_step_autoreset(act_sub, step_sub, logits, values, use_mask, sample, t: int, boards, masks, ep_len, tr_boards,
                          tr_meta, tr_rewards, tr_logp, tr_values, B_total: int, env0: int, rng_mode: int):
    """One lock-step of the fixed-horizon mode: every lane steps, a terminated lane starts its next episode at once."""
    B = masks.numel()
    need = (t + 1) * B
    _check(load().g2048_policy_step_autoreset(
        int(act_sub[0]), int(act_sub[1]), int(step_sub[0]), int(step_sub[1]), _dev(logits, f32, 4 * B, "logits"),
        _dev(values, f32, B, "values"), int(bool(use_mask)), int(bool(sample)), t, _dev(boards, u8, 16 * B, "boards"),
        _dev(masks, u8, B, "masks"), _dev(ep_len, i32, B, "ep_len"), _dev(tr_boards, u8, 16 * need, "tr_boards"),
        _dev(tr_meta, u8, need, "tr_meta"), _dev(tr_rewards, f32, need, "tr_rewards"), _dev(tr_logp, f32, need, "tr_logp"),
        _dev(tr_values, f32, need, "tr_values"), B, B_total, env0, rng_mode, _stream()), "g2048_policy_step_autoreset")


def reset_key(step_sub) -> np.ndarray:
    """jax.random.fold_in(step_sub, 0xFFFFFFFF): the reset sub-key of one lock-step of the fixed-horizon mode."""
    out = np.empty(2, np.uint32)
    _check(load().g2048_reset_key(int(step_sub[0]), int(step_sub[1]), out.ctypes.data), "g2048_reset_key")
    return out


def gae_tb_boot(tr_rewards, tr_values, tr_meta, last_values, tr_adv, tr_ret, T: int, B: int, gamma: float, lam: float):
    n = T * B
    _check(load().g2048_gae_tb_boot(_dev(tr_rewards, f32, n, "tr_rewards"), _dev(tr_values, f32, n, "tr_values"),
                                    _dev(tr_meta, u8, n, "tr_meta"), _dev(last_values, f32, B, "last_values"),
                                    _dev(tr_adv, f32, n, "tr_adv"), _dev(tr_ret, f32, n, "tr_ret"), T, B, float(gamma),
                                    float(lam), _stream()), "g2048_gae_tb_boot")


def gae_tb(tr_rewards, tr_values, ep_len, tr_adv, tr_ret, T: int, B: int, gamma: float, lam: float):
    n = T * B
    _check(load().g2048_gae_tb(_dev(tr_rewards, f32, n, "tr_rewards"), _dev(tr_values, f32, n, "tr_values"),
                               _dev(ep_len, i32, B, "ep_len"), _dev(tr_adv, f32, n, "tr_adv"),
                               _dev(tr_ret, f32, n, "tr_ret"), T, B, float(gamma), float(lam), _stream()),
           "g2048_gae_tb")


def gae_flat(rewards, values, terms, adv, ret, gamma: float, lam: float):
    N = rewards.numel()
    _check(load().g2048_gae_flat(_dev(rewards, f32, N, "rewards"), _dev(values, f32, N, "values"),
                                 _dev(terms, u8, N, "terms"), _dev(adv, f32, N, "adv"), _dev(ret, f32, N, "ret"), N,
                                 float(gamma), float(lam), _stream()), "g2048_gae_flat")


def compact(tr_boards, tr_meta, tr_rewards, tr_logp, tr_values, ep_len, offsets, out_boards, out_actions, out_masks,
            out_rewards, out_logp, out_values, out_terms, T: int, B: int, N: int, tr_adv=None, tr_ret=None, out_adv=None,
            out_ret=None):
    n = T * B
    _check(load().g2048_compact(
        _dev(tr_boards, u8, 16 * n, "tr_boards"), _dev(tr_meta, u8, n, "tr_meta"),
        _dev(tr_rewards, f32, n, "tr_rewards"), _dev(tr_logp, f32, n, "tr_logp", optional=True),
        _dev(tr_values, f32, n, "tr_values", optional=True), _dev(tr_adv, f32, n, "tr_adv", optional=True),
        _dev(tr_ret, f32, n, "tr_ret", optional=True), _dev(ep_len, i32, B, "ep_len"),
        _dev(offsets, i64, B, "offsets"), _dev(out_boards, u8, 16 * N, "out_boards"),
        _dev(out_actions, u8, N, "out_actions"), _dev(out_masks, u8, N, "out_masks"),
        _dev(out_rewards, f32, N, "out_rewards"), _dev(out_logp, f32, N, "out_logp", optional=True),
        _dev(out_values, f32, N, "out_values", optional=True), _dev(out_adv, f32, N, "out_adv", optional=True),
        _dev(out_ret, f32, N, "out_ret", optional=True), _dev(out_terms, u8, N, "out_terms"), T, B,
        _stream()), "g2048_compact")


def policy_encoder_workspace_bytes(B: int) -> int:
    return int(load().g2048_policy_encoder_workspace_bytes(B))


def policy_encoder(boards, embed_table, cls_token, weights_bf16, params_f32, n_layers: int, features, workspace=None):
    """workspace: None (single kernel) or a uint8 device tensor of policy_encoder_workspace_bytes(B) bytes (the last
    layer then runs CLS-only in a second kernel)."""
    B = boards.numel() // 16
    if workspace is not None:
        need = policy_encoder_workspace_bytes(B)
        if not workspace.is_cuda or workspace.dtype != u8 or workspace.numel() < need or not workspace.is_contiguous():
            raise NativeError(f"workspace: expected a contiguous uint8 device tensor of >= {need} bytes")
    _check(load().g2048_policy_encoder(
        _dev(boards, u8, 16 * B, "boards"), _dev(embed_table, f32, 16 * 31 * 256, "embed_table"),
        _dev(cls_token, f32, 256, "cls_token"), _dev(weights_bf16, torch.bfloat16, n_layers * 786432, "weights_bf16"),
        _dev(params_f32, f32, n_layers * 3328, "params_f32"), n_layers, _dev(features, f32, 256 * B, "features"), B,
        None if workspace is None else workspace.data_ptr(), _stream()), "g2048_policy_encoder")


def attn_fwd(q_ptr: int, k_ptr: int, v_ptr: int, o, lse, B: int, H: int, Sq: int, strides, scale: float, p_drop: float,
             seed: int, seed_state: int = 0):
    """q/k/v: raw device addresses inside bf16 tensors the caller keeps alive; strides = (q_sb, q_ss, k_sb, k_ss,
    v_sb, v_ss) in elements."""
    _check(load().g2048_attn_fwd(q_ptr, k_ptr, v_ptr, _dev(o, torch.bfloat16, B * Sq * H * 32, "o"),
                                 _dev(lse, f32, B * H * Sq, "lse"), B, H, Sq, *[int(x) for x in strides], float(scale),
                                 float(p_drop), int(seed), seed_state or None, _stream()), "g2048_attn_fwd")


def attn_bwd(q_ptr: int, k_ptr: int, v_ptr: int, dout, lse, dq_ptr: int, dk_ptr: int, dv_ptr: int, B: int, H: int,
             Sq: int, strides, scale: float, p_drop: float, seed: int, seed_state: int = 0):
    _check(load().g2048_attn_bwd(q_ptr, k_ptr, v_ptr, _dev(dout, torch.bfloat16, B * Sq * H * 32, "dout"),
                                 _dev(lse, f32, B * H * Sq, "lse"), dq_ptr, dk_ptr, dv_ptr, B, H, Sq,
                                 *[int(x) for x in strides], float(scale), float(p_drop), int(seed), seed_state or None,
                                 _stream()), "g2048_attn_bwd")


def add_ln_fwd(x_ptr: int, x_row_stride: int, a, gamma, beta, x_new, h, mean, rstd, T: int, eps: float, p_drop: float,
               seed: int, seed_state: int = 0):
    """x_ptr: raw device address of f32 rows (stride x_row_stride elements) the caller keeps alive.  gamma None: no
    LayerNorm, h = bf16(x + dropout(a)); beta, mean, rstd, x_new may then be None."""
    bf = torch.bfloat16
    _check(load().g2048_add_ln_fwd(x_ptr, int(x_row_stride), _dev(a, bf, 256 * T, "a", optional=True),
                                   _dev(gamma, f32, 256, "gamma", optional=True), _dev(beta, f32, 256, "beta", optional=True),
                                   _dev(x_new, f32, 256 * T, "x_new", optional=True), _dev(h, bf, 256 * T, "h"),
                                   _dev(mean, f32, T, "mean", optional=True), _dev(rstd, f32, T, "rstd", optional=True), T,
                                   float(eps), float(p_drop),
                                   int(seed), seed_state or None, _stream()), "g2048_add_ln_fwd")


def add_ln_bwd(xn_ptr: int, x_row_stride: int, g_x, g_h, mean, rstd, gamma, dx, da, dparams, T: int,
               p_drop: float, seed: int, seed_state: int = 0, g_x_period: int = 1):
    """dparams f32 [3, 256]: dgamma, dbeta, column sums of da.  dparams None: first stage only -> the workspace, f32
    [rows, 768] partial sums (for ``reduce_jobs``).  gamma None: the forward had no LayerNorm (xn_ptr 0, mean/rstd None)."""
    bf = torch.bfloat16
    ws = torch.empty(load().g2048_add_ln_bwd_workspace_floats(T), dtype=f32, device=dx.device)
    _check(load().g2048_add_ln_bwd(xn_ptr, int(x_row_stride), _dev(g_x, f32, 256 * (T // g_x_period), "g_x", optional=True),
                                   _dev(g_h, bf, 256 * T, "g_h"), _dev(mean, f32, T, "mean", optional=True),
                                   _dev(rstd, f32, T, "rstd", optional=True), _dev(gamma, f32, 256, "gamma", optional=True),
                                   _dev(dx, f32, 256 * T, "dx"),
                                   _dev(da, bf, 256 * T, "da", optional=True), _dev(dparams, f32, 768, "dparams", optional=True),
                                   ws.data_ptr(), T, float(p_drop), int(seed), seed_state or None, int(g_x_period), _stream()),
           "g2048_add_ln_bwd")
    return ws.view(-1, 768) if dparams is None else None


def rowgemm_ok(u2: torch.Tensor, w_packed: torch.Tensor, tile_stride: int = 0) -> bool:
    """Operands ``g2048_linear_add_ln_fwd / _bwd`` take: u2 bf16 [T, K] with unit column stride, K a multiple of 256, the packed weight
    a flat bf16 tensor of 256 * K elements - or, with ``tile_stride`` (elements between the 32-row tiles of a WIDER packed matrix), a
    flat view that starts at the first of K / 16 consecutive k-steps of it."""
    if u2.dim() != 2 or not u2.is_cuda or u2.dtype != torch.bfloat16 or u2.stride(1) != 1:
        return False
    T, K = u2.shape
    span = 256 * K if not tile_stride else 7 * int(tile_stride) + (K // 16) * 512
    return (T > 0 and K % 256 == 0 and 256 <= K <= 1024 and u2.stride(0) >= K and u2.stride(0) % 8 == 0 and u2.data_ptr() % 16 == 0
            and w_packed is not None and w_packed.is_cuda and w_packed.dtype == torch.bfloat16 and w_packed.is_contiguous()
            and (w_packed.numel() == span if not tile_stride else (w_packed.numel() >= span and tile_stride >= (K // 16) * 512
                                                                  and tile_stride % 8 == 0))
            and w_packed.data_ptr() % 16 == 0 and 160 * u2.stride(0) * 2 < 2 ** 31)


def linear_add_ln_fwd(u2, w_packed, bias, x_ptr: int, x_row_stride: int, gamma, beta, x_new, h, mean, rstd, eps: float,
                      p_drop: float, seed: int, seed_state: int = 0):
    """x_new = x + dropout(u2 W^T + bias); h = bf16(LayerNorm(x_new)) in one launch (``g2048_linear_add_ln_fwd``).  u2 bf16 [T, K];
    w_packed: fragment-packed bf16 [256][K] (``pack_fragments``); x_ptr: raw address of the f32 residual rows (stride in elements)."""
    if not rowgemm_ok(u2, w_packed):
        raise NativeError(f"linear_add_ln_fwd: operands {tuple(u2.shape)} (strides {u2.stride()}) x packed {w_packed.numel()} not supported")
    T, K = u2.shape
    bf = torch.bfloat16
    _check(load().g2048_linear_add_ln_fwd(u2.data_ptr(), u2.stride(0), w_packed.data_ptr(), _dev(bias, f32, 256, "bias", optional=True),
                                          K, x_ptr, int(x_row_stride), _dev(gamma, f32, 256, "gamma"), _dev(beta, f32, 256, "beta"),
                                          _dev(x_new, f32, 256 * T, "x_new"), _dev(h, bf, 256 * T, "h"), _dev(mean, f32, T, "mean"),
                                          _dev(rstd, f32, T, "rstd"), T, float(eps), float(p_drop), int(seed), seed_state or None,
                                          _stream()), "g2048_linear_add_ln_fwd")


def linear_add_ln_bwd(dy2, wt_packed, xn_ptr: int, x_row_stride: int, g_x, mean, rstd, gamma, dx, da, p_drop: float, seed: int,
                      seed_state: int = 0, g_x_period: int = 1, tile_stride: int = 0, g_h_extra=None, extra_period: int = 1):
    """g_h = bf16(dy2 Wt^T) and the add+LayerNorm backward on it in one launch (``g2048_linear_add_ln_bwd``) -> f32 [rows, 768] partial
    sums (dgamma | dbeta | column sums of da) for ``reduce_jobs``.  wt_packed: the fragment-packed TRANSPOSE [256][K] of the weight of
    the Linear that consumed h (``tile_stride``: see ``rowgemm_ok``).  ``g_h_extra`` bf16 [T / extra_period, 256]: added to g_h on the
    rows tok % extra_period == 0."""
    if not rowgemm_ok(dy2, wt_packed, tile_stride):
        raise NativeError(f"linear_add_ln_bwd: operands {tuple(dy2.shape)} (strides {dy2.stride()}) x packed {wt_packed.numel()} not supported")
    T, K = dy2.shape
    bf = torch.bfloat16
    ws = torch.empty((int(load().g2048_linear_add_ln_bwd_partial_rows(T)), 768), dtype=f32, device=dx.device)
    _check(load().g2048_linear_add_ln_bwd(dy2.data_ptr(), dy2.stride(0), wt_packed.data_ptr(), int(tile_stride), K, xn_ptr,
                                          int(x_row_stride), _dev(g_x, f32, 256 * (T // g_x_period), "g_x", optional=True),
                                          int(g_x_period),
                                          _dev(g_h_extra, bf, 256 * (T // max(int(extra_period), 1)), "g_h_extra", optional=True),
                                          int(extra_period), _dev(mean, f32, T, "mean"), _dev(rstd, f32, T, "rstd"), _dev(gamma, f32, 256, "gamma"),
                                          _dev(dx, f32, 256 * T, "dx"), _dev(da, bf, 256 * T, "da", optional=True), ws.data_ptr(), T,
                                          float(p_drop), int(seed), seed_state or None, _stream()), "g2048_linear_add_ln_bwd")
    return ws


# ---- MLP policy (configs[1]): csrc/g2048_mlp.hip ---------------------------------------------------------------------------------
GEMM_MAX_JOBS = 8


class GemmJob(C.Structure):
    _fields_ = [("x", _vp * 2), ("ldx", _i64 * 2), ("w", _vp * 2), ("ldw", _i64 * 2), ("k", _i32 * 2), ("bias", _vp), ("act", _vp),
                ("ldact", _i64), ("y", _vp), ("ldy", _i64), ("N", _i32), ("relu", _i32)]


def _bf16_rows(t: torch.Tensor, name: str):
    if not t.is_cuda or t.dtype != torch.bfloat16 or t.dim() != 2 or t.stride(1) != 1 or t.stride(0) % 8 or t.data_ptr() % 16:
        raise NativeError(f"{name}: expected a bf16 [rows, cols] device tensor with unit column stride, 16-byte aligned rows; got "
                          f"{tuple(t.shape)} {t.dtype} strides {t.stride()}")
    return t.data_ptr(), t.stride(0)


def gemm_jobs(jobs, M: int):
    """``g2048_gemm_jobs``: jobs = list of dicts ``segs`` [(x [M, k] bf16, w [N, k] bf16), ...] (one or two), ``y`` [M, N] bf16, optional
    ``bias`` f32 [N], ``relu`` bool, ``act`` bf16 [M, N] (the output is multiplied by (act > 0)).  Column-slice views are fine."""
    recs = []
    for q in jobs:
        segs = q["segs"]
        if not 1 <= len(segs) <= 2:
            raise NativeError("gemm_jobs: one or two K-segments per job")
        y = q["y"]
        yp, ldy = _bf16_rows(y, "y")
        N = y.shape[1]
        xs, ldxs, ws, ldws, ks = [0, 0], [0, 0], [0, 0], [0, 0], [0, 0]
        for i, (x, w) in enumerate(segs):
            xs[i], ldxs[i] = _bf16_rows(x, "x")
            ws[i], ldws[i] = _bf16_rows(w, "w")
            ks[i] = x.shape[1]
            if x.shape[0] != M or y.shape[0] != M or w.shape != (N, ks[i]) or ks[i] % 64 or N % 64:
                raise NativeError(f"gemm_jobs: shapes x {tuple(x.shape)} w {tuple(w.shape)} y {tuple(y.shape)} (M = {M})")
        act = q.get("act")
        ap, lda = (0, 0) if act is None else _bf16_rows(act, "act")
        if act is not None and tuple(act.shape) != (M, N):
            raise NativeError("gemm_jobs: act must be [M, N]")
        bias = q.get("bias")
        recs.append(GemmJob((_vp * 2)(*xs), (_i64 * 2)(*ldxs), (_vp * 2)(*ws), (_i64 * 2)(*ldws), (_i32 * 2)(*ks),
                            None if bias is None else _dev(bias, f32, N, "bias"), ap or None, lda, yp, ldy, N, int(bool(q.get("relu")))))
    if not 1 <= len(recs) <= GEMM_MAX_JOBS:
        raise NativeError(f"gemm_jobs: 1..{GEMM_MAX_JOBS} jobs per launch")
    arr = (GemmJob * len(recs))(*recs)
    _check(load().g2048_gemm_jobs(C.cast(arr, _vp), len(recs), int(M), _stream()), "g2048_gemm_jobs")


def mlp_embed_fwd(boards, wt, bias, y, onehot=None):
    """y[M, 512] (bf16) = relu(bias + sum of the 16 selected rows of wt [496, 512] (bf16, the transposed trunk_in weight)); ``onehot``
    (bf16 [M, 512]) receives the one-hot matrix."""
    M = boards.shape[0]
    bf = torch.bfloat16
    _check(load().g2048_mlp_embed_fwd(_dev(boards, u8, 16 * M, "boards"), _dev(wt, bf, 496 * 512, "wt"), _dev(bias, f32, 512, "bias"),
                                      _dev(y, bf, 512 * M, "y"), _dev(onehot, bf, 512 * M, "onehot", optional=True), M, _stream()),
           "g2048_mlp_embed_fwd")


def mlp_out_fwd(h2, w3, logits, values):
    M = h2.shape[0]
    bf = torch.bfloat16
    _check(load().g2048_mlp_out_fwd(_dev(h2, bf, 1024 * M, "h2"), _dev(w3, bf, 5 * 512, "w3"), _dev(logits, f32, 4 * M, "logits"),
                                    _dev(values, f32, M, "values"), M, _stream()), "g2048_mlp_out_fwd")


def mlp_out_bwd(dlogits, dvalues, h2, w3, dh2):
    """-> partial f32 [rows, 5, 512]: first-stage sums of the output layers' weight gradients."""
    M = h2.shape[0]
    bf = torch.bfloat16
    ws = torch.empty((int(load().g2048_mlp_out_bwd_partial_rows(M)), 5, 512), dtype=f32, device=h2.device)
    _check(load().g2048_mlp_out_bwd(_dev(dlogits, f32, 4 * M, "dlogits"), _dev(dvalues, f32, M, "dvalues"), _dev(h2, bf, 1024 * M, "h2"),
                                    _dev(w3, bf, 5 * 512, "w3"), _dev(dh2, bf, 1024 * M, "dh2"), ws.data_ptr(), M, _stream()),
           "g2048_mlp_out_bwd")
    return ws


def relu_dropout_fwd(x, y, p_drop: float, seed: int, seed_state: int = 0):
    bf = torch.bfloat16
    F = x.shape[-1]
    T = x.numel() // F
    _check(load().g2048_relu_dropout_fwd(_dev(x, bf, T * F, "x"), _dev(y, bf, T * F, "y"), T, F, float(p_drop), int(seed),
                                         seed_state or None, _stream()), "g2048_relu_dropout_fwd")


def relu_dropout_bwd(dy, y, dx, dbias, p_drop: float):
    """dbias None: first stage only -> the workspace, f32 [rows, F] partial sums (for ``reduce_jobs``)."""
    bf = torch.bfloat16
    F = y.shape[-1]
    T = y.numel() // F
    ws = torch.empty(load().g2048_relu_dropout_bwd_workspace_floats(T, F), dtype=f32, device=y.device)
    _check(load().g2048_relu_dropout_bwd(_dev(dy, bf, T * F, "dy"), _dev(y, bf, T * F, "y"), _dev(dx, bf, T * F, "dx"),
                                         _dev(dbias, f32, F, "dbias", optional=True), ws.data_ptr(), T, F, float(p_drop),
                                         _stream()), "g2048_relu_dropout_bwd")
    return ws.view(-1, F) if dbias is None else None


COLSUM_MAX_GROUPS = 512


def colsum(x: torch.Tensor, out: torch.Tensor | None = None) -> torch.Tensor:
    """f32 column sums of a 2-D bf16/f32 device tensor whose rows are contiguous (any row stride)."""
    if not x.is_cuda or x.dim() != 2 or x.stride(1) != 1 or x.dtype not in (torch.bfloat16, f32):
        raise NativeError(f"colsum: expected a 2-D bf16/f32 device tensor with contiguous rows, got {x.dtype} {tuple(x.shape)} "
                          f"strides {x.stride()}")
    T, N = x.shape
    if out is None:
        out = torch.empty(N, dtype=f32, device=x.device)
    ws = torch.empty(load().g2048_colsum_workspace_floats(T, N), dtype=f32, device=x.device)
    _check(load().g2048_colsum(x.data_ptr(), int(x.dtype == torch.bfloat16), x.stride(0), T, N, ws.data_ptr(),
                               _dev(out, f32, N, "out"), _stream()), "g2048_colsum")
    return out


def colsum_partial(x: torch.Tensor) -> torch.Tensor:
    """First stage of ``colsum`` only -> f32 [rows, N] partial sums (for ``reduce_jobs``); N <= 1024."""
    if not x.is_cuda or x.dim() != 2 or x.stride(1) != 1 or x.dtype not in (torch.bfloat16, f32):
        raise NativeError(f"colsum_partial: expected a 2-D bf16/f32 device tensor with contiguous rows, got {x.dtype} "
                          f"{tuple(x.shape)} strides {x.stride()}")
    T, N = x.shape
    rows = load().g2048_colsum_partial_rows(T, N)
    if rows <= 0:
        raise NativeError(f"colsum_partial: unsupported shape {tuple(x.shape)}")
    ws = torch.empty(load().g2048_colsum_workspace_floats(T, N), dtype=f32, device=x.device)
    _check(load().g2048_colsum(x.data_ptr(), int(x.dtype == torch.bfloat16), x.stride(0), T, N, ws.data_ptr(), None, _stream()),
           "g2048_colsum")
    return ws[:rows * N].view(rows, N)


class ReduceJob(C.Structure):  # g2048_reduce_job
    _fields_ = [("src", _vp), ("dst", _vp), ("part_stride", _i64), ("n", C.c_int32), ("parts", C.c_int32),
                ("src_bf16", C.c_int32), ("transpose_rows", C.c_int32)]


def reduce_jobs(jobs):
    """jobs: list of (src tensor whose first element is part 0 / column 0, dst f32 tensor, part_stride, n, parts[,
    transpose_rows]): dst[c] = sum_p src[p * part_stride + c] (transpose_rows R: the [R][n / R] sum stored as [n / R][R]); all
    of them in one launch (per 64)."""
    if not jobs:
        return
    recs = []
    for job in jobs:
        src, dst, stride, n, parts = job[:5]
        tr = int(job[5]) if len(job) > 5 else 0
        if not src.is_cuda or src.dtype not in (torch.bfloat16, f32) or not dst.is_cuda or dst.dtype != f32 \
                or not dst.is_contiguous() or dst.numel() < n:
            raise NativeError(f"reduce_jobs: bad job {src.dtype} {tuple(src.shape)} -> {dst.dtype} {tuple(dst.shape)} (n={n})")
        recs.append(ReduceJob(src.data_ptr(), dst.data_ptr(), int(stride), int(n), int(parts), int(src.dtype == torch.bfloat16), tr))
    arr = (ReduceJob * len(recs))(*recs)
    _check(load().g2048_reduce_jobs(C.cast(arr, _vp), len(recs), _stream()), "g2048_reduce_jobs")


def ppo_loss(logits, values, actions, mask_bits, old_logp, adv, ret, clip_eps: float, c_value: float, c_entropy: float,
             grad_scale=None, running=None):
    """-> (new_logp f32 [M], sums f32 [5], dlogits like logits, dvalues like values); see g2048_ppo_loss.  ``grad_scale``:
    optional device f32 scalar the two gradients are multiplied by; ``running``: optional device f64 [5] the kernel adds the
    five means to."""
    M = actions.numel()
    for name, t in (("logits", logits), ("values", values)):
        if not t.is_cuda or t.dtype not in (torch.bfloat16, f32) or not t.is_contiguous():
            raise NativeError(f"{name}: expected a contiguous bf16/f32 device tensor, got {t.dtype} on {t.device}")
    if logits.numel() != 4 * M or values.numel() != M:
        raise NativeError(f"ppo_loss: logits {tuple(logits.shape)} / values {tuple(values.shape)} do not match M={M}")
    new_logp = torch.empty(M, dtype=f32, device=logits.device)
    sums = torch.empty(5, dtype=f32, device=logits.device)
    dlogits, dvalues = torch.empty_like(logits), torch.empty_like(values)
    _check(load().g2048_ppo_loss(
        logits.data_ptr(), int(logits.dtype == torch.bfloat16), values.data_ptr(), int(values.dtype == torch.bfloat16),
        _dev(actions, u8, M, "actions"), _dev(mask_bits, u8, M, "mask_bits", optional=True), _dev(old_logp, f32, M, "old_logp"),
        _dev(adv, f32, M, "adv"), _dev(ret, f32, M, "ret"), M, float(clip_eps), float(c_value), float(c_entropy),
        new_logp.data_ptr(), sums.data_ptr(), dlogits.data_ptr(), dvalues.data_ptr(),
        _dev(grad_scale, f32, 1, "grad_scale", optional=True), _dev(running, torch.float64, 5, "running", optional=True),
        _stream()), "g2048_ppo_loss")
    return new_logp, sums, dlogits, dvalues


def linear_ok(x2: torch.Tensor, weight: torch.Tensor) -> bool:
    """Shapes/strides g2048_linear_bf16 takes: x2 [T, K] and weight [N, K] bf16 on the device with contiguous rows."""
    return (x2.is_cuda and x2.dtype == torch.bfloat16 and weight.dtype == torch.bfloat16 and x2.dim() == 2
            and weight.dim() == 2 and x2.stride(1) == 1 and weight.stride(1) == 1 and x2.shape[1] == weight.shape[1]
            and x2.shape[1] % 128 == 0 and weight.shape[0] % 128 == 0 and x2.stride(0) % 8 == 0 and weight.stride(0) % 8 == 0
            and x2.data_ptr() % 16 == 0 and weight.data_ptr() % 16 == 0 and x2.shape[0] > 0)


def linear_bf16(x2: torch.Tensor, weight: torch.Tensor, bias_f32=None, out=None) -> torch.Tensor:
    """out[T, N] (bf16) = x2[T, K] @ weight[N, K]^T (+ bias f32 [N])."""
    if not linear_ok(x2, weight):
        raise NativeError(f"linear_bf16: unsupported operands {tuple(x2.shape)} {x2.dtype} x {tuple(weight.shape)} {weight.dtype}")
    T, K = x2.shape
    N = weight.shape[0]
    if out is None:
        out = torch.empty((T, N), dtype=torch.bfloat16, device=x2.device)
    _check(load().g2048_linear_bf16(x2.data_ptr(), x2.stride(0), weight.data_ptr(), weight.stride(0),
                                    _dev(bias_f32, f32, N, "bias", optional=True), _dev(out, torch.bfloat16, T * N, "out"), N, T,
                                    K, N, _stream()), "g2048_linear_bf16")
    return out


def linear_relu_dropout(x2: torch.Tensor, weight: torch.Tensor, bias_f32: torch.Tensor, p_drop: float, seed: int = 0,
                        seed_state: int = 0, want_mask: bool = False):
    """dropout(relu(x2 @ weight^T + bias)) -> bf16 [T, N] in one launch (K <= 256); see g2048_linear_relu_dropout_bf16.
    ``want_mask``: -> (y, mask) with the opaque bit mask ``linear_mask_bwd`` reads."""
    if not linear_ok(x2, weight) or x2.shape[1] > 256:
        raise NativeError(f"linear_relu_dropout: unsupported operands {tuple(x2.shape)} {x2.dtype} x {tuple(weight.shape)}")
    T, K = x2.shape
    N = weight.shape[0]
    out = torch.empty((T, N), dtype=torch.bfloat16, device=x2.device)
    mask = torch.empty(load().g2048_ffn_mask_bytes(T, N), dtype=u8, device=x2.device) if want_mask else None
    _check(load().g2048_linear_relu_dropout_bf16(x2.data_ptr(), x2.stride(0), weight.data_ptr(), weight.stride(0),
                                                 _dev(bias_f32, f32, N, "bias"), out.data_ptr(), N, T, K, N, float(p_drop),
                                                 int(seed) & (2 ** 64 - 1), seed_state or None,
                                                 mask.data_ptr() if want_mask else None, _stream()),
           "g2048_linear_relu_dropout_bf16")
    return (out, mask) if want_mask else out


def linear_mask_bwd(dy2: torch.Tensor, weight_t: torch.Tensor, mask: torch.Tensor, p_drop: float, final: bool = True):
    """-> (dz bf16 [T, N], dbias f32 [N]): dz = (dy2 @ weight_t^T) / (1 - p) where the forward's output was non-zero
    (``mask`` from ``linear_relu_dropout(..., want_mask=True)`` with the same T and N); see g2048_linear_mask_bwd_bf16.
    ``final`` False: (dz, partial sums f32 [rows, N]) for ``reduce_jobs``."""
    if not linear_ok(dy2, weight_t) or dy2.shape[1] > 256:
        raise NativeError(f"linear_mask_bwd: unsupported operands {tuple(dy2.shape)} {dy2.dtype} x {tuple(weight_t.shape)}")
    T, K = dy2.shape
    N = weight_t.shape[0]
    if mask.dtype != u8 or not mask.is_cuda or mask.numel() != load().g2048_ffn_mask_bytes(T, N) or mask.data_ptr() % 8:
        raise NativeError(f"linear_mask_bwd: the mask does not belong to a [{T}, {N}] forward call")
    dz = torch.empty((T, N), dtype=torch.bfloat16, device=dy2.device)
    db = torch.empty(N, dtype=f32, device=dy2.device) if final else None
    ws = torch.empty(load().g2048_linear_mask_bwd_workspace_floats(T, N), dtype=f32, device=dy2.device)
    _check(load().g2048_linear_mask_bwd_bf16(dy2.data_ptr(), dy2.stride(0), weight_t.data_ptr(), weight_t.stride(0),
                                             mask.data_ptr(), dz.data_ptr(), N, db.data_ptr() if final else None, ws.data_ptr(),
                                             T, K, N, float(p_drop), _stream()), "g2048_linear_mask_bwd_bf16")
    if final:
        return dz, db
    rows = load().g2048_linear_mask_bwd_partial_rows(T, N)
    return dz, ws[:rows * N].view(rows, N)


def embed_fwd(boards, wt, pe, cls, x0, p_drop: float = 0.0, seed: int = 0, seed_state: int = 0, ln=None):
    """wt: the class-major table f32 [31, 256], or the nn.Linear weight f32 [256, 31] itself (read in place).  ``ln`` = (gamma f32
    [256], beta f32 [256], eps, h bf16 [M, 17, 256], mean f32 [M * 17], rstd f32 [M * 17]): also the LayerNorm of every row
    (``g2048_embed_ln_fwd``)."""
    M = boards.numel() // 16
    w_ld = 0 if tuple(wt.shape) == (31, 256) else int(wt.shape[1])
    if w_ld and (wt.dim() != 2 or wt.shape[0] != 256 or w_ld < 31):
        raise NativeError(f"embed_fwd: weight must be [31, 256] or [256, >= 31], got {tuple(wt.shape)}")
    if ln is not None:
        gamma, beta, eps, h, mean, rstd = ln
        _check(load().g2048_embed_ln_fwd(_dev(boards, u8, 16 * M, "boards"), _dev(wt, f32, 31 * 256, "wt"), w_ld,
                                         _dev(pe, f32, 16 * 256, "pe"), _dev(cls, f32, 256, "cls"), _dev(x0, f32, M * 17 * 256, "x0"), M,
                                         float(p_drop), int(seed), seed_state or None, _dev(gamma, f32, 256, "gamma"),
                                         _dev(beta, f32, 256, "beta"), float(eps), _dev(h, torch.bfloat16, M * 17 * 256, "h"),
                                         _dev(mean, f32, M * 17, "mean"), _dev(rstd, f32, M * 17, "rstd"), _stream()), "g2048_embed_ln_fwd")
        return
    _check(load().g2048_embed_fwd(_dev(boards, u8, 16 * M, "boards"), _dev(wt, f32, 31 * 256, "wt"), w_ld,
                                  _dev(pe, f32, 16 * 256, "pe"),
                                  _dev(cls, f32, 256, "cls"), _dev(x0, f32, M * 17 * 256, "x0"), M, float(p_drop), int(seed),
                                  seed_state or None, _stream()), "g2048_embed_fwd")


def embed_bwd(boards, dx0, dwt_dcls, p_drop: float = 0.0, seed: int = 0, seed_state: int = 0):
    """dwt_dcls None: first stage only -> the workspace, f32 [rows, 32 * 256] partial sums (for ``reduce_jobs``)."""
    M = boards.numel() // 16
    ws = torch.empty(load().g2048_embed_bwd_workspace_floats(M), dtype=f32, device=dx0.device)
    _check(load().g2048_embed_bwd(_dev(boards, u8, 16 * M, "boards"), _dev(dx0, f32, M * 17 * 256, "dx0"),
                                  _dev(dwt_dcls, f32, 32 * 256, "dwt_dcls", optional=True), ws.data_ptr(), M, float(p_drop),
                                  int(seed), seed_state or None, _stream()), "g2048_embed_bwd")
    return ws.view(-1, 32 * 256) if dwt_dcls is None else None


OPT_CHUNK = 2048  # G2048_OPT_CHUNK
OPT_MAX_GROUPS = 4  # G2048_OPT_MAX_GROUPS


class OptChunk(C.Structure):  # g2048_opt_chunk
    _fields_ = [("param", _vp), ("offset", _i64), ("n", C.c_int32), ("group", C.c_int32), ("shadow", _vp), ("shadow_t", _vp),
                ("shadow_p", _vp), ("shadow_tp", _vp), ("e0", C.c_int32), ("rows", C.c_int32), ("cols", C.c_int32),
                ("reserved", C.c_int32)]


OPT_CHUNK_BYTES = C.sizeof(OptChunk)


class OptGroup(C.Structure):  # g2048_opt_group
    _fields_ = [("lr", _dbl), ("beta1", _dbl), ("beta2", _dbl), ("eps", _dbl), ("weight_decay", _dbl)]


def opt_chunk_table(params, offsets, groups, device, shadows=None) -> torch.Tensor:
    """The device-resident chunk table of g2048_opt_step (u8 tensor holding g2048_opt_chunk records): parameter i (a
    contiguous f32 device tensor) has its gradient/moments at element offsets[i] of the flat buffers and belongs to
    hyper-parameter group groups[i].  ``shadows``: optional {id(parameter): (bf16 copy, bf16 transposed copy or None[, bf16
    fragment-packed copy or None, bf16 fragment-packed transposed copy or None])} the step keeps up to date."""
    recs = []
    shadows = shadows or {}
    for p, off, grp in zip(params, offsets, groups):
        if not p.is_cuda or p.dtype != f32 or not p.is_contiguous() or p.data_ptr() % 16 or off % 4:
            raise NativeError(f"opt_chunk_table: parameter {tuple(p.shape)} {p.dtype} on {p.device} (offset {off}) is not a "
                              "contiguous 16-byte aligned f32 device tensor at a flat offset that is a multiple of 4")
        n = p.numel()
        sh, sh_t, sh_p, sh_tp = (tuple(shadows.get(id(p), ())) + (None,) * 4)[:4]
        for t in (sh, sh_t, sh_p, sh_tp):
            if t is not None and (t.dtype != torch.bfloat16 or not t.is_cuda or not t.is_contiguous() or t.numel() != n):
                raise NativeError(f"opt_chunk_table: shadow of a {tuple(p.shape)} parameter must be a contiguous bf16 tensor of {n} elements")
        if sh_t is not None and (p.dim() != 2 or tuple(sh_t.shape) != tuple(p.shape[::-1])):
            raise NativeError("opt_chunk_table: a transposed shadow needs a 2-D parameter and the transposed shape")
        if sh_p is not None and (p.dim() != 2 or p.shape[0] % 32 or p.shape[1] % 16 or sh_p.data_ptr() % 16):
            raise NativeError("opt_chunk_table: a packed shadow needs a 2-D parameter with rows % 32 == 0 and cols % 16 == 0")
        if sh_tp is not None and (p.dim() != 2 or p.shape[1] % 32 or p.shape[0] % 16 or sh_tp.data_ptr() % 16):
            raise NativeError("opt_chunk_table: a packed transposed shadow needs a 2-D parameter with cols % 32 == 0 and rows % 16 == 0")
        rows, cols = (p.shape[0], p.shape[1]) if p.dim() == 2 else (1, max(n, 1))
        for c0 in range(0, n, OPT_CHUNK):
            recs.append(OptChunk(p.data_ptr() + 4 * c0, off + c0, min(OPT_CHUNK, n - c0), grp, sh.data_ptr() if sh is not None else None,
                                 sh_t.data_ptr() if sh_t is not None else None, sh_p.data_ptr() if sh_p is not None else None,
                                 sh_tp.data_ptr() if sh_tp is not None else None, c0, rows, cols, 0))
    arr = (OptChunk * len(recs))(*recs)
    host = torch.frombuffer(bytearray(bytes(arr)), dtype=u8)
    return host.to(device)


def opt_workspace(n_chunks: int, device) -> torch.Tensor:
    """Zeroed workspace for g2048_opt_step (partials + the completion counter the kernel keeps at zero)."""
    return torch.zeros(load().g2048_opt_workspace_floats(n_chunks), dtype=f32, device=device)


def opt_step(table, n_chunks: int, grads, exp_avg, exp_avg_sq, groups, max_grad_norm: float, steps, scale, growth_tracker,
             growth: float, backoff: float, growth_interval: int, workspace, info=None):
    """Clip + AdamW (+ GradScaler unscale / skip / update when ``scale`` is given) for every chunk of ``table``; ``groups``:
    list of (lr, beta1, beta2, eps, weight_decay)."""
    if len(groups) > OPT_MAX_GROUPS:
        raise NativeError(f"opt_step: {len(groups)} parameter groups, at most {OPT_MAX_GROUPS} are supported")
    g = (OptGroup * len(groups))(*[OptGroup(*map(float, t)) for t in groups])
    n = grads.numel()
    _check(load().g2048_opt_step(
        table.data_ptr(), n_chunks, _dev(grads, f32, None, "grads"), _dev(exp_avg, f32, n, "exp_avg"),
        _dev(exp_avg_sq, f32, n, "exp_avg_sq"), C.cast(g, _vp), len(groups), float(max_grad_norm if max_grad_norm else 0.0),
        _dev(steps, f32, 1, "steps"), steps.numel(), _dev(scale, f32, 1, "scale", optional=True),
        _dev(growth_tracker, torch.int32, 1, "growth_tracker", optional=True), float(growth), float(backoff),
        int(growth_interval), _dev(workspace, f32, None, "workspace"), _dev(info, f32, 2, "info", optional=True), _stream()),
        "g2048_opt_step")


def gather_minibatch(idx, boards, actions, masks, logp, adv, ret, out=None):
    """-> dict(obs u8 [M,16], actions u8 [M], masks u8 [M], old_lp, adv, ret f32 [M]) = rows idx of the buffer (``out``:
    pre-allocated tensors of that layout, e.g. the static inputs of a captured graph)."""
    M, N = idx.numel(), actions.numel()
    dev = boards.device
    if out is None:
        out = dict(obs=torch.empty((M, 16), dtype=u8, device=dev), actions=torch.empty(M, dtype=u8, device=dev),
                   masks=torch.empty(M, dtype=u8, device=dev), old_lp=torch.empty(M, dtype=f32, device=dev),
                   adv=torch.empty(M, dtype=f32, device=dev), ret=torch.empty(M, dtype=f32, device=dev))
    _check(load().g2048_gather_minibatch(
        _dev(idx, i64, M, "idx"), M, N, _dev(boards, u8, 16 * N, "boards"), _dev(actions, u8, N, "actions"),
        _dev(masks, u8, N, "masks"), _dev(logp, f32, N, "logp"), _dev(adv, f32, N, "adv"), _dev(ret, f32, N, "ret"),
        _dev(out["obs"], u8, 16 * M, "o_boards"), _dev(out["actions"], u8, M, "o_actions"), _dev(out["masks"], u8, M, "o_masks"),
        _dev(out["old_lp"], f32, M, "o_logp"), _dev(out["adv"], f32, M, "o_adv"), _dev(out["ret"], f32, M, "o_ret"), _stream()),
        "g2048_gather_minibatch")
    return out


# ---------------------------------------------------------------------------------------------------------------------
# the 2048-row tail of the update (csrc/g2048_tail.hip)
# ---------------------------------------------------------------------------------------------------------------------
TAIL_MASK_TILES = 96
DW_MAX_JOBS = 16


def pack_fragments(x: torch.Tensor) -> torch.Tensor:
    """Row-major [rows, cols] (rows % 32 == 0, cols % 16 == 0) -> the fragment-packed layout of include/g2048.h, as a flat tensor."""
    R, Cc = x.shape
    return x.reshape(R // 32, 32, Cc // 16, 2, 8).permute(0, 2, 3, 1, 4).reshape(-1).contiguous()


def unpack_fragments(flat: torch.Tensor, rows: int, cols: int) -> torch.Tensor:
    """Inverse of ``pack_fragments``: flat packed tensor -> row-major [rows, cols]."""
    return flat.reshape(rows // 32, cols // 16, 2, 32, 8).permute(0, 3, 1, 2, 4).reshape(rows, cols)


class TailWeights(C.Structure):
    _fields_ = [(n, _vp) for n in ("wo", "w1", "w2", "a1", "a2", "a3", "c1", "c2", "c3", "bo", "b1", "b2", "ab1", "ab2", "cb1", "cb2",
                                   "ln_g", "ln_b")]


class TailWeightsT(C.Structure):
    _fields_ = [(n, _vp) for n in ("woT", "w1T", "w2T", "a1T", "a2T", "a3", "c1T", "c2T", "c3", "ln_g")]


class TailSaved(C.Structure):
    _fields_ = [(n, _vp) for n in ("x_mid", "mean", "rstd", "masks", "oT", "h2T", "uT", "featsT", "a1T", "a2T", "c1T", "c2T")] \
        + [("ld", _i64)]


class TailGrads(C.Structure):
    _fields_ = [(n, _vp) for n in ("daoT", "dzT", "df2T", "da1T", "da2T", "dlT", "dc1T", "dc2T", "dvT", "ln_partial")]


class DwJob(C.Structure):
    _fields_ = [("dyT", _vp), ("xT", _vp), ("dw", _vp), ("db", _vp), ("N", C.c_int32), ("K", C.c_int32)]


_TAIL_SAVED_ROWS = dict(oT=256, h2T=256, uT=1024, featsT=256, a1T=512, a2T=512, c1T=512, c2T=512)
_TAIL_GRAD_ROWS = dict(daoT=256, dzT=1024, df2T=256, da1T=512, da2T=512, dlT=32, dc1T=512, dc2T=512, dvT=32)


class TailBuffers:
    """Device buffers of one g2048_cls_tail_fwd / _bwd pair for M rows (allocated once per minibatch size and reused: every
    element that is read is rewritten by each forward / backward; dlT / dvT rows that are never written stay zero)."""

    def unpacked(self, name: str) -> torch.Tensor:
        """Row-major [rows, ld] view-copy of a packed transposed buffer (tests / debugging)."""
        t = self.saved[name] if name in self.saved else self.grads[name]
        return unpack_fragments(t, self.rows[name], self.ld)

    def __init__(self, M: int, device):
        bf = torch.bfloat16
        self.M, self.blocks = int(M), (int(M) + 31) // 32
        self.ld = 32 * self.blocks
        # slices of the row axis in g2048_dweight_t: 8 (one per XCD) when every slice is a whole number of 16-row k-steps
        self.slices = next(s for s in (8, 4, 2, 1) if self.ld % (16 * s) == 0)
        z = lambda *shape, dtype=bf: torch.zeros(shape, dtype=dtype, device=device)
        self.saved = dict(x_mid=z(self.M, 256, dtype=f32), mean=z(self.M, dtype=f32), rstd=z(self.M, dtype=f32),
                          masks=z(self.blocks, TAIL_MASK_TILES, 64, dtype=torch.int16))
        # transposed activations / gradients, fragment-packed (``unpack_fragments(t, rows, ld)`` gives [rows, ld]): flat tensors
        self.saved.update({k: z(rows * self.ld) for k, rows in _TAIL_SAVED_ROWS.items()})
        self.grads = {k: z(rows * self.ld) for k, rows in _TAIL_GRAD_ROWS.items()}
        self.rows = dict(_TAIL_SAVED_ROWS, **_TAIL_GRAD_ROWS)
        self.grads["ln_partial"] = z(self.blocks, 512, dtype=f32)
        self.saved_c = TailSaved(*[self.saved[n].data_ptr() for n, _ in TailSaved._fields_[:-1]], self.ld)
        self.grads_c = TailGrads(*[self.grads[n].data_ptr() for n, _ in TailGrads._fields_])
        self.logits, self.values = z(self.M, 4, dtype=f32), z(self.M, dtype=f32)
        self.d_o, self.dx_cls = z(self.M, 256), z(self.M, 256, dtype=f32)
        # weight-gradient partials [slices][N][K] / [slices][N] per Linear: (dyT key, xT key, N (padded to 32), K, has bias)
        self.dw_spec = dict(wo=("daoT", "oT", 256, 256, True), w1=("dzT", "h2T", 1024, 256, True), w2=("df2T", "uT", 256, 1024, True),
                            a1=("da1T", "featsT", 512, 256, True), a2=("da2T", "a1T", 512, 512, True), a3=("dlT", "a2T", 32, 512, False),
                            c1=("dc1T", "featsT", 512, 256, True), c2=("dc2T", "c1T", 512, 512, True), c3=("dvT", "c2T", 32, 512, False))
        self.dw = {k: z(self.slices, n, kk, dtype=f32) for k, (_, _, n, kk, _) in self.dw_spec.items()}
        self.db = {k: z(self.slices, n, dtype=f32) for k, (_, _, n, _, b) in self.dw_spec.items() if b}


def _ptr_struct(cls, tensors: dict, dtypes: dict):
    vals = []
    for name, _ in cls._fields_:
        t = tensors[name]
        want = dtypes.get(name, torch.bfloat16)
        if not t.is_cuda or t.dtype != want or not t.is_contiguous() or t.data_ptr() % 16:
            raise NativeError(f"{cls.__name__}.{name}: expected a contiguous, 16-byte aligned {want} device tensor, got {t.dtype} "
                              f"on {t.device}")
        vals.append(t.data_ptr())
    return cls(*vals)


_TAIL_F32 = {n: f32 for n in ("bo", "b1", "b2", "ab1", "ab2", "cb1", "cb2", "ln_g", "ln_b")}


def tail_weights(tensors: dict) -> TailWeights:
    """bf16 weights (nn.Linear layout) + f32 biases / norm2 parameters by field name -> the C struct (the caller keeps them alive)."""
    return _ptr_struct(TailWeights, tensors, _TAIL_F32)


def tail_weights_t(tensors: dict) -> TailWeightsT:
    return _ptr_struct(TailWeightsT, tensors, _TAIL_F32)


def cls_tail_fwd(o, x_cls_ptr: int, x_row_stride: int, W: TailWeights, buf: TailBuffers, eps: float, p_drop: float, seed: int,
                 seed_state: int = 0):
    """o bf16 [M, 256]; x_cls_ptr: raw device address of f32 rows (element stride x_row_stride) the caller keeps alive.
    -> (buf.logits f32 [M, 4], buf.values f32 [M])."""
    _check(load().g2048_cls_tail_fwd(_dev(o, torch.bfloat16, 256 * buf.M, "o"), x_cls_ptr, int(x_row_stride), C.byref(W),
                                     C.byref(buf.saved_c), buf.logits.data_ptr(), buf.values.data_ptr(), buf.M, float(eps),
                                     float(p_drop), int(seed), seed_state or None, _stream()), "g2048_cls_tail_fwd")
    return buf.logits, buf.values


def cls_tail_bwd(dlogits, dvalues, WT: TailWeightsT, buf: TailBuffers, p_drop: float, seed: int, seed_state: int = 0):
    """-> (buf.d_o bf16 [M, 256], buf.dx_cls f32 [M, 256]); dY^T of every Linear and the LayerNorm partials land in buf.grads."""
    _check(load().g2048_cls_tail_bwd(_dev(dlogits, f32, 4 * buf.M, "dlogits"), _dev(dvalues, f32, buf.M, "dvalues"), C.byref(WT),
                                     C.byref(buf.saved_c), C.byref(buf.grads_c), buf.d_o.data_ptr(), buf.dx_cls.data_ptr(), buf.M,
                                     float(p_drop), int(seed), seed_state or None, _stream()), "g2048_cls_tail_bwd")
    return buf.d_o, buf.dx_cls


def dweight_ok(dy2: torch.Tensor, x2: torch.Tensor, slices: int) -> bool:
    """Operands ``g2048_dweight_bf16`` takes: bf16 [T, N] / [T, K] row views (unit inner stride), N and K multiples of 128,
    T a multiple of 64 * slices."""
    if dy2.dim() != 2 or x2.dim() != 2 or dy2.shape[0] != x2.shape[0]:
        return False
    T, N, K = dy2.shape[0], dy2.shape[1], x2.shape[1]
    for t in (dy2, x2):
        if not t.is_cuda or t.dtype != torch.bfloat16 or t.stride(1) != 1 or t.stride(0) % 8 or t.data_ptr() % 16 or t.stride(0) >= 1 << 27:
            return False
    return N % 128 == 0 and K % 128 == 0 and T > 0 and T % (64 * slices) == 0 and (slices < 8 or slices % 8 == 0)


def dweight_parts(dy2: torch.Tensor, x2: torch.Tensor, slices: int, out: torch.Tensor = None, block_rows: int = 0, colsum: bool = False):
    """bf16 [slices, N, K] whose sum over the first axis is dY^T X (``g2048_dweight_bf16``); with ``colsum`` also f32 [slices, N] whose
    sum over the first axis is ``dy2.sum(0)``: -> (parts, column sums)."""
    if not dweight_ok(dy2, x2, slices):
        raise NativeError(f"dweight_parts: operands {tuple(dy2.shape)} x {tuple(x2.shape)} with {slices} slices are not supported")
    T, N, K = dy2.shape[0], dy2.shape[1], x2.shape[1]
    if out is None:
        out = torch.empty((slices, N, K), dtype=torch.bfloat16, device=dy2.device)
    cs = torch.empty((slices, N), dtype=f32, device=dy2.device) if colsum else None
    _check(load().g2048_dweight_bf16(dy2.data_ptr(), dy2.stride(0), x2.data_ptr(), x2.stride(0), _dev(out, torch.bfloat16, slices * N * K, "parts"),
                                     None if cs is None else cs.data_ptr(), T, N, K, int(slices), int(block_rows), _stream()),
           "g2048_dweight_bf16")
    return (out, cs) if colsum else out


class DwgJob(C.Structure):
    _fields_ = [("dy", _vp), ("x", _vp), ("parts", _vp), ("colsum", _vp), ("lddy", _i64), ("ldx", _i64), ("T", _i64),
                ("N", _i32), ("K", _i32), ("slices", _i32), ("parts_f32", _i32)]


DWG_MAX_JOBS = 16


def dweight_jobs(jobs):
    """Several ``dweight_parts`` products in one launch per ``DWG_MAX_JOBS`` (``g2048_dweight_jobs``).  jobs: (dy2, x2, parts bf16
    (or f32: the job's ``parts_f32`` flag) [slices, N, K], colsum f32 [slices, N] or None); slices (= parts.shape[0]) a multiple of 8."""
    recs = []
    for dy2, x2, parts, cs in jobs:
        slices = parts.shape[0]
        if not dweight_ok(dy2, x2, slices) or slices % 8:
            raise NativeError(f"dweight_jobs: operands {tuple(dy2.shape)} x {tuple(x2.shape)} with {slices} slices are not supported")
        T, N, K = dy2.shape[0], dy2.shape[1], x2.shape[1]
        if parts.dtype not in (torch.bfloat16, f32):
            raise NativeError(f"dweight_jobs: parts must be bf16 or f32, got {parts.dtype}")
        recs.append(DwgJob(dy2.data_ptr(), x2.data_ptr(), _dev(parts, parts.dtype, slices * N * K, "parts"),
                           None if cs is None else _dev(cs, f32, slices * N, "colsum"), dy2.stride(0), x2.stride(0), T, N, K, slices,
                           int(parts.dtype == f32)))
    for i in range(0, len(recs), DWG_MAX_JOBS):
        chunk = recs[i:i + DWG_MAX_JOBS]
        arr = (DwgJob * len(chunk))(*chunk)
        _check(load().g2048_dweight_jobs(C.cast(arr, _vp), len(chunk), _stream()), "g2048_dweight_jobs")


def dweight_t(jobs, ld: int, m: int, slices: int):
    """jobs: list of (dyT, xT, dw f32 [slices, N, K], db f32 [slices, N] or None); dyT / xT: fragment-packed bf16 of N * ld / K * ld
    elements (``pack_fragments`` of the [N, ld] / [K, ld] matrices)."""
    recs = []
    for dyT, xT, dw, db in jobs:
        N, K = dw.shape[1], dw.shape[2]
        for t, dt, name in ((dyT, torch.bfloat16, "dyT"), (xT, torch.bfloat16, "xT"), (dw, f32, "dw")):
            if not t.is_cuda or t.dtype != dt or not t.is_contiguous():
                raise NativeError(f"dweight_t: {name} must be a contiguous {dt} device tensor")
        if dyT.numel() != N * ld or xT.numel() != K * ld or dw.shape[0] != slices or (db is not None and db.numel() != slices * N):
            raise NativeError(f"dweight_t: {dyT.numel()} x {xT.numel()} elements -> {tuple(dw.shape)} do not fit ld={ld}")
        recs.append(DwJob(dyT.data_ptr(), xT.data_ptr(), dw.data_ptr(), None if db is None else _dev(db, f32, slices * N, "db"), N, K))
    arr = (DwJob * len(recs))(*recs)
    _check(load().g2048_dweight_t(C.cast(arr, _vp), len(recs), int(ld), int(m), int(slices), _stream()), "g2048_dweight_t")

