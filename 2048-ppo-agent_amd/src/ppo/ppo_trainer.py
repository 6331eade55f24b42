"""PPO training loop (API of the reference src/ppo/ppo_trainer.py:21-727) on the device-resident pipeline.

collect:  BatchRunner.collect -> engine Trajectory in HBM -> RolloutBuffer.store_trajectory (HIP compaction)
update:   PPODataset (HIP GAE scan + z-scoring on the device) -> DeviceBatches -> clipped-surrogate/value/entropy
          loss in PyTorch-ROCm (bf16 autocast, GradScaler, global-norm clip, AdamW/LAMB, per-minibatch LR step)
Multi-GPU (torch.distributed initialised, backend nccl = RCCL): rollout_batch_size is the GLOBAL number of envs,
each rank steps its contiguous slice with the global key indices (bit-identical to the single-device boards);
per minibatch ONE all-reduce of the flat gradient bucket; advantage/return statistics and the KL early-stop
test are reduced globally so every rank takes the same decisions.
"""
from __future__ import annotations

import contextlib
import logging
import os
from collections import deque
from typing import Dict, Literal, Optional

import numpy as np
import torch
import torch.distributed as dist
import torch.nn.functional as F
from torch.amp import GradScaler, autocast

from ..optim.flat_step import FlatAdamWStep
from ..optim import configure_bert_optimizers
from ..runs.batch_runner import BatchRunner
from .data_loader import DeviceBatches, PPODataset
from .rollout_buffer import RolloutBuffer
from .torch_action_wrapper import TorchActionFunction
from . import torch_compat
from .capture import capture as capture_graph
from .hip_ops import Bf16Shadow, GradSink, grad_sink, graph_seed_state

logger = logging.getLogger(__name__)


def resolve_rollout_amp(rollout_amp, mixed_precision, environ=None) -> bool:
    """Whether the rollout forward runs in the update's autocast dtype: an explicit argument wins; else G2048_ROLLOUT_AMP=0/1
    (round-2 switch); else bf16 mixed precision implies the bf16 rollout unless G2048_ROLLOUT_FP32=1 (the reference's fp32)."""
    if rollout_amp is not None:
        return bool(rollout_amp)
    environ = os.environ if environ is None else environ
    on = ("1", "true", "yes", "on")
    env_amp = environ.get("G2048_ROLLOUT_AMP", "").strip().lower()
    if env_amp:
        return env_amp in on
    return mixed_precision == "bfloat16" and environ.get("G2048_ROLLOUT_FP32", "0").strip().lower() not in on


class _NullWriter:
    """Stand-in when tensorboard is not installed: same calls, no output."""

    def add_scalar(self, *a, **k):
        pass

    def add_histogram(self, *a, **k):
        pass

    def close(self):
        pass


def _make_writer(log_dir: str):
    try:
        from torch.utils.tensorboard import SummaryWriter

        return SummaryWriter(log_dir)
    except Exception:  # tensorboard missing
        return _NullWriter()


class _History(deque):
    """deque that also answers slices (``hist[-100:]``): the reference slices its deque in
    ppo_trainer.py:237-239,627,715 and run/train_ppo_agent.py:132, which raises TypeError there."""

    def __getitem__(self, idx):
        if isinstance(idx, slice):
            return list(self)[idx]
        return super().__getitem__(idx)


def _tail(seq, k: int):
    """Last k entries of a deque/list (the reference slices a deque here, which raises TypeError)."""
    items = list(seq)
    return items[-k:] if k > 0 else []


class _FusedPPOLoss(torch.autograd.Function):
    """``_compute_ppo_loss`` in one HIP launch (``g2048_ppo_loss``): forward, the logged means and the gradient.

    -> (sums f32 [5] = mean policy, value, entropy, TOTAL loss, mean(old - new log-prob); new_log_probs [M]).
    Only ``sums[3]`` is differentiable: backward scales the gradients the kernel already produced.  With ``grad_scale`` (the
    GradScaler's device scalar) the kernel itself multiplies them by it and backward passes them on untouched: the caller
    then seeds backward with d(total) = 1 (``sums.backward(selector)``) and gets the gradients of ``grad_scale * total``,
    i.e. ``scaler.scale(loss).backward()``, without the two scaling launches per minibatch.  ``running`` (device f64 [5]): the
    kernel also adds the five means to it (the update's logged sums: one launch less per minibatch)."""

    @staticmethod
    def forward(ctx, logits, values, actions_u8, mask_bits, old_lp, adv, ret, clip_eps, c_value, c_entropy, grad_scale=None,
                running=None):
        from ..g2048 import native as nv

        new_lp, sums, dlogits, dvalues = nv.ppo_loss(logits.contiguous(), values.contiguous(), actions_u8, mask_bits, old_lp,
                                                     adv, ret, clip_eps, c_value, c_entropy, grad_scale, running)
        ctx.save_for_backward(dlogits, dvalues)
        ctx.prescaled = grad_scale is not None
        ctx.mark_non_differentiable(new_lp)
        ctx.set_materialize_grads(False)  # no zero-filled [M] gradient for new_lp (one fill launch per minibatch)
        return sums, new_lp

    @staticmethod
    def backward(ctx, g_sums, _g_new_lp):
        dlogits, dvalues = ctx.saved_tensors
        if ctx.prescaled:
            return (dlogits, dvalues) + (None,) * 10
        g = g_sums[3].to(dlogits.dtype)
        return (dlogits * g, dvalues * g.to(dvalues.dtype)) + (None,) * 10


class _GraphedFwdBwd:
    """One minibatch of ``_compute_ppo_loss`` + (scaled) backward captured in a hipGraph.

    Inputs are copied into static buffers, the graph is replayed, gradients land in static ``.grad`` tensors
    (or in the flat all-reduce bucket when it exists); clipping, the optimizer and the LR schedule stay eager.
    Build it while no autograd graph over the agent's parameters is alive (see ``PPOTrainer._eager_fwd_bwd``).
    Dropout draws fresh masks on every replay (philox offsets for PyTorch's kernels, ``graph_seed_state`` for ours)."""

    def __init__(self, trainer: "PPOTrainer", M: int, sample: dict, capture_allreduce: bool = False):
        self.tr, self.M = trainer, M
        self.allreduce_captured = bool(capture_allreduce)
        dev = trainer.device
        self.static = {k: torch.empty_like(v) for k, v in sample.items()}
        for k, v in sample.items():
            self.static[k].copy_(v)
        side = torch.cuda.Stream(device=dev)
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            trainer._running_sums()  # (allocated outside the capture)
            trainer._acc_in_kernel = False  # (the warm-up passes must not count in the update's logged sums)
            try:
                for _ in range(3):  # warm-up off the capture stream (allocator, autocast caches, lazy inits)
                    self._zero()
                    self._fwd_bwd()
            finally:
                trainer._acc_in_kernel = True
        torch.cuda.current_stream().wait_stream(side)
        trainer.optimizer.zero_grad(set_to_none=True)  # backward inside the capture creates the (static) gradients
        self.graph = torch.cuda.CUDAGraph()
        if trainer._flat_step is not None:  # the optimiser kernel keeps the bf16 shadows current: no cast kernels in the graph
            trainer._flat_step.adopt_shadows(list(Bf16Shadow._live))
        Bf16Shadow.invalidate_all()  # (for the others) the bf16 weight refresh must be part of the graph
        seed_word = graph_seed_state(dev)  # allocated outside the capture
        # capture.capture(): thread-local error mode, and no cyclic garbage collection while the stream is capturing (a
        # collection that finalises an earlier trainer's CUDAGraph / graph pool on this thread aborts the process: capture.py)
        trainer._capturing_allreduce = trainer._early_armed = bool(capture_allreduce)
        try:
            with capture_graph(self.graph):
                if any(isinstance(m, torch.nn.Dropout) and m.p > 0 for m in trainer.agent.modules()):
                    seed_word.add_(1)  # new dropout masks on every replay (the HIP kernels read it at run time); an agent
                    # without dropout (the MLP policy) does not pay the launch
                self.out = self._fwd_bwd()
                if trainer._flat_grad is not None:
                    trainer._collect_grads(point_grads=False)  # graph-owned gradients -> all-reduce bucket, one copy
                    if capture_allreduce:
                        trainer._allreduce_bucket()  # the step's collective(s) as nodes of the graph
        finally:
            trainer._capturing_allreduce = trainer._early_armed = False
            trainer._early_work = None
        Bf16Shadow.invalidate_all()  # nothing was copied during the capture itself
        # parameters that received no gradient inside the capture: a replay leaves zeros in their bucket slices (the captured
        # ``v.zero_()``), and after it every ``p.grad`` points somewhere, so the list cannot be recomputed from ``p.grad``; the
        # optimiser step after a replay takes it from here (torch's AdamW leaves such parameters alone: no decay, no moments)
        self.no_grad = list(getattr(trainer, "_no_grad", ())) if trainer._flat_grad is not None else []
        # the gradients the replay writes (graph pool) / the tensors the optimizer reads after a replay
        self.grads = [p.grad for p in trainer.agent.parameters()]
        if trainer._flat_grad is not None:
            for p, v in zip(trainer._params, trainer._flat_views):
                p.grad = v
            self.grads = [p.grad for p in trainer.agent.parameters()]

    def _zero(self):
        self.tr._zero_grad()

    def _fwd_bwd(self):
        st = self.static
        # autocast's weight-cast cache must be off inside a captured region (PyTorch CUDA-graphs + AMP rule)
        return self.tr._loss_backward(st["obs"], st["actions"], st["masks"], st["old_lp"], st["adv"], st["ret"],
                                      cache_enabled=False, zero=False)

    def run(self, batch: dict):
        for k, v in batch.items():
            self.static[k].copy_(v)
        return self.replay()

    def replay(self):
        """Replay on whatever ``self.static`` holds (the minibatch gather kernel writes there directly)."""
        # A replay runs no Python: shadows that the optimiser kernel maintains have no cast kernel inside the graph.  If somebody
        # else changed a parameter since (agent.load_state_dict, a manual edit, a broadcast), refresh them eagerly first (host-side
        # key comparison only; no launch in the normal case).
        fs = self.tr._flat_step
        if fs is not None:
            for sh in fs._shadows:
                if sh.key != sh.current_key():
                    sh()
        self.graph.replay()
        for p, g in zip(self.tr.agent.parameters(), self.grads):  # an eager step in between may have re-pointed them
            p.grad = g
        return self.out


class PPOTrainer:
    def __init__(self, agent, batch_runner: BatchRunner, rollout_buffer: RolloutBuffer, optimizer_param_dict: Dict,
                 max_steps: int, gamma: float = 0.99, lambda_gae: float = 0.95, clip_epsilon: float = 0.2,
                 value_loss_coef: float = 0.5, entropy_coef: float = 0.01, max_grad_norm: float = 0.5,
                 target_kl: float = 0.01, use_action_mask: bool = False, device: torch.device = torch.device("cpu"),
                 mixed_precision: Optional[Literal["float16", "bfloat16"]] = "bfloat16",
                 max_samples_per_epoch: int = None, shuffle_on_reset: bool = False,
                 rollout_amp: Optional[bool] = None,
                 log_dir: str = "logs", use_hip_graph: Optional[bool] = None, rollout_mode: Optional[str] = None,
                 rollout_horizon: Optional[int] = None, allreduce_dtype: Optional[str] = None,
                 allreduce_in_graph: Optional[bool] = None):
        self.agent = agent.to(device)
        self._acc_in_kernel, self._acc5 = True, None  # see _running_sums
        self.batch_runner = batch_runner
        self.rollout_buffer = rollout_buffer
        self.gamma, self.lambda_gae, self.clip_epsilon = gamma, lambda_gae, clip_epsilon
        self.value_loss_coef, self.entropy_coef = value_loss_coef, entropy_coef
        self.max_grad_norm, self.target_kl = max_grad_norm, target_kl
        self.use_action_mask = use_action_mask
        self.device = torch.device(device)
        self.max_samples_per_epoch, self.shuffle_on_reset = max_samples_per_epoch, shuffle_on_reset

        self.mixed_precision = mixed_precision
        self.use_amp = mixed_precision is not None and self.device.type == "cuda"
        if self.device.type == "cuda":
            torch_compat.check()  # the private torch interfaces of the device path: one clear error instead of a late one
        if self.use_amp:
            self.scaler = GradScaler()
            self.amp_dtype = torch.float16 if mixed_precision == "float16" else torch.bfloat16
        else:
            self.scaler, self.amp_dtype = None, None
            if mixed_precision is not None:
                logger.warning("Mixed precision requested but device is %s; disabled.", self.device.type)
        # Rollout precision.  The reference rolls out in fp32 (its torch_action_wrapper.py has no autocast) while its trainer
        # config sets mixed_precision: bfloat16 for the update.  Here ``mixed_precision="bfloat16"`` implies the bf16 rollout
        # forward as well -- for the reference's default model shape that is the fused MFMA encoder kernel
        # (g2048_policy_encoder), 12x the collect rate of the fp32 module forward -- so that an unmodified
        # run/train_ppo_agent.py gets the fast path.  The numeric deviation (rollout log-probs / values at bf16-autocast
        # distance from the fp32 forward, i.e. the same distance the update's own forward has) is documented in
        # INTEGRATION.md.  ``rollout_amp=False`` or G2048_ROLLOUT_FP32=1 keeps the reference's fp32 rollout;
        # G2048_ROLLOUT_AMP=0/1 (round-2 switch) is still honoured.
        implied = rollout_amp is None and not os.environ.get("G2048_ROLLOUT_AMP", "").strip()
        rollout_amp = resolve_rollout_amp(rollout_amp, mixed_precision)
        self.rollout_amp = bool(rollout_amp) and self.use_amp
        if self.rollout_amp and implied:
            logger.warning("rollout forward runs in %s (implied by mixed_precision; the reference rolls out in fp32): stored "
                           "log-probs / values are at autocast distance from the fp32 forward.  G2048_ROLLOUT_FP32=1 or "
                           "rollout_amp=False restores the reference's precision.", mixed_precision)
        # How collect_rollouts gathers experience.  "episodes" (default) is the reference: lock-step batches of complete
        # episodes (src/runs/batch_runner.py:117), finished envs idle until the slowest one ends.  "fixed_horizon" is the
        # throughput mode: every env steps rollout_horizon times per batch, an env whose episode ends starts the next one at
        # once, and GAE bootstraps from V(s_T) where the horizon cut an episode.  For the unmodified reference CLI:
        # G2048_ROLLOUT_MODE=fixed_horizon [G2048_ROLLOUT_HORIZON=128].
        if rollout_mode is None:
            rollout_mode = os.environ.get("G2048_ROLLOUT_MODE", "episodes").strip().lower()
        if rollout_mode not in ("episodes", "fixed_horizon"):
            raise ValueError(f"rollout_mode must be 'episodes' or 'fixed_horizon', got {rollout_mode!r}")
        self.rollout_mode = rollout_mode
        if rollout_horizon is None:
            rollout_horizon = int(os.environ.get("G2048_ROLLOUT_HORIZON", "128"))
        self.rollout_horizon = int(rollout_horizon)

        opt = configure_bert_optimizers(self.agent, steps=max_steps, **dict(optimizer_param_dict))
        self.optimizer = opt["optimizer"]
        self.lr_scheduler = opt["lr_scheduler"]["scheduler"]

        self.world = dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1
        self.rank = dist.get_rank() if self.world > 1 else 0
        self._group = dist.group.WORLD if self.world > 1 else None
        self._flat_grad = None
        # The one collective per optimiser step.  allreduce_dtype "float32" (default: exact mean of the ranks' f32 gradients) or
        # "bfloat16" (G2048_ALLREDUCE_DTYPE=bf16): the bucket travels as bf16 -- 7.9 MB instead of 15.8 MB per step over xGMI, two
        # cast launches extra; gradients are still loss-scaled there, bf16 has f32's exponent range, so nothing overflows and
        # every element keeps 8 significant bits (what the bf16 backward produced it with anyway).
        if allreduce_dtype is None:
            allreduce_dtype = os.environ.get("G2048_ALLREDUCE_DTYPE", "float32").strip().lower()
        allreduce_dtype = {"bf16": "bfloat16", "fp32": "float32", "f32": "float32"}.get(allreduce_dtype, allreduce_dtype)
        if allreduce_dtype not in ("float32", "bfloat16"):
            raise ValueError(f"allreduce_dtype must be 'float32' or 'bfloat16', got {allreduce_dtype!r}")
        self.allreduce_dtype = torch.bfloat16 if allreduce_dtype == "bfloat16" else torch.float32
        self._comm_buf = None
        # allreduce_in_graph (G2048_ALLREDUCE_IN_GRAPH=1; opt-in): capture the collective inside the update's hipGraph (RCCL
        # collectives are capturable), so that a replay launches it without host involvement.  If that capture fails the graph is
        # captured again WITHOUT the collective (never the eager update), and ``allreduce_in_graph_fallback`` says why.
        if allreduce_in_graph is None:
            allreduce_in_graph = os.environ.get("G2048_ALLREDUCE_IN_GRAPH", "0").strip().lower() in ("1", "true", "yes", "on")
        self.allreduce_in_graph = bool(allreduce_in_graph) and self.world > 1 and self.device.type == "cuda"
        self.allreduce_in_graph_fallback = None
        # Two all-reduce buckets (world > 1; G2048_ALLREDUCE_BUCKETS=1 keeps the single one, =2 forces the split also on one
        # rank, for tests): the parameters whose gradients are final after the first third of the backward (the agent's
        # ``early_grad_parameters()``: the fused CLS tail, 1.4 M of 3.96 M) come FIRST in the flat bucket; where the collective
        # can start mid-backward (eager update, or captured inside the hipGraph) their slice is summed and all-reduced while the
        # three full layers' backward still runs.  Same result as one bucket, element for element (a sum over ranks per element).
        nb = os.environ.get("G2048_ALLREDUCE_BUCKETS", "").strip()
        early = list(getattr(self.agent, "early_grad_parameters", lambda: [])()) if (nb == "2" or (nb != "1" and self.world > 1)) else []
        self._early_params = [p for p in early if p.requires_grad]
        self._split_always = nb == "2"
        self._early_n, self._early_work, self._early_launched = 0, None, 0
        # armed only around a forward+backward whose gradients ARE all-reduced afterwards (update_policy's eager minibatch, a capture
        # that includes the collectives): a stray backward (graph warm-up, a test calling _loss_backward) must not start one
        self._early_armed = False
        # clip + AdamW + GradScaler bookkeeping as two launches over flat buffers (g2048_opt_step) for the reference's
        # default optimiser on the device; anything else (LAMB, Adam, CPU) takes the PyTorch calls of the reference
        self._flat_step = None
        if os.environ.get("G2048_FLAT_OPT", "1").strip().lower() not in ("0", "false", "no", "off") \
                and FlatAdamWStep.supports(self.optimizer, self.device):
            self._flat_step = FlatAdamWStep(self.optimizer, self.device, first=self._early_params)
        if self.world > 1 or self._flat_step is not None:
            self._bind_flat_grads()
        if self.world > 1:
            self._broadcast_parameters()

        # forward + loss + backward of one minibatch replayed as a hipGraph: the update is ~330 small kernels per
        # minibatch and launch-bound in eager mode (7.0 ms of host time for 4.7 ms of GPU work at minibatch 2048).
        # Default (None): on for the bf16 update of agents whose large row reductions all run through g2048_colsum
        # (``hip_graph_safe``): at::sum's cross-workgroup stage zeroes its semaphores with a memset that a replayed
        # hipGraph does not reproduce on ROCm 7.2 / torch 2.10 (wrong bias gradients on every batch but the
        # captured one; tools/debug_graph_grads.py).  Built lazily per minibatch size, see _GraphedFwdBwd.
        if use_hip_graph is None:
            use_hip_graph = (self.use_amp and self.amp_dtype == torch.bfloat16
                             and getattr(self.agent, "hip_graph_safe", False))
        self.use_hip_graph = bool(use_hip_graph) and self.device.type == "cuda"
        # second stage of all gradient reductions in one launch (GradSink); needs the flat gradient bucket
        self.grad_sink = self.device.type == "cuda" and \
            os.environ.get("G2048_GRAD_SINK", "1").strip().lower() not in ("0", "false", "no", "off")
        self._graphs = {}
        self._rollout_graphs = {}  # captured rollout forwards of agents that ask for it (``rollout_graph_ok``: the MLP policy)
        self.hip_graph_fallback = None  # repr of the exception that made _build_graph drop to the eager update

        self.writer = _make_writer(log_dir) if self.rank == 0 else _NullWriter()
        self.total_timesteps = 0
        self.total_epochs = 0
        self.total_update_steps = 0
        hist = max_samples_per_epoch if max_samples_per_epoch is not None else 10000
        self.episode_rewards = _History(maxlen=hist)
        self.episode_lengths = _History(maxlen=hist)
        self.last_save_timestep = 0
        self.load_checkpoint_path = None
        self.last_rollout_stats: Dict[str, float] = {}
        self._fixed_reward_carry = None  # fixed_horizon mode: running max reward of every lane's unfinished episode

    @property
    def rollout_graph_fallback(self):
        """repr of the exception that made the captured rollout forward (TorchActionFunction._graphed) drop to eager mode."""
        return self._rollout_graphs.get("fallback")

    # ------------------------------------------------------------------ multi-GPU plumbing
    def _bind_flat_grads(self):
        """One contiguous gradient bucket for the single all-reduce per step; ``_flat_views[i]`` is the slice of
        parameter i.  Backward writes fresh ``.grad`` tensors (no accumulate kernel per parameter) which
        ``_collect_grads`` gathers into the bucket with one multi-tensor copy and then re-points ``.grad`` at."""
        if self._flat_step is not None:  # the optimiser kernel's gradient buffer IS the bucket
            self._params, self._flat_grad = self._flat_step.params, self._flat_step.grad
            self._flat_views = self._flat_step.grad_views
            self._early_n = self._flat_step.n_first
            self._flat_grad.zero_()
            for p in self._params:
                p.grad = None
            self._sink_targets = {id(p): v for p, v in zip(self._params, self._flat_views)}
            self._alloc_comm_buf()
            return
        self._params = [p for p in self.agent.parameters() if p.requires_grad]
        early_ids = {id(p) for p in self._early_params}
        self._params = [p for p in self._params if id(p) in early_ids] + [p for p in self._params if id(p) not in early_ids]
        self._early_n = sum(p.numel() for p in self._params if id(p) in early_ids)
        total = sum(p.numel() for p in self._params)
        self._flat_grad = torch.zeros(total, dtype=torch.float32, device=self.device)
        self._flat_views, off = [], 0
        for p in self._params:
            n = p.numel()
            self._flat_views.append(self._flat_grad[off:off + n].view_as(p))
            p.grad = None
            off += n
        self._sink_targets = {id(p): v for p, v in zip(self._params, self._flat_views)}
        self._alloc_comm_buf()

    def _alloc_comm_buf(self):
        """The bf16 wire buffer of the all-reduce, allocated BEFORE any stream capture: allocated lazily inside the capture it
        lived in that graph's private pool while eager collectives and later graphs kept using it."""
        if self.world > 1 and self.allreduce_dtype == torch.bfloat16 and self._flat_grad is not None and (
                self._comm_buf is None or self._comm_buf.device != self._flat_grad.device
                or self._comm_buf.numel() != self._flat_grad.numel()):
            self._comm_buf = torch.empty_like(self._flat_grad, dtype=torch.bfloat16)

    def _collect_grads(self, point_grads: bool = True):
        src, dst = [], []
        self._no_grad = []  # indices of parameters without a gradient this step: the optimiser leaves them alone, as torch's does
        for i, (p, v) in enumerate(zip(self._params, self._flat_views)):
            if p.grad is None:
                v.zero_()
                self._no_grad.append(i)
            elif p.grad.data_ptr() != v.data_ptr():
                src.append(p.grad)
                dst.append(v)
            if point_grads:
                p.grad = v
        if src:
            torch._foreach_copy_(dst, src)

    def _broadcast_parameters(self):
        for t in list(self.agent.parameters()) + list(self.agent.buffers()):
            dist.broadcast(t.data, src=0, group=self._group)

    def _zero_grad(self):
        # fresh gradients every step: backward then writes them without an accumulate kernel per parameter
        self.optimizer.zero_grad(set_to_none=True)

    def _allreduce_grads(self, collective: bool = True):
        """Gather this step's gradients into the flat bucket and average it over the ranks (``collective`` False: the captured
        graph already did the second part)."""
        if self._flat_grad is not None:
            self._collect_grads()  # no-op for gradients that already live in the bucket (hipGraph replay)
        if self.world > 1 and collective:
            self._allreduce_bucket()

    def _allreduce_bucket(self):
        """mean over ranks of the flat gradient bucket, in place: ONE collective (f32, or bf16 on the wire) -- or, when the
        early part of the bucket is already under way (``_start_early_allreduce``), the rest of it plus a wait for that part."""
        lo = 0
        if self._early_work is not None:
            lo = self._early_n
        elif 0 < self._early_n < self._flat_grad.numel() and self._two_buckets_now() and (
                self.device.type != "cuda" or self._split_always):
            # nothing to hide it under at this point: on the device a replayed graph (or a backward without the sink) is followed
            # by ONE collective; the host path (gloo, the CPU tests) runs the same two-bucket protocol so that it is covered
            self._start_early_allreduce()
            lo = self._early_n
        grad = self._flat_grad[lo:] if lo else self._flat_grad
        if self.allreduce_dtype == torch.bfloat16:
            self._alloc_comm_buf()  # (no-op: _bind_flat_grads allocated it, outside any capture)
            comm = self._comm_buf[lo:] if lo else self._comm_buf
            comm.copy_(grad)
            dist.all_reduce(comm, op=dist.ReduceOp.SUM, group=self._group)
            grad.copy_(comm)
        else:
            dist.all_reduce(grad, op=dist.ReduceOp.SUM, group=self._group)
        if self._early_work is not None:
            self._early_work.wait()
            self._early_work = None
            if self.allreduce_dtype == torch.bfloat16:
                self._flat_grad[:lo].copy_(self._comm_buf[:lo])
        self._flat_grad.div_(self.world)

    def _two_buckets_now(self) -> bool:
        """Whether a collective may start at this point of the step: not while a hipGraph is being captured unless the capture
        includes the collectives (a replay runs no Python; the eager collective after it is then a single one)."""
        if self.world <= 1 or not self._early_n or not self._early_armed:
            return False
        capturing = self.device.type == "cuda" and torch.cuda.is_current_stream_capturing()
        return (not capturing) or bool(getattr(self, "_capturing_allreduce", False))

    def _start_early_allreduce(self):
        """SUM over ranks of the first ``_early_n`` elements of the bucket, asynchronously (the process group's own stream /
        thread); ``_allreduce_bucket`` waits for it.  Called from ``GradSink.early_complete`` on autograd's worker thread, right
        after the launch that summed those gradients into the bucket (the collective is ordered behind it on the device)."""
        n = self._early_n
        if self.allreduce_dtype == torch.bfloat16:
            self._alloc_comm_buf()
            self._comm_buf[:n].copy_(self._flat_grad[:n])
            self._early_work = dist.all_reduce(self._comm_buf[:n], op=dist.ReduceOp.SUM, group=self._group, async_op=True)
        else:
            self._early_work = dist.all_reduce(self._flat_grad[:n], op=dist.ReduceOp.SUM, group=self._group, async_op=True)
        self._early_launched += 1

    def _on_early_grads(self):
        if self._two_buckets_now():
            self._start_early_allreduce()

    def _global_max(self, value: int) -> int:
        t = torch.tensor([int(value)], dtype=torch.int64, device=self.device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX, group=self._group)
        return int(t.item())

    def _shard(self, batch_size: int):
        """This rank's slice of a global batch of ``batch_size`` envs."""
        if self.world == 1:
            return batch_size, 0, None
        per = batch_size // self.world
        if per * self.world != batch_size:
            raise ValueError(f"rollout batch size {batch_size} must be divisible by world size {self.world}")
        return per, self.rank * per, batch_size

    # ------------------------------------------------------------------ collect
    def collect_rollouts(self, batch_size: int, num_batches: int) -> None:
        """``num_batches`` lock-step batches of ``batch_size`` complete episodes with the current policy."""
        self.rollout_buffer.reset()
        self.agent.eval()
        act = TorchActionFunction(self.agent, use_mask=self.use_action_mask, device=self.device,
                                  amp_dtype=self.amp_dtype if self.rollout_amp else None,
                                  graph_cache=self._rollout_graphs if getattr(self.agent, "rollout_graph_ok", False) else None)
        self.batch_runner.act_fn = act
        local_b, env0, total = self._shard(batch_size)
        self.batch_runner.env0, self.batch_runner.total_envs = env0, total
        if self.world > 1:
            self.batch_runner._engine.global_max = self._global_max
        total_episodes = 0
        ep_rew, ep_len = [], []
        if self.rollout_mode == "fixed_horizon":
            with torch.no_grad():
                for _ in range(num_batches):
                    traj, last_v = self.batch_runner.collect_fixed(local_b, self.rollout_horizon)
                    self.rollout_buffer.store_fixed_trajectory(traj, last_v, self.gamma, self.lambda_gae)
                    lens = traj.finished_episode_lengths()
                    total_episodes += int(lens.numel())
                    ep_len.append(lens)
                    # per FINISHED episode, like the lengths (and like the reference's statistic): the largest single-step reward
                    # of the episode, carried across rollouts for episodes that span several of them
                    vals, self._fixed_reward_carry = traj.finished_episode_max_rewards(
                        self._fixed_reward_carry if (self._fixed_reward_carry is not None
                                                     and self._fixed_reward_carry.shape[0] == traj.B) else None)
                    ep_rew.append(vals)
            self._finish_collect(ep_rew, ep_len, total_episodes)
            return
        with torch.no_grad():
            for _ in range(num_batches):
                traj = self.batch_runner.collect(local_b)
                self.rollout_buffer.store_trajectory(traj, self.gamma, self.lambda_gae)
                total_episodes += traj.B
                # "episode reward" statistic of the reference: the largest single-step reward of the env's row
                valid = traj.valid()
                rmax = torch.where(valid, traj.rewards, torch.full_like(traj.rewards, float("-inf"))).max(dim=0).values
                frozen_zero = traj.ep_len < traj.T  # rows padded with zero-reward frozen frames
                rmax = torch.where(frozen_zero, rmax.clamp_min(0.0), rmax)
                ep_rew.append(rmax)
                ep_len.append(torch.where(traj.ep_len > 0, traj.ep_len, torch.full_like(traj.ep_len, traj.T)))
        self._finish_collect(ep_rew, ep_len, total_episodes)

    def _finish_collect(self, ep_rew, ep_len, total_episodes: int) -> None:
        """Episode statistics, the global timestep count and the rollout/* scalars of one collect_rollouts call."""
        ep_rew = torch.cat(ep_rew).cpu().numpy()
        ep_len = torch.cat(ep_len).cpu().numpy()
        self.episode_rewards.extend(ep_rew.tolist())
        self.episode_lengths.extend(ep_len.tolist())
        n_local = self.rollout_buffer.buffer_size
        if self.world > 1:
            t = torch.tensor([n_local], dtype=torch.int64, device=self.device)
            dist.all_reduce(t, group=self._group)
            n_global = int(t.item())
        else:
            n_global = n_local
        self.total_timesteps += n_global
        logger.info("Collected %d timesteps from %d episodes", n_global, total_episodes * self.world)
        if len(ep_rew):
            self.last_rollout_stats = {
                "mean_max_episode_reward": float(np.mean(ep_rew)), "max_episode_reward": float(np.max(ep_rew)),
                "mean_episode_length": float(np.mean(ep_len)) if len(ep_len) else float("nan"), "timesteps": n_global,
                "episodes": int(total_episodes) * self.world}
            for k in ("mean_max_episode_reward", "max_episode_reward", "mean_episode_length"):
                self.writer.add_scalar(f"rollout/{k}", self.last_rollout_stats[k], self.total_timesteps)

    # ------------------------------------------------------------------ loss
    def _compute_ppo_loss(self, observations, action_indices, action_masks, old_log_probs, advantages, returns):
        """-> (total loss scalar, policy_loss [M], value_loss [M], entropy_loss [M], new_log_probs [M]).

        ratio = exp(new - old); policy = -min(ratio*A, clip(ratio, 1-eps, 1+eps)*A); value = (V - R)^2 on the
        z-scored returns; entropy = -H; total = mean(policy + c_v*value + c_e*entropy)."""
        new_log_probs, values, entropy = self.agent.evaluate_actions(
            observations, action_indices, action_mask=action_masks if self.use_action_mask else None)
        ratio = torch.exp(new_log_probs - old_log_probs)
        surr1 = ratio * advantages
        surr2 = torch.clamp(ratio, 1 - self.clip_epsilon, 1 + self.clip_epsilon) * advantages
        policy_loss = -torch.min(surr1, surr2)
        value_loss = F.mse_loss(values.flatten(), returns, reduction="none")
        entropy_loss = -entropy
        loss = (policy_loss + self.value_loss_coef * value_loss + self.entropy_coef * entropy_loss).mean()
        return loss, policy_loss, value_loss, entropy_loss, new_log_probs

    def _build_graph(self, gkey, batch_size: int, sample: dict):
        """Capture the update for this minibatch layout; on failure fall back to eager mode for good (None)."""
        if self.allreduce_in_graph and self.allreduce_in_graph_fallback is None:
            try:
                self._graphs[gkey] = _GraphedFwdBwd(self, batch_size, sample, capture_allreduce=True)
                return self._graphs[gkey]
            except Exception as e:  # capture again without the collective (below): never the eager update because of this option
                logger.warning("capturing the gradient all-reduce inside the hipGraph failed (%s); capturing without it", e)
                self.allreduce_in_graph_fallback = repr(e)
                Bf16Shadow.invalidate_all()
                self.optimizer.zero_grad(set_to_none=True)
        try:
            self._graphs[gkey] = _GraphedFwdBwd(self, batch_size, sample)
        except Exception as e:  # the eager path computes the same thing, only slower: never lose a run
            logger.warning("hipGraph capture of the update failed (%s); continuing in eager mode", e)
            self.use_hip_graph = False
            self.hip_graph_fallback = repr(e)  # reported by update_policy's metrics and by bench.py
            self._graphs.clear()
            Bf16Shadow.invalidate_all()  # shadows touched inside the aborted capture were never really copied
            self.optimizer.zero_grad(set_to_none=True)
            return None
        return self._graphs[gkey]

    def _fused_loss_ok(self) -> bool:
        return self.device.type == "cuda" and getattr(self.agent, "action_dim", 4) == 4

    def _loss_backward(self, obs, actions, masks, old_lp, adv, ret, cache_enabled: bool = True, zero: bool = True):
        """Forward + loss + (scaled) backward of one minibatch -> (stats [4] f32: mean policy, value, entropy, total
        loss; kl [1] f32), both detached.  On the device the loss, its means and its gradient are one HIP launch
        (``_FusedPPOLoss``; actions/masks arrive packed, see ``_unpack_batch``), else ``_compute_ppo_loss``.
        Nothing that references the autograd graph leaves this frame: a live graph keeps the parameters'
        AccumulateGrad nodes (and the stream they were created on) alive, and a later hipGraph capture on another
        stream would then have to synchronise with that stream, which breaks the capture."""
        fused = self._fused_loss_ok()
        ctx = autocast(device_type="cuda", dtype=self.amp_dtype, cache_enabled=cache_enabled) if self.use_amp \
            else contextlib.nullcontext()
        with ctx:
            if fused:
                logits, values = self.agent(obs, None)
                scale = None
                if self.use_amp and self.scaler.is_enabled():  # the kernel applies the loss scale to its gradients
                    if self.scaler._scale is None:
                        self.scaler._lazy_init_scale_growth_tracker(self.device)
                    scale = self.scaler._scale
                sums, new_lp = _FusedPPOLoss.apply(logits, values.reshape(-1), actions,
                                                   masks if self.use_action_mask else None, old_lp, adv, ret,
                                                   self.clip_epsilon, self.value_loss_coef, self.entropy_coef, scale,
                                                   self._running_sums() if self._acc_in_kernel else None)
            else:
                loss, pl, vl, el, new_lp = self._compute_ppo_loss(obs, actions, masks, old_lp, adv, ret)
        if zero:
            self._zero_grad()
        if fused:
            out = sums if (scale is not None or not self.use_amp) else self.scaler.scale(sums)
            # weight / bias / LayerNorm gradients: first-stage partials only, summed into the bucket by ONE launch at the end
            sink = None
            if self.grad_sink and self._flat_grad is not None:
                early = [id(p) for p in self._early_params] if 0 < self._early_n < self._flat_grad.numel() else None
                sink = GradSink(self._sink_targets, early=early, on_early=self._on_early_grads if early else None)
            with grad_sink(sink):
                out.backward(self._loss_selector())
            if sink is not None and sink.written:  # autograd never saw these: point .grad at what the sink wrote
                for p, v in zip(self._params, self._flat_views):
                    if id(p) in sink.written:
                        p.grad = v
            d = sums.detach()  # f32 [5] (the kernel has added them to ``_running_sums()`` already)
            return d[:4], d[4:5]
        if self.use_amp:
            self.scaler.scale(loss).backward()
        else:
            loss.backward()
        with torch.no_grad():
            stats = torch.stack([pl.mean(), vl.mean(), el.mean(), loss.detach()]).double()
            kl = (old_lp - new_lp).mean().double().reshape(1)
        return stats, kl

    def _armed_loss_backward(self, sample: dict):
        """update_policy's eager minibatch: the early bucket's all-reduce may start from inside this backward."""
        self._early_armed = self.world > 1
        try:
            return self._loss_backward(**sample)
        except BaseException:
            self._early_armed, self._early_work = False, None
            raise

    def _running_sums(self) -> torch.Tensor:
        """f64 [5] on the device: policy, value, entropy, total loss and KL estimate summed over the minibatches of the current
        ``update_policy`` call.  ONE buffer for the trainer's lifetime (replayed graphs hold its address); the fused loss kernel adds
        to it, everything else through ``acc5 +=``."""
        if getattr(self, "_acc5", None) is None or self._acc5.device != self.device:
            self._acc5 = torch.zeros(5, dtype=torch.float64, device=self.device)
        return self._acc5

    def _loss_selector(self) -> torch.Tensor:
        """d(total)/d(sums): picks the total loss out of the fused loss kernel's five means."""
        if getattr(self, "_sel", None) is None or self._sel.device != self.device:
            self._sel = torch.tensor([0.0, 0.0, 0.0, 1.0, 0.0], device=self.device)
        return self._sel

    def _unpack_batch(self, batch, packed: bool = False):
        """-> (obs, actions, masks, old_log_probs, advantages, returns).  ``packed`` (the fused loss kernel's layout):
        actions u8 [M] and masks u8 [M] bitmasks, else actions int64 [M] and masks bool [M, 4]."""
        obs = batch["observations"]
        actions = batch["actions"]
        masks = batch["action_masks"]
        if actions.dim() > 1:  # reference layout: one-hot float actions
            actions = actions.argmax(dim=-1)
        if packed:
            actions = actions.to(torch.uint8)
            if masks.dim() > 1:
                bits = torch.tensor([1, 2, 4, 8], dtype=torch.uint8, device=masks.device)
                masks = (masks.to(torch.uint8) * bits).sum(-1).to(torch.uint8)
        else:
            actions = actions.long()
            if masks.dtype == torch.uint8 and masks.dim() == 1:  # packed bitmask -> [M, 4]
                bits = torch.tensor([1, 2, 4, 8], dtype=torch.uint8, device=masks.device)
                masks = (masks.unsqueeze(-1) & bits) != 0
        return obs, actions, masks, batch["log_probs"], batch["advantages"], batch["returns"]

    def per_rank_samples_per_epoch(self):
        """This rank's share of the reference's ONE subset of ``max_samples_per_epoch`` per epoch (None: no subset)."""
        per_rank = self.max_samples_per_epoch
        if per_rank is not None and self.world > 1:
            per_rank = -(-int(per_rank) // self.world)
        return per_rank

    # ------------------------------------------------------------------ update
    def update_policy(self, batch_size: int = 64, n_epochs: int = 4) -> Dict[str, float]:
        """``n_epochs`` passes of minibatch PPO over the rollout buffer; early stop when mean(old - new log-prob)
        of an epoch exceeds ``target_kl``."""
        if self.rollout_buffer.buffer_size == 0:
            logger.warning("No data in rollout buffer")
            return {}
        data = self.rollout_buffer.device_data(self.device)
        # the reference draws ONE subset of max_samples_per_epoch from the whole buffer (src/ppo/data_loader.py:73-101);
        # sharded, every rank draws its 1/world share of it from its own slice of the buffer, so the global number of
        # samples (and of optimiser steps) per epoch is the single-device one
        per_rank = self.per_rank_samples_per_epoch()
        dataset = PPODataset(data, gamma=self.gamma, lambda_gae=self.lambda_gae, max_samples_per_epoch=per_rank,
                             shuffle_on_reset=self.shuffle_on_reset, group=self._group)
        batches = DeviceBatches(dataset, batch_size, drop_last=True)
        n_per_epoch = len(batches)
        if self.world > 1:  # every rank must run the same number of collectives
            t = torch.tensor([n_per_epoch], dtype=torch.int64, device=self.device)
            dist.all_reduce(t, op=dist.ReduceOp.MIN, group=self._group)
            n_per_epoch = int(t.item())
        self.agent.train()
        acc5 = self._running_sums().zero_()  # sums over all minibatches: policy, value, entropy, total, kl
        kl_before = 0.0
        n_updates = 0
        mean_kl = 0.0
        packed = self._fused_loss_ok() and batches.packed()
        for epoch in range(n_epochs):
            done_batches = 0
            for idx in batches.indices():
                if done_batches >= n_per_epoch:
                    break
                full = idx.numel() == batch_size
                if packed:
                    # device-resident packed buffer: one gather launch, straight into the graph's static inputs
                    gkey = (batch_size, "packed")
                    graphed = self._graphs.get(gkey) if (self.use_hip_graph and full) else None
                    if graphed is not None:
                        batches.gather_packed(idx, out=graphed.static)
                        stats, kl = graphed.replay()
                    else:
                        sample = batches.gather_packed(idx)
                        if self.use_hip_graph and full:
                            graphed = self._build_graph(gkey, batch_size, sample)
                        stats, kl = graphed.run(sample) if graphed is not None else self._armed_loss_backward(sample)
                else:
                    obs, actions, masks, old_lp, adv, ret = self._unpack_batch(batches.gather(idx),
                                                                               packed=self._fused_loss_ok())
                    sample = dict(obs=obs, actions=actions, masks=masks, old_lp=old_lp, adv=adv, ret=ret)
                    graphed = None
                    if self.use_hip_graph and full:
                        gkey = (batch_size, obs.dtype, tuple(obs.shape[1:]))
                        graphed = self._graphs.get(gkey) or self._build_graph(gkey, batch_size, sample)
                    stats, kl = graphed.run(sample) if graphed is not None else self._armed_loss_backward(sample)
                self._early_armed = self.world > 1  # (the collective below may itself split the bucket: CPU / non-sink path)
                self._allreduce_grads(collective=not (graphed is not None and graphed.allreduce_captured))
                self._early_armed = False
                if graphed is not None and self._flat_grad is not None:
                    self._no_grad = list(graphed.no_grad)  # (see _GraphedFwdBwd: not recoverable from p.grad after a replay)
                if self._flat_step is not None:
                    self._flat_step.adopt_shadows(list(Bf16Shadow._live))  # (no-op once they are adopted)
                    self._flat_step.step(self.max_grad_norm, self.scaler if self.use_amp else None,
                                         skip=getattr(self, "_no_grad", ()))
                elif self.use_amp:
                    self.scaler.unscale_(self.optimizer)
                    torch.nn.utils.clip_grad_norm_(self.agent.parameters(), self.max_grad_norm)
                    self.scaler.step(self.optimizer)
                    self.scaler.update()
                else:
                    torch.nn.utils.clip_grad_norm_(self.agent.parameters(), self.max_grad_norm)
                    self.optimizer.step()
                if self._flat_step is None and self.device.type == "cuda":
                    # torch's fused optimisers update the parameters without advancing autograd's version counters
                    # (measured: an eager update kept running its forward on the bf16 shadows of the FIRST step); everything
                    # that caches derived weights keys on them (Bf16Shadow, the fused rollout encoder's packed weights)
                    torch.autograd.graph.increment_version(list(self.agent.parameters()))
                self.lr_scheduler.step()
                if kl._base is not None and kl._base is stats._base and kl._base.numel() == 5:
                    pass  # the fused loss (views of its one [5] tensor): g2048_ppo_loss has added them to acc5 itself
                else:
                    acc5[:4] += stats
                    acc5[4:] += kl
                n_updates += 1
                done_batches += 1
                self.total_update_steps += 1
            self.total_epochs += 1
            kl_now = acc5[4:5].clone()  # this rank's running sum; the epoch's share is the difference to the last epoch's
            if self.world > 1:
                dist.all_reduce(kl_now, group=self._group)
                kl_now /= self.world
            kl_now = float(kl_now.item())
            mean_kl = (kl_now - kl_before) / done_batches if done_batches else 0.0
            kl_before = kl_now
            if mean_kl > self.target_kl:
                logger.info("Early stopping at epoch %d due to high KL divergence: %.6f", epoch, mean_kl)
                break
        s = (acc5[:4] / max(n_updates, 1)).tolist()
        metrics = {"policy_loss": s[0] if n_updates else 0, "value_loss": s[1] if n_updates else 0,
                   "entropy_loss": s[2] if n_updates else 0, "total_loss": s[3] if n_updates else 0,
                   "kl_divergence": mean_kl, "n_updates": n_updates}
        for k, v in metrics.items():
            self.writer.add_scalar(f"train/{k}", v, self.total_timesteps)
        # how the minibatches ran: replayed hipGraph (and how many captured layouts) or the eager fallback
        metrics["hip_graph"] = bool(self.use_hip_graph and self._graphs)
        metrics["hip_graphs_captured"] = len(self._graphs)
        if self.hip_graph_fallback:
            metrics["hip_graph_fallback"] = self.hip_graph_fallback
        # the rollout forward's own capture (agents with ``rollout_graph_ok``): why it runs eagerly, if it does
        if self.rollout_graph_fallback:
            metrics["rollout_graph_fallback"] = self.rollout_graph_fallback
        if self.world > 1:
            metrics["allreduce"] = {"dtype": str(self.allreduce_dtype).replace("torch.", ""), "bytes": int(
                self._flat_grad.numel() * (2 if self.allreduce_dtype == torch.bfloat16 else 4)) if self._flat_grad is not None else 0,
                "in_graph": bool(self.allreduce_in_graph and self.allreduce_in_graph_fallback is None and self._graphs)}
            if self.allreduce_in_graph_fallback:
                metrics["allreduce"]["in_graph_fallback"] = self.allreduce_in_graph_fallback
        if self.lr_scheduler is not None:
            self.writer.add_scalar("train/lr", self.lr_scheduler.get_last_lr()[0], self.total_timesteps)
        self.writer.add_scalar("train/total_epochs", self.total_epochs, self.total_timesteps)
        self.writer.add_scalar("train/total_update_steps", self.total_update_steps, self.total_timesteps)
        if not isinstance(self.writer, _NullWriter):
            for name, p in self.agent.named_parameters():
                self.writer.add_histogram(f"train/param_magnitude/{name}", p.data, self.total_timesteps)
        return metrics

    # ------------------------------------------------------------------ checkpoints
    def save_checkpoint(self, filename: str) -> None:
        """torch.save of the reference's checkpoint dict (same keys; rank 0 only in a multi-GPU run)."""
        if self.rank != 0:
            return
        ckpt = {
            "agent_state_dict": self.agent.state_dict(), "optimizer_state_dict": self.optimizer.state_dict(),
            "total_timesteps": self.total_timesteps, "total_epochs": self.total_epochs,
            "total_update_steps": self.total_update_steps, "episode_rewards": list(self.episode_rewards),
            "episode_lengths": list(self.episode_lengths), "last_save_timestep": self.last_save_timestep,
        }
        if self.use_amp and self.scaler is not None:
            ckpt["scaler_state_dict"] = self.scaler.state_dict()
        torch.save(ckpt, filename)
        logger.info("Checkpoint saved to %s (timesteps: %d)", filename, self.total_timesteps)

    def load_checkpoint(self, filename: str, load_optimizer: bool = False) -> None:
        """Restore agent weights and counters; optimizer/scaler state only when asked / available."""
        self.load_checkpoint_path = filename
        ckpt = torch.load(filename, map_location=self.device, weights_only=False)
        missing = [k for k in ("agent_state_dict", "optimizer_state_dict") if k not in ckpt]
        if missing:
            raise ValueError(f"Checkpoint missing required keys: {missing}")
        self.agent.load_state_dict(ckpt["agent_state_dict"])
        if load_optimizer:
            try:
                self.optimizer.load_state_dict(ckpt["optimizer_state_dict"])
            except Exception as e:  # keep training with a fresh optimizer, as the reference does
                logger.warning("Failed to load optimizer state: %s", e)
            if self._flat_step is not None:
                try:
                    self._flat_step.adopt_state()
                except ValueError as e:
                    # e.g. a plain-AdamW checkpoint whose parameters disagree on the step count: like a state that failed to
                    # load at all, keep training with a fresh optimizer state (the reference does, ppo_trainer.py:570-582); the
                    # rest of the restore below still runs
                    logger.warning("Optimizer state not usable by the flat step (%s); continuing with a fresh state", e)
                    self._flat_step.reset_state()
        self.total_timesteps = ckpt.get("total_timesteps", 0)
        self.total_epochs = ckpt.get("total_epochs", 0)
        self.total_update_steps = ckpt.get("total_update_steps", 0)
        hist = self.episode_rewards.maxlen
        self.episode_rewards = _History(ckpt.get("episode_rewards", []), maxlen=hist)
        self.episode_lengths = _History(ckpt.get("episode_lengths", []), maxlen=hist)
        self.last_save_timestep = ckpt.get("last_save_timestep", 0)
        if self.use_amp and self.scaler is not None and "scaler_state_dict" in ckpt:
            try:
                self.scaler.load_state_dict(ckpt["scaler_state_dict"])
            except Exception as e:
                logger.warning("Failed to load GradScaler state: %s", e)
        self._graphs.clear()  # captured graphs point at the old gradient buffers
        if self._flat_grad is not None:
            self._bind_flat_grads()
        else:
            self.optimizer.zero_grad(set_to_none=True)
        logger.info("Checkpoint loaded: %d timesteps, %d epochs, %d update steps", self.total_timesteps,
                    self.total_epochs, self.total_update_steps)
        if self.episode_rewards:
            logger.info("  - Recent mean reward: %.2f", float(np.mean(_tail(self.episode_rewards, 100))))

    # ------------------------------------------------------------------ main loop
    def train(self, total_timesteps: int, rollout_batch_size: int = 32, rollout_batches: int = 4,
              update_epochs: int = 4, train_batch_size: int = 64, save_freq: int = 10000,
              resume_extend_steps: bool = True) -> None:
        """collect -> update until ``total_timesteps`` more (``resume_extend_steps``) or in total have been
        gathered; ``checkpoint_{iteration}.pt`` every ``save_freq`` timesteps and ``final_model.pt`` at the end."""
        start = self.total_timesteps
        target = start + total_timesteps if resume_extend_steps else total_timesteps
        verb = "Resuming" if self.load_checkpoint_path is not None else "Starting"
        logger.info("%s training from %d timesteps to reach %d", verb, start, target)
        if start >= target:
            logger.warning("Already trained for %d timesteps, target is %d. No training needed.", start, target)
            return
        iteration = 0
        while self.total_timesteps < target:
            iteration += 1
            self.collect_rollouts(rollout_batch_size, rollout_batches)
            metrics = self.update_policy(batch_size=train_batch_size, n_epochs=update_epochs)
            logger.info("Iteration %d, Timesteps: %d/%d", iteration, self.total_timesteps, target)
            if metrics:
                logger.info("Policy Loss: %.4f, Value Loss: %.4f, Entropy: %.4f", metrics["policy_loss"],
                            metrics["value_loss"], metrics["entropy_loss"])
            if self.episode_rewards:
                logger.info("Mean Episode Reward (last 100): %.2f", float(np.mean(_tail(self.episode_rewards, 100))))
            if self.total_timesteps - self.last_save_timestep >= save_freq:
                self.save_checkpoint(f"checkpoint_{iteration}.pt")
                self.last_save_timestep = self.total_timesteps
        logger.info("Training completed!")
        self.save_checkpoint("final_model.pt")
        self.writer.close()
