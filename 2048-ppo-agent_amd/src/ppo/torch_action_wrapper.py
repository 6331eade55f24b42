"""Policy -> act_fn bridge (API of the reference src/ppo/torch_action_wrapper.py:10-104).

The reference converts the torch agent to JAX (torch2jax) so it can be vmapped next to the Pgx env.  Here the
env lives on the GPU, so the agent simply runs batched in PyTorch-ROCm on the packed boards and the tail of
``__call__`` (clamp, categorical draw from the JAX-compatible key stream, log-softmax pick) is fused with the
env step in ``g2048_policy_step`` / available stand-alone as ``g2048_act_logits``.
"""
from __future__ import annotations

from typing import Optional

import numpy as np
import torch

from ..actions import _common as C
from ..env_definitions import BOARD_FLAT_DIM, OBS_DIM
from ..g2048 import native as nv
from .capture import capture as capture_graph


class TorchActionFunction:
    """Wrap an actor-critic ``agent`` as an ``act_fn`` plug-in for BatchRunner.

    Parameters mirror the reference: ``agent``, ``use_mask`` (apply the legal-action mask to the logits),
    ``sample_actions`` (categorical sample vs argmax), ``device`` (where the agent runs).  Extra:
    ``amp_dtype`` runs the rollout forward under autocast (the reference rolls out in fp32) -- for bfloat16 and a
    PPOAgent of the reference's default shape the encoder then runs in the fused MFMA kernel unless
    ``use_fused=False``; ``sync_every`` is how many lock-steps are enqueued between polls of the device-side
    live-env counter.
    Side effect as in the reference: ``agent`` is moved to ``device`` and put in eval mode.
    """

    def __init__(self, agent, use_mask: bool = False, sample_actions: bool = True,
                 device: torch.device = torch.device("cpu"), amp_dtype: Optional[torch.dtype] = None,
                 sync_every: int = 8, rng_mode=None, use_fused: Optional[bool] = None, graph_cache: Optional[dict] = None):
        self.agent = agent.to(device).eval()
        self.use_mask = use_mask
        self.sample_actions = sample_actions
        self.device = device
        self.amp_dtype = amp_dtype
        self.sync_every = sync_every
        self.rng_mode = rng_mode
        # bf16 rollouts of a default-shape PPOAgent go through the fused MFMA encoder kernel (csrc/g2048_policy.hip)
        self._fused = None
        if amp_dtype == torch.bfloat16 and use_fused is not False:
            from . import fused_policy

            if fused_policy.supports(self.agent):
                self._fused = fused_policy.FusedPolicy(self.agent)
        # a cheap policy (the MLP of BASELINE configs[1]: ~15 launches of microseconds per lock-step) is launch-bound in eager
        # mode: with a ``graph_cache`` (owned by the caller, it outlives this object) the forward over ALL boards of the batch is
        # replayed from a hipGraph and the engine skips the live-board compaction (``compact``), whose gathers cost more than
        # the forward they would save
        self._graph_cache = graph_cache if (self._fused is None and torch.device(device).type == "cuda") else None
        self.compact = self._graph_cache is None
        self._agent_params = dict(self.agent.named_parameters())
        self._agent_buffers = dict(self.agent.named_buffers())
        self._agent_state = {**self._agent_params, **self._agent_buffers}

    # batched device path used by the rollout engine: raw actor logits (masking happens in the kernel)
    @torch.no_grad()
    def policy_fn(self, boards: torch.Tensor, masks: torch.Tensor):
        """boards u8 [B, 16], masks u8 [B] or None (unused here) -> (logits f32 [B, 4], values f32 [B])."""
        agent_dev = next(self.agent.parameters()).device
        if self._fused is not None and boards.device == agent_dev:
            return self._fused(boards)
        if self._graph_cache is not None and boards.device == agent_dev:
            out = self._graphed(boards)
            if out is not None:
                return out
        return self._forward(boards, agent_dev)

    policy_fn.needs_masks = False  # (RolloutEngine.rollout_policy: no per-lock-step gather of the masks for this policy)

    def _forward(self, boards, agent_dev):
        x = boards if boards.device == agent_dev else boards.to(agent_dev)
        if self.amp_dtype is not None and agent_dev.type == "cuda":
            with torch.autocast(device_type="cuda", dtype=self.amp_dtype):
                logits, values = self.agent(x, None)
        else:
            logits, values = self.agent(x, None)
        return logits.float().to(boards.device), values.float().reshape(-1).to(boards.device)

    def _graphed(self, boards: torch.Tensor):
        """The forward replayed from a hipGraph (captured once per batch shape; the parameters are read in place, so later
        optimiser steps are seen).  None: capture is not possible for this agent (remembered in the cache), run eagerly."""
        prepare = getattr(self.agent, "prepare_rollout", None)
        key = (tuple(boards.shape), boards.dtype, self.amp_dtype, next(self.agent.parameters()).data_ptr())
        entry = self._graph_cache.get(key, False)
        if entry is False:
            entry = None
            try:
                if prepare is not None:
                    prepare()  # (before the capture: the refresh must not become part of the graph)
                static_in = torch.empty_like(boards)
                static_in.copy_(boards)
                side = torch.cuda.Stream(device=boards.device)
                side.wait_stream(torch.cuda.current_stream(boards.device))
                with torch.cuda.stream(side):
                    for _ in range(2):
                        self._forward(static_in, boards.device)
                torch.cuda.current_stream(boards.device).wait_stream(side)
                graph = torch.cuda.CUDAGraph()
                with capture_graph(graph):  # thread-local error mode + no garbage collection while capturing (capture.py)
                    out = self._forward(static_in, boards.device)
                entry = (graph, static_in, out)
            except Exception as e:  # not capturable: eager from now on (the reason stays visible in the cache)
                self._graph_cache["fallback"] = repr(e)
            self._graph_cache[key] = entry
        if entry is None:
            return None
        graph, static_in, out = entry
        if prepare is not None:
            prepare()
        static_in.copy_(boards)
        graph.replay()
        return out

    @torch.no_grad()
    def __call__(self, rng_key, obs, mask):
        """Un-batched plug-in protocol: ``(rng_key[2], obs[4,4,31], mask[4]) -> (action, log_prob, value)``.
        Leading batch dimensions are accepted.  The draw and the log-prob come from ``g2048_act_logits``."""
        obs_t = torch.as_tensor(np.asarray(obs.cpu() if isinstance(obs, torch.Tensor) else obs))
        batched = obs_t.ndim > 3
        obs_t = obs_t.reshape(-1, BOARD_FLAT_DIM, OBS_DIM).float()
        agent_dev = next(self.agent.parameters()).device
        logits, values = self.agent(obs_t.to(agent_dev), None)
        dev = C.device()
        bits = C.mask_to_bits(mask)
        keys = C.keys_tensor(rng_key)
        n = bits.numel()
        actions = torch.empty(n, dtype=torch.int32, device=dev)
        logp = torch.empty(n, dtype=torch.float32, device=dev)
        mode = C.default_rng_mode() if self.rng_mode is None else self.rng_mode
        nv.act_logits(keys, logits.float().to(dev).contiguous(), bits, self.use_mask, self.sample_actions, actions,
                      logp, mode)
        a, lp, v = actions.cpu().numpy(), logp.cpu().numpy(), values.float().reshape(-1).cpu().numpy()
        if batched:
            return a, lp, v
        return np.int32(a[0]), np.float32(lp[0]), np.float32(v[0])
