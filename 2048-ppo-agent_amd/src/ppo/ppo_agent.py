"""Actor-critic policy (state-dict compatible with the reference src/ppo/ppo_agent.py:11-191).

Besides the reference's one-hot float observations [B, 16, 31] the agent accepts the engine's packed
boards (uint8 [B, 16] of log2 tiles): the bias-free input Linear applied to a one-hot vector is a row
gather from its weight, so the 496-element one-hot never has to exist.
"""
from typing import Tuple

import torch
import torch.nn as nn
import torch.nn.functional as F
from torch.distributions import Categorical

from ..env_definitions import ACTION_DIM, OBS_DIM
from .hip_ops import TAIL_PARAM_ORDER, Bf16Shadow, TailBufferCache, TailPlan, _ClsTailHeads, _linear, _train_bf16
from .transformer_encoder import TransformerEncoder


_ARANGE = {}


def _one_hot(boards: torch.Tensor, classes: int, dtype) -> torch.Tensor:
    """[B, 16] integer boards -> [B, 16, classes] one-hot without the host-side range check of F.one_hot.  (The class index row is
    cached per device: built per call it is one more launch in every forward of the MLP policy, whose update is launch-bound.)"""
    key = (boards.device, boards.dtype, classes)
    ar = _ARANGE.get(key)
    if ar is None:
        ar = _ARANGE[key] = torch.arange(classes, device=boards.device, dtype=boards.dtype)
    return (boards.unsqueeze(-1) == ar).to(dtype)


def _head(d_in: int, hidden: int, d_out: int) -> nn.Sequential:
    return nn.Sequential(nn.Linear(d_in, hidden), nn.ReLU(), nn.Linear(hidden, hidden), nn.ReLU(),
                         nn.Linear(hidden, d_out, bias=False))


class _ActorCritic(nn.Module):
    """Shared action/evaluation logic over ``features(observations) -> [B, d]``."""

    # every large row reduction of the bf16 update (bias gradients) goes through g2048_colsum, so forward+backward
    # can be replayed from a hipGraph (at::sum's cross-workgroup stage does not survive a replay on this stack)
    hip_graph_safe = True

    def features(self, observations: torch.Tensor) -> torch.Tensor:
        raise NotImplementedError

    _head_shadow = None

    def _heads(self, feats: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
        """(actor(feats), critic(feats)); on the update path (bf16 autocast with gradients) the Linear layers read
        pre-cast bf16 shadows of their weights and produce f32 gradients directly (``_LinearSplitK``)."""
        lins = [m for head in (self.actor, self.critic) for m in head if isinstance(m, nn.Linear)]
        if not _train_bf16(feats, lins[0].weight):
            return self.actor(feats), self.critic(feats)
        if self._head_shadow is None:
            self._head_shadow = Bf16Shadow([p for m in lins for p in (m.weight, m.bias) if p is not None])
        views = iter(self._head_shadow())
        outs = []
        for head in (self.actor, self.critic):
            x, mods, i = feats, list(head), 0
            while i < len(mods):
                m = mods[i]
                if isinstance(m, nn.Linear):
                    relu = i + 1 < len(mods) and type(mods[i + 1]) is nn.ReLU  # fused into the Linear's node
                    x = _linear(x, m.weight, m.bias, next(views), None if m.bias is None else next(views), relu=relu)
                    i += 2 if relu else 1
                else:
                    x = m(x)
                    i += 1
            outs.append(x)
        return outs[0], outs[1]

    def forward(self, observations: torch.Tensor, action_mask: torch.Tensor = None) -> Tuple[torch.Tensor, torch.Tensor]:
        """-> (action_logits [B, action_dim], values [B, 1]); masked actions get ``logit - 1e8``."""
        logits, values = self._heads(self.features(observations))
        if action_mask is not None:
            logits = logits - 1e8 * (1 - action_mask.float())
        return logits, values

    def get_action(self, observations, action_mask=None):
        """Sample -> (actions, log_probs, values)."""
        logits, values = self.forward(observations, action_mask)
        dist = Categorical(logits=logits, validate_args=False)  # no host sync: capturable in a hipGraph
        actions = dist.sample()
        return actions, dist.log_prob(actions), values

    def evaluate_actions(self, observations, actions, action_mask=None):
        """-> (log_probs of ``actions``, values, entropy)."""
        logits, values = self.forward(observations, action_mask)
        dist = Categorical(logits=logits, validate_args=False)  # no host sync: capturable in a hipGraph
        return dist.log_prob(actions), values, dist.entropy()


class PPOAgent(_ActorCritic):
    """Transformer encoder over the 16 cells (+CLS) with 3-layer actor and critic heads."""

    def __init__(self, observation_dim: int = OBS_DIM, action_dim: int = ACTION_DIM, hidden_dim: int = 512,
                 d_model: int = 256, nhead: int = 8, num_layers: int = 4, dim_feedforward: int = 1024,
                 dropout: float = 0.1, reduction: str = "mean"):
        super().__init__()
        self.observation_dim, self.action_dim = observation_dim, action_dim
        self.hidden_dim, self.reduction = hidden_dim, reduction
        self.input_embedding = nn.Linear(observation_dim, d_model, bias=False)
        self.transformer = TransformerEncoder(d_model=d_model, nhead=nhead, num_layers=num_layers,
                                              dim_feedforward=dim_feedforward, dropout=dropout)
        self.actor = _head(d_model, hidden_dim, action_dim)
        self.critic = _head(d_model, hidden_dim, 1)

    def embed(self, observations: torch.Tensor) -> torch.Tensor:
        if observations.dtype in (torch.uint8, torch.int16, torch.int32, torch.int64):  # packed boards [B, 16]
            w = self.input_embedding.weight
            if torch.is_grad_enabled() and w.requires_grad:
                # training: expand to one-hot on the fly and use the Linear, so the weight gradient is a small
                # dense GEMM (the sort-based embedding backward sizes its launches on the host and cannot be
                # replayed from a hipGraph)
                return self.input_embedding(_one_hot(observations, self.observation_dim, w.dtype))
            return F.embedding(observations.long(), w.t())
        return self.input_embedding(observations)

    def features(self, observations):
        w = self.input_embedding.weight
        if self.transformer.embed_boards_ok(observations, w):
            return self.transformer.forward_boards(observations, w, reduction=self.reduction)
        return self.transformer(self.embed(observations), reduction=self.reduction)

    # ---- the fused CLS tail of the update path (hip_ops._ClsTailHeads): last layer's out_proj .. both heads in one node
    def _tail_heads_ok(self) -> bool:
        def std(head, n_out):
            m = list(head)
            return (len(m) == 5 and isinstance(m[0], nn.Linear) and type(m[1]) is nn.ReLU and isinstance(m[2], nn.Linear)
                    and type(m[3]) is nn.ReLU and isinstance(m[4], nn.Linear) and m[4].bias is None and m[0].bias is not None
                    and m[2].bias is not None and tuple(m[0].weight.shape) == (512, 256) and tuple(m[2].weight.shape) == (512, 512)
                    and tuple(m[4].weight.shape) == (n_out, 512))

        return (self.reduction == "cls" and self.action_dim == 4 and std(self.actor, 4) and std(self.critic, 1)
                and all(p.dtype == torch.float32 for p in self.parameters()))

    def _tail_heads(self, o, x_cls, params, dense, transposed, eps, p_drop):
        """Closure handed to the encoder (``TransformerEncoder.encode``): adds the heads' parameters and shadows to the last
        layer's and runs the node."""
        a, c = list(self.actor), list(self.critic)
        lins = [a[0], a[2], a[4], c[0], c[2], c[4]]
        if self._head_shadow is None or not self._head_shadow.packed:
            ps = [q for m in lins for q in (m.weight, m.bias) if q is not None]  # a1.w a1.b a2.w a2.b a3.w c1.w c1.b c2.w c2.b c3.w
            self._head_shadow = Bf16Shadow(ps, packed=[0, 2, 5, 7])
        v = self._head_shadow()
        pv, ptv = self._head_shadow.pviews, self._head_shadow.ptviews
        params = dict(params, a1=a[0].weight, ab1=a[0].bias, a2=a[2].weight, ab2=a[2].bias, a3=a[4].weight, c1=c[0].weight,
                      cb1=c[0].bias, c2=c[2].weight, cb2=c[2].bias, c3=c[4].weight)
        dense = dict(dense, a1=pv[0], a2=pv[2], a3=v[4], c1=pv[5], c2=pv[7], c3=v[9])
        transposed = dict(transposed, a1=ptv[0], a2=ptv[2], c1=ptv[5], c2=ptv[7])
        if not hasattr(self, "_tail_buffers"):
            self._tail_buffers = TailBufferCache()
        plan = TailPlan(params, dense, transposed, self._tail_buffers, eps, p_drop)
        return _ClsTailHeads.apply(o, x_cls, plan, *[params[k] for k in TAIL_PARAM_ORDER])

    def early_grad_parameters(self):
        """Parameters whose gradients are complete as soon as the fused CLS tail's backward has run (``TAIL_PARAM_ORDER``: the last
        layer's out_proj, norm2, linear1, linear2 and both heads) -- the trainer's early all-reduce bucket; [] when the update
        does not go through that node."""
        if not self._tail_heads_ok() or self.transformer.encoder.norm is not None:
            return []
        last = self.transformer.encoder.layers[-1]
        if last.linear1.out_features != 1024 or self.transformer.d_model != 256:
            return []
        mods = [last.self_attn.out_proj, last.norm2, last.linear1, last.linear2] + [m for h in (self.actor, self.critic)
                                                                                    for m in h if isinstance(m, nn.Linear)]
        return [p for m in mods for p in m.parameters()]

    def forward(self, observations: torch.Tensor, action_mask: torch.Tensor = None):
        w = self.input_embedding.weight
        if self.transformer.embed_boards_ok(observations, w) and self._tail_heads_ok():
            out = self.transformer.forward_boards(observations, w, reduction=self.reduction, tail_heads=self._tail_heads)
            if isinstance(out, tuple):
                logits, values = out
            else:
                logits, values = self._heads(out)
            if action_mask is not None:
                logits = logits - 1e8 * (1 - action_mask.float())
            return logits, values
        return super().forward(observations, action_mask)


class MLPAgent(_ActorCritic):
    """MLP policy for BASELINE config 2 (the reference has none): the flattened one-hot board (16 x 31) through
    a 2-layer ReLU trunk, then the same actor/critic heads and the same call interface as PPOAgent."""

    _trunk_shadow = None
    # ~15 launches of a few microseconds per lock-step: the rollout forward is replayed from a hipGraph over all boards of the
    # batch, without the live-board compaction (TorchActionFunction / RolloutEngine.rollout_policy)
    rollout_graph_ok = True

    def __init__(self, observation_dim: int = OBS_DIM, action_dim: int = ACTION_DIM, hidden_dim: int = 512,
                 trunk_dim: int = 512, board_cells: int = 16):
        super().__init__()
        self.observation_dim, self.action_dim, self.hidden_dim = observation_dim, action_dim, hidden_dim
        self.board_cells = board_cells
        self.trunk_in = nn.Linear(board_cells * observation_dim, trunk_dim)
        self.trunk_hidden = nn.Linear(trunk_dim, trunk_dim)
        self.actor = _head(trunk_dim, hidden_dim, action_dim)
        self.critic = _head(trunk_dim, hidden_dim, 1)

    def _plan(self):
        """The ``MLPPlan`` of the standard shapes (None otherwise): ONE bf16 shadow of all fourteen parameters, which both the
        ten-launch update node and the bf16 rollout forward read - two shadows over the same parameter cannot both be kept
        current by the optimiser kernel (``FlatAdamWStep.adopt_shadows``)."""
        from .mlp_ops import MLPPlan

        if self._mlp_plan is None and MLPPlan.supports(self):
            self._mlp_plan = MLPPlan(self)
        return self._mlp_plan

    def _shadows(self):
        """(trunk views, head views) of the bf16 weight shadows (created on first use; the optimiser kernel keeps them current
        once it has adopted them, otherwise they re-copy when a parameter's version moved)."""
        plan = self._plan() if next(self.parameters()).is_cuda else None
        if plan is not None:
            d = plan.views()[0]
            return ([d[k] for k in ("win", "bin", "wh", "bh")],
                    [d[k] for k in ("a1w", "a1b", "a2w", "a2b", "a3w", "c1w", "c1b", "c2w", "c2b", "c3w")])
        if self._trunk_shadow is None:
            self._trunk_shadow = Bf16Shadow([self.trunk_in.weight, self.trunk_in.bias, self.trunk_hidden.weight,
                                             self.trunk_hidden.bias])
        if self._head_shadow is None:
            lins = [m for head in (self.actor, self.critic) for m in head if isinstance(m, nn.Linear)]
            self._head_shadow = Bf16Shadow([p for m in lins for p in (m.weight, m.bias) if p is not None])
        return self._trunk_shadow(), self._head_shadow()

    def prepare_rollout(self):
        """Called eagerly before a captured rollout forward is replayed: a replay runs no Python, so shadows that nobody
        maintains (PyTorch optimiser instead of g2048_opt_step, a loaded checkpoint) are refreshed here."""
        if next(self.parameters()).is_cuda:
            self._shadows()

    def _rollout_bf16_ok(self, observations) -> bool:
        return (observations.is_cuda and observations.dtype == torch.uint8 and observations.dim() == 2
                and not torch.is_grad_enabled() and torch.is_autocast_enabled()
                and torch.get_autocast_dtype("cuda") == torch.bfloat16 and self.trunk_in.weight.dtype == torch.float32)

    _mlp_plan = None

    def _update_node_ok(self, observations) -> bool:
        """The update path on packed boards at the standard shapes: the whole policy is one autograd node over ten launches
        (``mlp_ops._MLPUpdate``; G2048_MLP_FUSED=0 keeps the per-layer nodes, the A/B switch)."""
        import os

        from .mlp_ops import MLPPlan

        return (observations.dtype == torch.uint8 and observations.dim() == 2 and observations.shape[1] == 16
                and _train_bf16(observations, self.trunk_in.weight) and MLPPlan.supports(self)
                and os.environ.get("G2048_MLP_FUSED", "1").strip().lower() not in ("0", "false", "no", "off"))

    def forward(self, observations: torch.Tensor, action_mask: torch.Tensor = None):
        if self._update_node_ok(observations):
            from .mlp_ops import MLPPlan, _MLPUpdate

            plan = self._plan()
            logits, values = _MLPUpdate.apply(observations, plan, *plan.params)
            if action_mask is not None:
                logits = logits - 1e8 * (1 - action_mask.float())
            return logits, values
        if not self._rollout_bf16_ok(observations):
            return super().forward(observations, action_mask)
        # bf16 rollout inference on packed boards: one-hot GEMM + bias/ReLU epilogues on the pre-cast shadows, 12 launches per
        # lock-step instead of 42 (the f32 gather-sum of the generic path alone is 130 us at 4 096 boards); same arithmetic
        # as the update's autocast forward (bf16 operands, f32 accumulation, bf16 activations)
        plan = self._plan()
        if plan is not None and observations.shape[1] == 16:
            # the standard shapes: the five launches of the update node's forward (trunk_in as a gather-sum on the packed boards, job
            # tables for the layers); while being captured the shadows are read as they are (see the other branch)
            from .mlp_ops import forward_nograd

            logits, values = forward_nograd(observations, plan, refresh=not torch.cuda.is_current_stream_capturing())
            if action_mask is not None:
                logits = logits - 1e8 * (1 - action_mask.float())
            return logits, values
        ts, hs = self._trunk_shadow, self._head_shadow
        if torch.cuda.is_current_stream_capturing() and ts is not None and hs is not None and ts.views is not None \
                and hs.views is not None:
            # being captured (TorchActionFunction): whoever replays the graph calls prepare_rollout() first, so the refresh
            # must not be recorded into the graph
            trunk, heads = ts.views, hs.views
        else:
            trunk, heads = self._shadows()
        with torch.autocast("cuda", enabled=False):
            h = _one_hot(observations, self.observation_dim, torch.bfloat16).flatten(1)
            h = torch._addmm_activation(trunk[1], h, trunk[0].t())
            h = torch._addmm_activation(trunk[3], h, trunk[2].t())
            views, outs = iter(heads), []
            for head in (self.actor, self.critic):
                x, mods = h, list(head)
                for i, m in enumerate(mods):
                    if not isinstance(m, nn.Linear):
                        continue
                    w = next(views)
                    b = None if m.bias is None else next(views)
                    if i + 1 < len(mods) and type(mods[i + 1]) is nn.ReLU and b is not None:
                        x = torch._addmm_activation(b, x, w.t())
                    else:
                        x = F.linear(x, w, b)
                        if i + 1 < len(mods) and type(mods[i + 1]) is nn.ReLU:
                            x = F.relu(x)
                outs.append(x)
        logits, values = outs
        if action_mask is not None:
            logits = logits - 1e8 * (1 - action_mask.float())
        return logits, values

    def features(self, observations):
        if observations.dtype in (torch.uint8, torch.int16, torch.int32, torch.int64):
            if torch.is_grad_enabled() and self.trunk_in.weight.requires_grad:
                ps = (self.trunk_in.weight, self.trunk_in.bias, self.trunk_hidden.weight, self.trunk_hidden.bias)
                sh = (None,) * 4
                bf16_path = _train_bf16(observations, ps[0])
                # (on the bf16 update path the one-hot is built in bf16 at once: exact, and one cast launch less per minibatch)
                oh = _one_hot(observations, self.observation_dim, torch.bfloat16 if bf16_path else ps[0].dtype).flatten(1)
                if bf16_path:  # pre-cast bf16 shadows (kept current by the optimiser kernel), Linear+ReLU as one node
                    if self._trunk_shadow is None:
                        self._trunk_shadow = Bf16Shadow(list(ps))
                    sh = self._trunk_shadow()
                h = _linear(oh, ps[0], ps[1], sh[0], sh[1], relu=True)
                return _linear(h, ps[2], ps[3], sh[2], sh[3], relu=True)
            # inference: one-hot @ W^T == sum over cells of the selected weight columns
            cols = observations.long() + torch.arange(self.board_cells, device=observations.device) * self.observation_dim
            h = F.embedding(cols, self.trunk_in.weight.t()).sum(dim=1) + self.trunk_in.bias
        else:
            h = self.trunk_in(observations.flatten(1))
        return F.relu(self.trunk_hidden(F.relu(h)))
