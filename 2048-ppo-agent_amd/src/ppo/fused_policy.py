"""Rollout-time policy forward through the fused MFMA encoder kernel (csrc/g2048_policy.hip).

Usable when the agent is a PPOAgent of the reference's default shape (d_model 256, 8 heads, feed-forward 1024,
"cls" reduction) on a HIP device and the rollout is asked to run in bf16.  The encoder (embedding -> 17-token
Transformer -> CLS feature) is one kernel; the two 3-layer heads stay in PyTorch (bf16 autocast) on the [B, 256]
features.  Same numerics class as torch.autocast(bf16): bf16 GEMM inputs, f32 accumulation/residual/statistics.
"""
from __future__ import annotations

import os

import torch

from ..g2048 import native as nv


def supports(agent) -> bool:
    from .ppo_agent import PPOAgent

    if not isinstance(agent, PPOAgent) or agent.reduction != "cls":
        return False
    t = agent.transformer
    p = next(agent.parameters())
    return (t.d_model == 256 and t.nhead == 8 and t.dim_feedforward == 1024 and agent.observation_dim == 31
            and t.encoder.norm is None and p.is_cuda
            and all(abs(l.norm1.eps - 1e-5) < 1e-12 and abs(l.norm2.eps - 1e-5) < 1e-12 for l in t.encoder.layers))


class FusedPolicy:
    """Packs the agent's encoder weights for ``g2048_policy_encoder`` (re-pack with ``refresh()`` after an update)."""

    def __init__(self, agent):
        if not supports(agent):
            raise ValueError("agent shape not supported by the fused encoder kernel")
        self.agent = agent
        self._key = None
        self.refresh()

    def _params_key(self):
        """Changes whenever a parameter was updated in place (optimizer step, load_state_dict) or re-allocated."""
        ps = list(self.agent.parameters())
        return (ps[0].data_ptr(), len(ps), sum(p._version for p in ps))

    def refresh_if_stale(self):
        """Re-pack when the agent's parameters changed since the last pack: an act_fn kept across optimizer steps
        (an evaluation callback, user code that reuses one TorchActionFunction as the reference does) must not roll
        out with the weights of the moment it was built."""
        if self._params_key() != self._key:
            self.refresh()

    @torch.no_grad()
    def refresh(self):
        self._key = self._params_key()
        a, t = self.agent, self.agent.transformer
        emb = a.input_embedding.weight.t().float()  # [31, 256]
        pe = t.positional_encoding.flat_table().float()  # [16, 256]
        self.table = (pe[:, None, :] + emb[None, :, :]).contiguous()  # [16, 31, 256]
        self.cls = t.cls_token.detach().float().reshape(256).contiguous()
        w, p = [], []
        # the kernel feeds accumulator tiles straight back in as MFMA operands; the hardware layout then walks the
        # reduction index of every group of 16 in the order KPERM, so the matrices multiplied against such operands
        # (in_proj, out_proj, linear1, linear2) are stored with their input columns in that order
        kperm = torch.tensor([0, 1, 2, 3, 8, 9, 10, 11, 4, 5, 6, 7, 12, 13, 14, 15], device=self.cls.device)

        def permuted(m):
            k = m.shape[1]
            idx = (torch.arange(0, k, 16, device=m.device)[:, None] + kperm[None, :]).reshape(-1)
            return m[:, idx]

        D = 256
        one, zero = torch.ones(D, device=self.cls.device), torch.zeros(D, device=self.cls.device)
        for l in t.encoder.layers:
            # parameter folding in f32 (include/g2048.h, g2048_policy_encoder): LayerNorm's affine goes into the Linear
            # behind it, the key bias is dropped (softmax is invariant to it), the value bias moves into out_proj's
            f = lambda x: x.detach().float()
            g1, be1, g2, be2 = f(l.norm1.weight), f(l.norm1.bias), f(l.norm2.weight), f(l.norm2.bias)
            wqkv, bqkv = f(l.self_attn.in_proj_weight), f(l.self_attn.in_proj_bias)
            wo, bo = f(l.self_attn.out_proj.weight), f(l.self_attn.out_proj.bias)
            w1, b1 = f(l.linear1.weight), f(l.linear1.bias)
            bqkv = bqkv + wqkv @ be1
            wqkv = wqkv * g1[None, :]
            bo = bo + wo @ bqkv[2 * D:]
            bqkv = torch.cat([bqkv[:D], zero, zero])
            b1 = b1 + w1 @ be2
            w1 = w1 * g2[None, :]
            w += [permuted(wqkv), permuted(wo), permuted(w1), permuted(f(l.linear2.weight))]
            p += [one, zero, bqkv, bo, one, zero, b1, f(l.linear2.bias)]
        self.weights = torch.cat([x.detach().reshape(-1).to(torch.bfloat16) for x in w]).contiguous()
        self.params = torch.cat([x.detach().reshape(-1).float() for x in p]).contiguous()
        # actor / critic heads: bf16 copies made once per refresh (autocast would re-cast all of them at every lock-step)
        self.heads = [[(m.weight.detach().to(torch.bfloat16), None if m.bias is None else m.bias.detach().to(torch.bfloat16))
                       for m in head if isinstance(m, torch.nn.Linear)] for head in (a.actor, a.critic)]
        # both heads read the same features: their first layers as ONE GEMM (weights stacked), ReLU in the GEMM epilogue
        (wa, ba), (wc, bc) = self.heads[0][0], self.heads[1][0]
        self.head1 = None
        if ba is not None and bc is not None and wa.shape == wc.shape:
            self.head1 = (torch.cat([wa, wc]).t().contiguous(), torch.cat([ba, bc]), wa.shape[0])
        # small live counts (the long tail of a lock-step rollout): the heads on the MLP policy's kernels (csrc/g2048_mlp.hip) - both
        # heads' first layers as one job, the second layers as two jobs of one launch, the 4 + 1 output rows straight to f32: 3 launches
        # instead of 5 GEMMs + 2 casts, each of them a ~4.5 us node of a lock-step that is launch-bound at these sizes
        self.own = None
        hs = self.heads
        if (all(len(h) == 3 for h in hs) and all(b is not None for h in hs for (_, b) in h[:2]) and all(h[2][1] is None for h in hs)
                and tuple(hs[0][0][0].shape) == (512, 256) and tuple(hs[1][0][0].shape) == (512, 256)
                and all(tuple(h[1][0].shape) == (512, 512) for h in hs) and tuple(hs[0][2][0].shape) == (4, 512)
                and tuple(hs[1][2][0].shape) == (1, 512)):
            f32 = lambda m: m.bias.detach().float().contiguous()
            la, lc = [m for m in a.actor if isinstance(m, torch.nn.Linear)], [m for m in a.critic if isinstance(m, torch.nn.Linear)]
            self.own = dict(w1=torch.cat([hs[0][0][0], hs[1][0][0]]).contiguous(), b1=torch.cat([f32(la[0]), f32(lc[0])]).contiguous(),
                            a2=hs[0][1][0].contiguous(), c2=hs[1][1][0].contiguous(), ba2=f32(la[1]), bc2=f32(lc[1]),
                            w3=torch.cat([hs[0][2][0], hs[1][2][0]]).contiguous())
        self.n_layers = len(t.encoder.layers)

    # from this many boards on, the last layer runs CLS-only in a second kernel (g2048_policy_encoder with a workspace):
    # below it the second kernel's fixed ~0.1 ms outweighs what the first one saves
    SPLIT_MIN_BOARDS = 4096
    # G2048_OWN_HEADS_MAX=<boards>: up to this many boards the heads run on g2048_gemm_jobs / g2048_mlp_out_fwd (64 x 64 tiles: above
    # ~32 768 hipBLASLt's 256 x 256 tiles move fewer operand bytes per output; measured collect of 65 536 boards with the threshold at
    # 0 / 8 192 / 16 384 / 32 768 / 65 536: 0.865 / 0.850 / 0.847 / 0.850 / 0.862 s on one box; frozen-policy bench 3.751 -> 3.776 M).
    # OFF by default (0): the heads' summation order is part of the rollout's numerics, and every learning measurement of round 4 (the
    # seed tables, the config-5 runs, the bench line's learning trajectory) was made with the library heads - see NOTES, round-4 appendix
    OWN_HEADS_MAX_BOARDS = int(os.environ.get("G2048_OWN_HEADS_MAX", "0"))

    def _workspace(self, B: int, device) -> torch.Tensor:
        need = nv.policy_encoder_workspace_bytes(B)
        ws = getattr(self, "_ws", None)
        if ws is None or ws.numel() < need or ws.device != device:
            ws = self._ws = torch.empty(need, dtype=torch.uint8, device=device)
        return ws

    @torch.no_grad()
    def features(self, boards: torch.Tensor, split=None) -> torch.Tensor:
        """boards u8 [B, 16] -> CLS features f32 [B, 256].  ``split``: force (True) / forbid (False) the two-kernel form."""
        boards = boards.contiguous()
        B = boards.shape[0]
        out = torch.empty((B, 256), dtype=torch.float32, device=boards.device)
        if split is None:
            split = B >= self.SPLIT_MIN_BOARDS
        ws = self._workspace(B, boards.device) if split else None
        nv.policy_encoder(boards, self.table, self.cls, self.weights, self.params, self.n_layers, out, ws)
        return out

    @torch.no_grad()
    def __call__(self, boards: torch.Tensor):
        """boards u8 [B, 16] -> (logits f32 [B, 4] (unmasked), values f32 [B])."""
        self.refresh_if_stale()
        feats = self.features(boards).to(torch.bfloat16)
        B = feats.shape[0]
        if self.own is not None and B <= self.OWN_HEADS_MAX_BOARDS:
            o, bf = self.own, torch.bfloat16
            h1 = torch.empty((B, 1024), dtype=bf, device=feats.device)
            h2 = torch.empty((B, 1024), dtype=bf, device=feats.device)
            logits = torch.empty((B, 4), dtype=torch.float32, device=feats.device)
            values = torch.empty(B, dtype=torch.float32, device=feats.device)
            nv.gemm_jobs([dict(segs=[(feats, o["w1"])], bias=o["b1"], relu=True, y=h1)], B)
            nv.gemm_jobs([dict(segs=[(h1[:, :512], o["a2"])], bias=o["ba2"], relu=True, y=h2[:, :512]),
                          dict(segs=[(h1[:, 512:], o["c2"])], bias=o["bc2"], relu=True, y=h2[:, 512:])], B)
            nv.mlp_out_fwd(h2, o["w3"], logits, values)
            return logits, values
        outs = []
        h1 = None
        if self.head1 is not None:
            w1t, b1, n1 = self.head1
            h1 = torch._addmm_activation(b1, feats, w1t)  # relu(feats @ W^T + b), both heads, one launch
        for k, layers in enumerate(self.heads):  # Linear-ReLU-Linear-ReLU-Linear in bf16, what autocast computes
            x = feats if h1 is None else h1[:, k * n1:(k + 1) * n1]
            for i, (w, b) in enumerate(layers):
                if i == 0 and h1 is not None:
                    continue
                if i + 1 < len(layers) and b is not None:
                    x = torch._addmm_activation(b, x, w.t())  # ReLU in the GEMM epilogue
                else:
                    x = torch.nn.functional.linear(x, w, b)
                    if i + 1 < len(layers):
                        x = torch.relu_(x)
            outs.append(x)
        return outs[0].float(), outs[1].float().reshape(-1)
