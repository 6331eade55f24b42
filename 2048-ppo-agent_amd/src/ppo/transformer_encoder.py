"""Transformer trunk of the policy (state-dict compatible with the reference src/ppo/transformer_encoder.py).

Parameter/buffer names are the reference's (``positional_encoding.{inv_freq,pe}``, ``cls_token``,
``encoder.layers.N.*`` of ``nn.TransformerEncoder``) so checkpoints interchange; the forward is written
out explicitly (pre-norm layers over fused QKV + SDPA) so the rollout and update paths control dtype and
can be captured in hipGraphs.  The GEMMs run on MFMA through hipBLASLt.
"""
import math

import torch
import torch.nn as nn
import torch.nn.functional as F

from ..env_definitions import BOARD_DIM


class _LinearSplitK(torch.autograd.Function):
    """``F.linear`` under bf16 autocast whose WEIGHT gradient is a split-K product.

    dW = dY^T X reduces over every token of the minibatch (34 816 at minibatch 2048) into a tiny [out, in] matrix;
    hipBLASLt runs that as 16-64 workgroups on 256 CUs (177 us per GEMM).  Cutting the token axis into 16 slices turns
    it into a batched GEMM with 16x the workgroups plus an f32 sum of the partials: 42 us, 4x faster, and the f32 sum is
    at least as accurate as the single bf16-output GEMM it replaces (tools/probe_splitk.py).  Forward, dX and the
    bias gradient are exactly what autocast does."""

    SLICES = 16

    @staticmethod
    def forward(ctx, x, weight, bias):
        with torch.autocast("cuda", enabled=False):
            xb, wb = x.to(torch.bfloat16), weight.to(torch.bfloat16)
            y = F.linear(xb, wb, None if bias is None else bias.to(torch.bfloat16))
        ctx.save_for_backward(xb, wb)
        ctx.meta = (x.dtype, weight.dtype, None if bias is None else bias.dtype)
        return y

    @staticmethod
    def backward(ctx, dy):
        xb, wb = ctx.saved_tensors
        x_dtype, w_dtype, b_dtype = ctx.meta
        with torch.autocast("cuda", enabled=False):
            dy2 = dy.reshape(-1, dy.shape[-1]).to(torch.bfloat16)
            x2 = xb.reshape(-1, xb.shape[-1])
            T, S = x2.shape[0], _LinearSplitK.SLICES
            dx = (dy2 @ wb).view(xb.shape).to(x_dtype) if ctx.needs_input_grad[0] else None
            if T % S == 0 and T // S >= 1024:
                dw = torch.bmm(dy2.view(S, T // S, -1).transpose(1, 2), x2.view(S, T // S, -1)).float().sum(0)
            else:
                dw = dy2.t() @ x2
            db = None if b_dtype is None else dy2.sum(0).to(b_dtype)
        return dx, dw.to(w_dtype), db


def _seed() -> int:
    return int(torch.randint(0, 2 ** 62, (1,)).item())  # CPU generator: no device sync


class _AttnPacked(torch.autograd.Function):
    """Self-attention over the packed in_proj output ``qkv`` [B, 17, 3*H*32] (bf16) through the HIP kernels
    ``g2048_attn_fwd/bwd``; returns [B, 17, H*32] already in the layout out_proj reads.  The gradient is written
    straight into a packed d(qkv) buffer.  Dropout acts on the attention probabilities, as nn.MultiheadAttention's."""

    @staticmethod
    def forward(ctx, qkv, nhead, p_drop):
        from ..g2048 import native as nv

        B, S, W = qkv.shape
        hw = W // 3
        qkv = qkv.contiguous()
        o = torch.empty((B, S, hw), dtype=torch.bfloat16, device=qkv.device)
        lse = torch.empty((B, nhead, S), dtype=torch.float32, device=qkv.device)
        seed = _seed() if p_drop > 0 else 0
        base = qkv.data_ptr()
        strides = (S * W, W) * 3
        nv.attn_fwd(base, base + 2 * hw, base + 4 * hw, o, lse, B, nhead, S, strides, (hw // nhead) ** -0.5, p_drop, seed)
        ctx.save_for_backward(qkv, lse)
        ctx.meta = (nhead, p_drop, seed)
        return o

    @staticmethod
    def backward(ctx, do):
        from ..g2048 import native as nv

        qkv, lse = ctx.saved_tensors
        nhead, p_drop, seed = ctx.meta
        B, S, W = qkv.shape
        hw = W // 3
        dqkv = torch.empty_like(qkv)
        base, dbase = qkv.data_ptr(), dqkv.data_ptr()
        nv.attn_bwd(base, base + 2 * hw, base + 4 * hw, do.contiguous(), lse, dbase, dbase + 2 * hw, dbase + 4 * hw, B,
                    nhead, S, (S * W, W) * 3, (hw // nhead) ** -0.5, p_drop, seed)
        return dqkv, None, None


class _AttnCls(torch.autograd.Function):
    """The CLS query of the last layer against all 17 keys: q [B, 1, H*32], kv [B, 17, 2*H*32] (bf16)."""

    @staticmethod
    def forward(ctx, q, kv, nhead, p_drop):
        from ..g2048 import native as nv

        B, S, W = kv.shape
        hw = W // 2
        q, kv = q.contiguous(), kv.contiguous()
        o = torch.empty((B, 1, hw), dtype=torch.bfloat16, device=kv.device)
        lse = torch.empty((B, nhead, 1), dtype=torch.float32, device=kv.device)
        seed = _seed() if p_drop > 0 else 0
        kb = kv.data_ptr()
        nv.attn_fwd(q.data_ptr(), kb, kb + 2 * hw, o, lse, B, nhead, 1, (hw, 0, S * W, W, S * W, W),
                    (hw // nhead) ** -0.5, p_drop, seed)
        ctx.save_for_backward(q, kv, lse)
        ctx.meta = (nhead, p_drop, seed)
        return o

    @staticmethod
    def backward(ctx, do):
        from ..g2048 import native as nv

        q, kv, lse = ctx.saved_tensors
        nhead, p_drop, seed = ctx.meta
        B, S, W = kv.shape
        hw = W // 2
        dq, dkv = torch.empty_like(q), torch.empty_like(kv)
        kb, db = kv.data_ptr(), dkv.data_ptr()
        nv.attn_bwd(q.data_ptr(), kb, kb + 2 * hw, do.contiguous(), lse, dq.data_ptr(), db, db + 2 * hw, B, nhead, 1,
                    (hw, 0, S * W, W, S * W, W), (hw // nhead) ** -0.5, p_drop, seed)
        return dq, dkv, None, None


def _fused_attention_ok(t: torch.Tensor, S: int, head_dim: int) -> bool:
    return (t.is_cuda and t.dtype == torch.bfloat16 and S == 17 and head_dim == 32 and torch.is_grad_enabled()
            and t.requires_grad)


def _linear(x: torch.Tensor, weight: torch.Tensor, bias) -> torch.Tensor:
    if (x.is_cuda and torch.is_grad_enabled() and weight.requires_grad and torch.is_autocast_enabled()
            and torch.get_autocast_dtype("cuda") == torch.bfloat16):
        return _LinearSplitK.apply(x, weight, bias)
    return F.linear(x, weight, bias)


def get_emb(sin_inp: torch.Tensor) -> torch.Tensor:
    """Interleave sin and cos of ``sin_inp`` along the last axis: [..., n] -> [..., 2n]."""
    return torch.stack((sin_inp.sin(), sin_inp.cos()), dim=-1).flatten(-2, -1)


class PositionalEncoding2D(nn.Module):
    """Fixed 2-D sinusoidal code: the first half of the channels encodes the row, the second the column.

    Buffers: ``inv_freq`` [channels/2] and ``pe`` [1, x, y, 2*channels] with channels = 2*ceil(C/4).
    """

    def __init__(self, x_size, y_size, channels, dropout=0.1, dtype_override=None):
        super().__init__()
        self.org_channels = channels
        self.dtype_override = dtype_override
        self.channels = int(math.ceil(channels / 4) * 2)
        inv_freq = 1.0 / (10000 ** (torch.arange(0, self.channels, 2).float() / self.channels))
        self.register_buffer("inv_freq", inv_freq)
        self.dropout = nn.Dropout(p=dropout)
        ex = get_emb(torch.outer(torch.arange(x_size, dtype=inv_freq.dtype), inv_freq))  # [x, channels]
        ey = get_emb(torch.outer(torch.arange(y_size, dtype=inv_freq.dtype), inv_freq))  # [y, channels]
        pe = torch.zeros((x_size, y_size, 2 * self.channels), dtype=torch.float)
        pe[:, :, : self.channels] = ex[:, None, :]
        pe[:, :, self.channels:] = ey[None, :, :]
        self.register_buffer("pe", pe.unsqueeze(0))

    def flat_table(self) -> torch.Tensor:
        """[x*y, C] table in row-major cell order."""
        return self.pe.reshape(-1, self.org_channels)

    def forward(self, tensor):  # [B, x, y, C]
        return self.dropout(tensor + self.pe[:, : tensor.shape[1], : tensor.shape[2]])

    def forward_flat(self, tensor: torch.Tensor) -> torch.Tensor:  # [B, x*y, C]
        return self.dropout(tensor + self.flat_table().unsqueeze(0))

    def forward_with_inds(self, x: torch.Tensor, inds: torch.Tensor) -> torch.Tensor:  # [B, S, C], [B, S]
        return self.dropout(x + self.flat_table()[inds])


class TransformerEncoder(nn.Module):
    """CLS token + 16 board tokens through ``num_layers`` pre-norm encoder layers; reduce to [B, d_model]."""

    def __init__(self, d_model: int, nhead: int, num_layers: int, dim_feedforward: int, dropout: float = 0.1):
        super().__init__()
        self.d_model, self.nhead, self.num_layers = d_model, nhead, num_layers
        self.dim_feedforward, self.dropout = dim_feedforward, dropout
        self.positional_encoding = PositionalEncoding2D(BOARD_DIM[0], BOARD_DIM[1], channels=d_model, dropout=dropout)
        self.cls_token = nn.Parameter(torch.randn(1, 1, d_model))
        # parameter container with the reference's names; its own forward is not used
        self.encoder = nn.TransformerEncoder(
            nn.TransformerEncoderLayer(d_model=d_model, nhead=nhead, dim_feedforward=dim_feedforward, dropout=dropout,
                                       norm_first=True, batch_first=True),
            num_layers=num_layers, enable_nested_tensor=False)

    def _layer(self, layer: nn.TransformerEncoderLayer, x: torch.Tensor, cls_only: bool = False) -> torch.Tensor:
        """One pre-norm encoder layer.  With ``cls_only`` only the CLS row of the output is produced (keys and
        values still come from every token): that is all the "cls" reduction reads from the LAST layer, and it
        skips 16/17 of that layer's query/out-projection/FFN work."""
        B, S, D = x.shape
        H = self.nhead
        p = self.dropout if self.training else 0.0
        attn = layer.self_attn
        h = F.layer_norm(x, (D,), layer.norm1.weight, layer.norm1.bias, layer.norm1.eps)
        if cls_only:
            w, b = attn.in_proj_weight, attn.in_proj_bias
            q = _linear(h[:, :1], w[:D], b[:D])
            kv = _linear(h, w[D:], b[D:])
            x = x[:, :1]
            S_out = 1
            if _fused_attention_ok(kv, S, D // H):
                a = _AttnCls.apply(q, kv, H, p)
            else:
                k, v = kv.view(B, S, 2, H, D // H).unbind(dim=2)
                a = F.scaled_dot_product_attention(q.view(B, 1, H, D // H).transpose(1, 2), k.transpose(1, 2),
                                                   v.transpose(1, 2), dropout_p=p).transpose(1, 2).reshape(B, 1, D)
        else:
            qkv = _linear(h, attn.in_proj_weight, attn.in_proj_bias)
            S_out = S
            if _fused_attention_ok(qkv, S, D // H):
                a = _AttnPacked.apply(qkv, H, p)
            else:
                q, k, v = qkv.view(B, S, 3, H, D // H).unbind(dim=2)
                a = F.scaled_dot_product_attention(q.transpose(1, 2), k.transpose(1, 2), v.transpose(1, 2),
                                                   dropout_p=p).transpose(1, 2).reshape(B, S, D)
        a = _linear(a, attn.out_proj.weight, attn.out_proj.bias)
        x = x + F.dropout(a, p, self.training)
        h = F.layer_norm(x, (D,), layer.norm2.weight, layer.norm2.bias, layer.norm2.eps)
        f = F.dropout(F.relu(_linear(h, layer.linear1.weight, layer.linear1.bias)), p, self.training)
        f = _linear(f, layer.linear2.weight, layer.linear2.bias)
        return x + F.dropout(f, p, self.training)

    def forward(self, src: torch.Tensor, reduction: str = "mean") -> torch.Tensor:
        """``src`` [B, 16, d_model] token embeddings (no positions yet) -> [B, d_model]."""
        if reduction not in ["mean", "cls"]:
            raise ValueError(f"reduction must be 'mean' or 'cls', got {reduction}")
        x = self.positional_encoding.forward_flat(src)
        x = torch.cat([self.cls_token.to(x.dtype).expand(x.shape[0], -1, -1), x], dim=1)
        last = len(self.encoder.layers) - 1
        for i, layer in enumerate(self.encoder.layers):
            x = self._layer(layer, x, cls_only=(reduction == "cls" and i == last))
        if self.encoder.norm is not None:
            x = self.encoder.norm(x)
        return x[:, 0, :] if reduction == "cls" else x[:, 1:, :].mean(dim=1)
