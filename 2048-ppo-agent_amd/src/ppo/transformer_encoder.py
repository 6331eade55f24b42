"""Transformer trunk of the policy (state-dict compatible with the reference src/ppo/transformer_encoder.py).

Parameter/buffer names are the reference's (``positional_encoding.{inv_freq,pe}``, ``cls_token``,
``encoder.layers.N.*`` of ``nn.TransformerEncoder``) so checkpoints interchange; the forward is written
out explicitly (pre-norm layers over fused QKV) so that the update path can run through the HIP kernels bound in
``hip_ops.py`` (attention, add + dropout + LayerNorm, Linear blocks, token embedding) and be captured in a hipGraph;
without a HIP device, gradients or bf16 autocast every step falls back to the PyTorch operator it mirrors.
"""
import math
import os

import torch
import torch.nn as nn
import torch.nn.functional as F

from ..env_definitions import BOARD_DIM
from .hip_ops import (Bf16Shadow, ClsLink, FFNLink, HLink, LNPre, _AddLayerNorm, _ClsRows, _AttnCls, _AttnPacked, _EmbedBoards, _ExpandRows, _InProjCls,  # noqa: F401
                      _LinearAddCast, _LinearAddLayerNorm, _LinearReluDropout, _LinearSplitK, _add_norm, _fused_attention_ok, _fused_norm_ok,
                      _linear, _train_bf16, graph_seed_state)



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
        self._shadow = None  # Bf16Shadow of the layers' Linear parameters, built on first use by the update path
        # parameter container with the reference's names; its own forward is not used
        self.encoder = nn.TransformerEncoder(
            nn.TransformerEncoderLayer(d_model=d_model, nhead=nhead, dim_feedforward=dim_feedforward, dropout=dropout,
                                       norm_first=True, batch_first=True),
            num_layers=num_layers, enable_nested_tensor=False)

    def _bf16_weights(self, like: torch.Tensor):
        """Per layer [in_proj w, b, out_proj w, b, linear1 w, b, linear2 w, b] as bf16 shadows on the update path
        (``Bf16Shadow``), else a list of None (``_linear`` then uses the masters directly)."""
        layers = self.encoder.layers
        if not _train_bf16(like, layers[0].linear1.weight):
            return [[None] * 14 for _ in layers]
        if self._shadow is None:
            ps = []
            for l in layers:
                ps += [l.self_attn.in_proj_weight, l.self_attn.in_proj_bias, l.self_attn.out_proj.weight,
                       l.self_attn.out_proj.bias, l.linear1.weight, l.linear1.bias, l.linear2.weight, l.linear2.bias]
            last = len(layers) - 1
            # [in, out] copies: linear2 of every layer (masked input-gradient GEMM); fragment-packed copies (tensor and transpose) of
            # out_proj, linear1 and linear2 of the LAST layer (the fused CLS tail, g2048_cls_tail_fwd / _bwd)
            # (+ out_proj^T of the layers in front of it: the input gradient of out_proj through g2048_linear_bf16)
            # (+ for the layers in front of the last, at d_model 256: fragment-packed out_proj and linear2 - the forward operands of
            # g2048_linear_add_ln_fwd - and fragment-packed TRANSPOSES of in_proj and linear1 - the operands of the input-gradient GEMM
            # fused with the LayerNorm backward, g2048_linear_add_ln_bwd)
            fuse = range(last) if self.d_model == 256 else range(0)
            self._shadow = Bf16Shadow(ps, transposed=[8 * i + 6 for i in range(len(layers))] + [8 * i + 2 for i in range(last)],
                                      packed=[8 * last + 2, 8 * last + 4, 8 * last + 6],
                                      packed_only=[8 * i + o for i in fuse for o in (2, 6)],
                                      packed_t_only=[8 * i + o for i in fuse for o in (0, 4)] + ([8 * last] if self.d_model == 256 and last >= 1 else []))
        v = self._shadow()
        tv, pv, ptv = self._shadow.tviews, self._shadow.pviews, self._shadow.ptviews
        full = range(len(layers) - 1)  # (the last layer's packed copies belong to the fused CLS tail)
        return [v[8 * i:8 * i + 8] + [tv[8 * i + 6], tv.get(8 * i + 2)]
                + ([pv.get(8 * i + 2), pv.get(8 * i + 6), ptv.get(8 * i + 0), ptv.get(8 * i + 4)] if i in full
                   else [None, None, ptv.get(8 * i + 0), None])  # (last layer: in_proj^T for the CLS-only K/V input gradient)
                for i in range(len(layers))]

    def _layer(self, layer: nn.TransformerEncoderLayer, x: torch.Tensor, h: torch.Tensor, next_norm,
               cls_only: bool = False, sh=(None,) * 14, cls_link_out=None, cls_link_in=None, tail_heads=None, h_link_in=None,
               h_link_out=None):
        """One pre-norm encoder layer.  ``h`` = norm1(x), already computed (by the previous layer's tail); returns
        (x_out, next_norm(x_out)) so that every residual add + dropout + LayerNorm is one fused kernel
        (``_add_norm``); ``next_norm`` None: (x_out, None).  With ``cls_only`` only the CLS row of the output is
        produced (keys and values still come from every token): that is all the "cls" reduction reads from the LAST
        layer, and it skips 16/17 of that layer's query/out-projection/FFN work."""
        B, S, D = x.shape
        H = self.nhead
        p = self.dropout if self.training else 0.0
        attn = layer.self_attn
        if cls_only:
            w, b = attn.in_proj_weight, attn.in_proj_bias
            if sh[0] is not None and h.dtype == torch.bfloat16:
                q, kv = _InProjCls.apply(h, w, b, sh[0], sh[1], h_link_in, sh[12])
            else:
                q = _linear(h[:, :1], w[:D], b[:D])
                kv = _linear(h, w[D:], b[D:])
            x = _ClsRows.apply(x, cls_link_in) if cls_link_in is not None else x[:, :1]
            if _fused_attention_ok(kv, S, D // H):
                a = _AttnCls.apply(q, kv, H, p)
            else:
                k, v = kv.view(B, S, 2, H, D // H).unbind(dim=2)
                a = F.scaled_dot_product_attention(q.view(B, 1, H, D // H).transpose(1, 2), k.transpose(1, 2),
                                                   v.transpose(1, 2), dropout_p=p).transpose(1, 2).reshape(B, 1, D)
        else:
            # (h_link_in: the node that produced h runs this Linear's input-gradient GEMM inside its own LayerNorm backward)
            qkv = _linear(h, attn.in_proj_weight, attn.in_proj_bias, sh[0], sh[1], h_link=h_link_in, wt_packed=sh[12])
            if _fused_attention_ok(qkv, S, D // H):
                a = _AttnPacked.apply(qkv, H, p)
            else:
                q, k, v = qkv.view(B, S, 3, H, D // H).unbind(dim=2)
                a = F.scaled_dot_product_attention(q.transpose(1, 2), k.transpose(1, 2), v.transpose(1, 2),
                                                   dropout_p=p).transpose(1, 2).reshape(B, S, D)
        if cls_only and tail_heads is not None and sh[0] is not None and a.dtype == torch.bfloat16 and _fused_norm_ok(x, None) \
                and layer.linear1.out_features == 1024 and D == 256:
            # update path, CLS-only last layer: everything from out_proj to the heads' outputs is ONE node (g2048_cls_tail_fwd/bwd);
            # the caller's closure adds the heads' parameters and returns (logits, values)
            i_last = len(self.encoder.layers) - 1
            pv, ptv = self._shadow.pviews, self._shadow.ptviews
            n2 = layer.norm2
            return tail_heads(a, x, dict(wo=attn.out_proj.weight, bo=attn.out_proj.bias, ln_g=n2.weight, ln_b=n2.bias,
                                         w1=layer.linear1.weight, b1=layer.linear1.bias, w2=layer.linear2.weight,
                                         b2=layer.linear2.bias),
                              {k: pv[8 * i_last + o] for k, o in (("wo", 2), ("w1", 4), ("w2", 6))},
                              {k: ptv[8 * i_last + o] for k, o in (("wo", 2), ("w1", 4), ("w2", 6))}, n2.eps, p), None
        if sh[0] is not None and a.dtype == torch.bfloat16 and _fused_norm_ok(x, None) \
                and layer.linear1.out_features % 8 == 0 and layer.linear1.out_features <= 2048:
            # update path: out_proj + add + LayerNorm, linear1 + ReLU + dropout, linear2 + add + LayerNorm as three ops
            n2 = layer.norm2
            mid = HLink() if sh[13] is not None else None  # out_proj's add + LayerNorm <-> linear1's input gradient
            x, h = _LinearAddLayerNorm.apply(a, attn.out_proj.weight, attn.out_proj.bias, sh[2], sh[3], x, n2.weight,
                                             n2.bias, n2.eps, p, sh[9], None, None, mid, sh[10])
            link = FFNLink(p) if next_norm is not None else None
            f = _LinearReluDropout.apply(h, layer.linear1.weight, layer.linear1.bias, sh[4], sh[5], p, link, mid, sh[13])
            if next_norm is None:
                if sh[6] is not None and layer.linear2.bias is not None:
                    # the encoder's output: residual add + dropout + bf16 cast in one launch (the heads read bf16 anyway)
                    return _LinearAddCast.apply(f, layer.linear2.weight, layer.linear2.bias, sh[6], sh[7], x, p), None
                f = _linear(f, layer.linear2.weight, layer.linear2.bias, sh[6], sh[7])
                return x + F.dropout(f, p, self.training), None
            return _LinearAddLayerNorm.apply(f, layer.linear2.weight, layer.linear2.bias, sh[6], sh[7], x, next_norm.weight,
                                             next_norm.bias, next_norm.eps, p, sh[8], link, cls_link_out, h_link_out, sh[11])
        a = _linear(a, attn.out_proj.weight, attn.out_proj.bias, sh[2], sh[3])
        x, h = _add_norm(x, a, layer.norm2, p, self.training)
        f = F.dropout(F.relu(_linear(h, layer.linear1.weight, layer.linear1.bias, sh[4], sh[5])), p, self.training)
        f = _linear(f, layer.linear2.weight, layer.linear2.bias, sh[6], sh[7])
        if next_norm is None:
            return x + F.dropout(f, p, self.training), None
        return _add_norm(x, f, next_norm, p, self.training)

    def forward(self, src: torch.Tensor, reduction: str = "mean") -> torch.Tensor:
        """``src`` [B, 16, d_model] token embeddings (no positions yet) -> [B, d_model]."""
        if reduction not in ["mean", "cls"]:
            raise ValueError(f"reduction must be 'mean' or 'cls', got {reduction}")
        x = self.positional_encoding.forward_flat(src)
        x = torch.cat([_ExpandRows.apply(self.cls_token.to(x.dtype), x.shape[0]), x], dim=1)
        return self.encode(x, reduction)

    def embed_boards_ok(self, boards: torch.Tensor, emb_weight: torch.Tensor) -> bool:
        """The update path with packed boards at the reference shape: ``_EmbedBoards`` applies."""
        return (boards.dtype == torch.uint8 and boards.dim() == 2 and boards.shape[1] == 16 and self.d_model == 256
                and tuple(emb_weight.shape) == (256, 31) and _train_bf16(boards, emb_weight))

    def forward_boards(self, boards: torch.Tensor, emb_weight: torch.Tensor, reduction: str = "mean", tail_heads=None):
        """Packed boards u8 [B, 16] -> [B, d_model] (embedding, positions, CLS and the encoder layers); with ``tail_heads`` (see
        ``encode``) possibly the heads' outputs instead."""
        pe = self.positional_encoding.flat_table().float().contiguous()
        p = self.positional_encoding.dropout.p if self.training else 0.0
        # (the embedding kernel also normalises its rows for layers[0].norm1: LNPre)
        pre = LNPre(self.encoder.layers[0].norm1) if os.environ.get("G2048_EMBED_LN", "1") != "0" else None
        return self.encode(_EmbedBoards.apply(boards, emb_weight, pe, self.cls_token, p, pre), reduction, tail_heads, pre=pre)

    def encode(self, x: torch.Tensor, reduction: str = "mean", tail_heads=None, pre=None):
        """[B, 17, d_model] tokens (CLS first, positions added) through the encoder layers -> [B, d_model].
        ``tail_heads(o, x_cls, params, dense, transposed, eps, p) -> (logits, values)``: offered by an agent whose heads can run
        inside the fused CLS tail; when the update path takes it, the return value is that tuple instead of the features."""
        if reduction not in ["mean", "cls"]:
            raise ValueError(f"reduction must be 'mean' or 'cls', got {reduction}")
        layers = self.encoder.layers
        last = len(layers) - 1
        sh = self._bf16_weights(x)
        # h_links[i]: joins the node that produces layer i's normalised input with that layer's in_proj (whose fragment-packed
        # transposed weight sh[i][12] exists for the layers in front of the last one): see hip_ops.HLink
        h_links = [HLink() if sh[i][12] is not None else None for i in range(len(layers))]
        x, h = _add_norm(x, None, layers[0].norm1, 0.0, self.training, h_link=h_links[0], pre=pre)
        # update path with a CLS-only last layer: its CLS-row gradient goes straight into the backward kernel of the layer
        # before it (ClsLink) instead of through a zero-filled [B, 17, 256] tensor
        cls_link = ClsLink() if (reduction == "cls" and last >= 1 and sh[0][0] is not None) else None
        for i, layer in enumerate(layers):
            x, h = self._layer(layer, x, h, layers[i + 1].norm1 if i < last else None,
                               cls_only=(reduction == "cls" and i == last), sh=sh[i],
                               cls_link_out=cls_link if i == last - 1 else None,
                               cls_link_in=cls_link if i == last else None,
                               tail_heads=tail_heads if (i == last and self.encoder.norm is None) else None,
                               h_link_in=h_links[i], h_link_out=h_links[i + 1] if i < last else None)
            if isinstance(x, tuple):  # (logits, values) from the fused CLS tail
                return x
        if self.encoder.norm is not None:
            x = self.encoder.norm(x)
        if reduction == "cls":
            # a CLS-only last layer leaves [B, 1, D]: a view (its backward is free) instead of a select (zero-fill + copy)
            return x.reshape(x.shape[0], x.shape[2]) if x.shape[1] == 1 else x[:, 0, :]
        return x[:, 1:, :].mean(dim=1)
