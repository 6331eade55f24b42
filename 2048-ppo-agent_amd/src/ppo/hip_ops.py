"""Autograd bindings of the HIP kernels used by the PPO update (C ABI in include/g2048.h, ctypes in src/g2048/native.py).

Each ``torch.autograd.Function`` here stands in for a group of PyTorch operators inside the reference's modules
(nn.TransformerEncoderLayer of src/ppo/transformer_encoder.py:138-148, the actor/critic heads of src/ppo/ppo_agent.py)
on the update path = HIP device + gradients wanted + bf16 autocast (``_train_bf16``); everywhere else (CPU, fp32, eval)
the callers fall back to the plain PyTorch operators.  INTEGRATION.md lists which operator each one replaces.
"""
import weakref

import torch
import torch.nn as nn
import torch.nn.functional as F


class Bf16Shadow:
    """bf16 copies of a list of f32 master parameters inside one flat buffer, refreshed by ONE multi-tensor copy
    whenever a master changed (in-place updates bump ``Tensor._version``; a re-allocated master changes the data
    pointer).  Under autocast every minibatch otherwise re-casts each weight and bias with its own 4-microsecond kernel
    (about 50 launches per PPO minibatch, and the update is launch-bound)."""

    _live = weakref.WeakSet()

    def __init__(self, params, transposed=(), packed=(), packed_only=(), packed_t_only=()):
        self.params = list(params)
        self.key, self.flat, self.views = None, None, None
        self.transposed = tuple(transposed)  # indices of 2-D params that also get a [in, out] copy (``tviews[i]``)
        self.tviews = {}
        # indices of 2-D params that also get fragment-packed copies of the tensor and of its transpose (``pviews[i]``,
        # ``ptviews[i]``: flat bf16 tensors in the MFMA operand order of include/g2048.h, read by the fused CLS tail kernels)
        self.packed = tuple(packed)
        # the same, one orientation only: ``packed_only`` the tensor itself (forward operand of g2048_linear_add_ln_fwd),
        # ``packed_t_only`` its transpose (operand of the fused input-gradient GEMM, g2048_linear_add_ln_bwd)
        self.packed_only, self.packed_t_only = tuple(packed_only), tuple(packed_t_only)
        self.pviews, self.ptviews = {}, {}
        # set by an optimiser that rewrites the shadow together with the parameters (optim.flat_step.FlatAdamWStep): the
        # shadow then copies only when its key is stale (someone else changed a parameter), also during a hipGraph capture
        self.maintainer = None
        Bf16Shadow._live.add(self)

    def __getstate__(self):
        """copy.deepcopy / pickle of an agent: the copy starts cold - no buffers and no maintainer (both belong to the
        original's optimiser, which must not travel with a pickled module) - and registers itself like a new shadow."""
        return {"params": self.params, "transposed": self.transposed, "packed": self.packed, "packed_only": self.packed_only,
                "packed_t_only": self.packed_t_only}

    def __setstate__(self, state):
        self.__init__(state["params"], state["transposed"], state.get("packed", ()), state.get("packed_only", ()),
                      state.get("packed_t_only", ()))

    def invalidate(self):
        self.key = None

    @staticmethod
    def invalidate_all():
        """Force the next use of every shadow to re-copy (called before a hipGraph capture so that the copy becomes
        part of the graph: a replay runs no Python and would otherwise read stale shadows)."""
        for s in list(Bf16Shadow._live):
            if not s.maintained():  # a maintained shadow is rewritten by the optimiser kernel, outside any graph
                s.invalidate()

    def maintained(self) -> bool:
        """An optimiser rewrites this very object's buffers (a copy.deepcopy / unpickled shadow carries the attribute of its
        original but is not in the optimiser's list)."""
        m = self.maintainer
        return m is not None and any(s is self for s in getattr(m, "_shadows", ()))

    def current_key(self):
        ps = self.params
        return (ps[0].data_ptr(), sum(p._version for p in ps))

    def __call__(self):
        ps = self.params
        key = self.current_key()
        # while a hipGraph is being captured the copy must be part of the graph whatever the key says: a replay runs no
        # Python, so a shadow that skipped the copy here (e.g. one created by copy.deepcopy / unpickling, which the
        # _live registry never saw and invalidate_all() therefore missed) would keep its capture-time weights forever
        if key != self.key or (not self.maintained() and ps[0].is_cuda and torch.cuda.is_current_stream_capturing()):
            if self.flat is None or self.flat.device != ps[0].device:
                offs, n = [], 0
                for q in ps:
                    offs.append(n)
                    n += (q.numel() + 63) // 64 * 64  # keep every view 128-byte aligned for the GEMMs
                self.flat = torch.empty(n, dtype=torch.bfloat16, device=ps[0].device)
                self.views = [self.flat[o:o + q.numel()].view(q.shape) for o, q in zip(offs, ps)]
                self.tviews = {i: torch.empty(ps[i].shape[::-1], dtype=torch.bfloat16, device=ps[0].device)
                               for i in self.transposed}
                self.pviews = {i: torch.empty(ps[i].numel(), dtype=torch.bfloat16, device=ps[0].device)
                               for i in self.packed + self.packed_only}
                self.ptviews = {i: torch.empty(ps[i].numel(), dtype=torch.bfloat16, device=ps[0].device)
                                for i in self.packed + self.packed_t_only}
            with torch.no_grad():
                torch._foreach_copy_(self.views, [q.detach() for q in ps])
                for i, tv in self.tviews.items():
                    tv.copy_(self.views[i].t())
                if self.pviews or self.ptviews:
                    from ..g2048 import native as nv

                    for i in self.pviews:
                        self.pviews[i].copy_(nv.pack_fragments(self.views[i]))
                    for i in self.ptviews:
                        self.ptviews[i].copy_(nv.pack_fragments(self.views[i].t()))
            # a copy recorded into a hipGraph has not run yet: leave the key stale so that the next eager use copies for real
            self.key = None if (ps[0].is_cuda and torch.cuda.is_current_stream_capturing()) else key
        return self.views


class GradSink:
    """All second-stage gradient reductions of one backward pass in ONE launch (``g2048_reduce_jobs``).

    ``targets`` maps ``id(parameter)`` to the f32 buffer its gradient must end up in (a view of the optimiser's flat
    gradient buffer).  While a sink is active (``with grad_sink(sink): loss.backward()``) the autograd nodes of this module
    do not reduce their weight / bias / LayerNorm gradients themselves: they leave the partials where the first stage put
    them (the 16 split-K slices of a weight gradient, the per-workgroup column sums of a bias gradient), register a job
    and return None for that input; ``flush()`` then sums every job straight into its target.  That replaces ~50
    launches per minibatch (at::sum per weight, k_colsum_final per bias, bf16 -> f32 copies) by one.
    ``written`` holds the ids of the parameters whose gradient the sink produced (autograd never saw them).

    Concurrency: ``GradSink.active`` (like ``Bf16Shadow._live`` and the capture-site counter of ``_seed_pair``) is
    PROCESS-GLOBAL state, on purpose: the backward nodes that read it run on autograd's device worker thread, not on the thread
    that entered ``grad_sink``, so a thread-local would be invisible to them.  The consequence is the engine's model of one
    process per GPU with ONE update in flight: two threads running backward passes of this module at the same time would
    share (and corrupt) the job list.  ``grad_sink.__enter__`` refuses to nest over a different sink for that reason."""

    active = None

    def __init__(self, targets: dict, early=None, on_early=None):
        """``early``: ids of parameters whose gradients are final long before the backward pass ends (the fused CLS tail: the last
        layer's out_proj .. both heads, 1.4 M of the default model's 3.96 M parameters); their jobs are kept apart, and when the
        node that produces them says so (``early_complete``) they are summed by a launch of their own and ``on_early()`` runs --
        the trainer starts the all-reduce of that part of the bucket there, under the backward of the three full layers."""
        self.targets, self.jobs, self.written = targets, [], set()
        self.dw_jobs = []  # minibatch-sized weight gradients deferred to flush(): one grouped launch (g2048_dweight_jobs)
        self.early = frozenset(early or ())
        self.early_jobs, self.on_early, self.early_done = [], on_early, False

    def takes(self, *params) -> bool:
        return all(p is not None and id(p) in self.targets for p in params)

    def add(self, param, src: torch.Tensor, part_stride: int, n: int, parts: int, dst_offset: int = 0, transpose_rows: int = 0):
        """grad(param).flatten()[dst_offset : dst_offset + n] = sum over parts of src (src's first element = part 0,
        column 0; the tensor is kept alive until ``flush``); ``transpose_rows`` R: the summed [R][n / R] matrix is stored
        transposed."""
        dst = self.targets[id(param)].view(-1)[dst_offset:dst_offset + n]
        (self.early_jobs if (id(param) in self.early and not self.early_done) else self.jobs).append(
            (src, dst, part_stride, n, parts, transpose_rows))
        self.written.add(id(param))

    def early_complete(self):
        """Called by the node that owns the early parameters once all their jobs are registered: sum them now (one launch) and
        tell the owner.  A no-op unless EVERY early parameter has been written (else ``flush`` sums everything at the end)."""
        if self.early_done or not self.early or not self.early <= self.written:
            return
        from ..g2048 import native as nv

        nv.reduce_jobs(self.early_jobs)
        self.early_jobs, self.early_done = [], True
        if self.on_early is not None:
            self.on_early()

    def add_dweight(self, weight, bias, dy2: torch.Tensor, x2: torch.Tensor, slices: int, w_offset: int = 0, b_offset: int = 0,
                    parts_dtype=torch.bfloat16):
        """dW = dy2^T x2 (and db = column sums of dy2) computed at ``flush`` together with every other deferred product; the partials
        are registered for the reduction now.  dy2 / x2 stay alive until then (~0.5 GB at minibatch 2048)."""
        N, K = dy2.shape[1], x2.shape[1]
        parts = torch.empty((slices, N, K), dtype=parts_dtype, device=dy2.device)
        cs = torch.empty((slices, N), dtype=torch.float32, device=dy2.device) if bias is not None else None
        self.dw_jobs.append((dy2, x2, parts, cs))
        self.add(weight, parts, N * K, N * K, slices, w_offset)
        if bias is not None:
            self.add(bias, cs, N, N, slices, b_offset)

    def flush(self):
        from ..g2048 import native as nv

        if self.dw_jobs:
            nv.dweight_jobs(self.dw_jobs)
            self.dw_jobs = []
        nv.reduce_jobs(self.early_jobs + self.jobs)
        self.jobs, self.early_jobs = [], []


class grad_sink:
    """``with grad_sink(sink): loss.backward()`` -- activates ``sink`` (None: no-op) and flushes it on exit."""

    def __init__(self, sink):
        self.sink = sink

    def __enter__(self):
        if GradSink.active is not None and self.sink is not None and GradSink.active is not self.sink:
            raise RuntimeError("a GradSink is already active in this process: one update in flight per process (see GradSink)")
        self.prev, GradSink.active = GradSink.active, self.sink
        return self.sink

    def __exit__(self, exc_type, *exc):
        GradSink.active = self.prev
        if self.sink is not None and exc_type is None:
            self.sink.flush()
        return False


def _sink_for(*params):
    """The active sink if it takes the gradients of all ``params`` (None entries = absent parameters are ignored)."""
    sink = GradSink.active
    if sink is None:
        return None
    ps = [p for p in params if p is not None]
    return sink if ps and sink.takes(*ps) else None


def _sum_f32(t: torch.Tensor, dim: int = 0) -> torch.Tensor:
    return torch.sum(t, dim, dtype=torch.float32)


def _colsum(t: torch.Tensor, out=None) -> torch.Tensor:
    """f32 column sums of a [rows, N] device tensor (bias gradients): ``g2048_colsum`` (fixed summation order and safe
    inside a replayed hipGraph, unlike at::sum's semaphore-based cross-workgroup stage) when the shape allows."""
    N = t.shape[-1]
    if t.is_cuda and t.dim() == 2 and t.stride(1) == 1 and N % 4 == 0 and t.stride(0) % 4 == 0 \
            and t.dtype in (torch.bfloat16, torch.float32) and t.data_ptr() % 16 == 0:
        from ..g2048 import native as nv

        return nv.colsum(t, out)
    if t.is_cuda and torch.cuda.is_current_stream_capturing():
        # at::sum's cross-workgroup stage yields wrong values when replayed from a hipGraph on this stack (NOTES.md 3):
        # refuse to capture it, PPOTrainer._build_graph then falls back to the eager update
        raise RuntimeError(f"column sum of a {tuple(t.shape)} {t.dtype} tensor (strides {t.stride()}) is outside "
                           "g2048_colsum's shapes and at::sum must not be captured in a hipGraph")
    if out is None:
        return _sum_f32(t)
    return torch.sum(t, 0, dtype=torch.float32, out=out)


class _ExpandRows(torch.autograd.Function):
    """``t.expand(B, -1, -1)`` of a [1, 1, D] parameter whose gradient (a sum over B rows) goes through ``_colsum``."""

    @staticmethod
    def forward(ctx, t, B):
        return t.expand(B, -1, -1)

    @staticmethod
    def backward(ctx, g):
        return _colsum(g.reshape(g.shape[0], -1)).view(1, 1, -1), None


def _hip_linear(x2: torch.Tensor, wb: torch.Tensor, bias_f32=None):
    """x2 @ wb^T (+ bias) through ``g2048_linear_bf16`` for the shapes where its weights-stationary layout beats
    hipBLASLt's pick on this chip (K <= 256 with a wide output: linear1 forward 55 -> 38 us, linear2 input gradient
    48 -> 38 us at 34 816 tokens; tools/probe_linear.py), else None."""
    K, N = x2.shape[-1], wb.shape[0]
    if K > 256 or N % 256 or x2.shape[0] < 4096:
        return None
    from ..g2048 import native as nv

    if not nv.linear_ok(x2, wb):
        return None
    return nv.linear_bf16(x2, wb, bias_f32)


def _stationary_ok(x2: torch.Tensor, wb: torch.Tensor) -> bool:
    """The shapes of the fused feed-forward kernels (weights-stationary GEMM: K <= 256, wide output, many tokens)."""
    K, N = x2.shape[-1], wb.shape[0]
    if K > 256 or N < 512 or N % 256 or x2.shape[0] < 4096:
        return False
    from ..g2048 import native as nv

    return nv.linear_ok(x2, wb)


def _rowgemm_on() -> bool:
    """G2048_ROWGEMM=0 switches the fused Linear + add + LayerNorm launches (g2048_linear_add_ln_fwd / _bwd) off: the A/B switch."""
    import os

    return os.environ.get("G2048_ROWGEMM", "1").strip().lower() not in ("0", "false", "no", "off")


class HLink:
    """Joins the node that PRODUCES a normalised activation h (``_AddLayerNorm`` / ``_LinearAddLayerNorm``) with the Linear that consumes
    it (in_proj: ``_LinearSplitK``, linear1: ``_LinearReluDropout``): in the backward the consumer does not run its input-gradient GEMM
    but leaves its operands here (``pending`` = (dy [T, K] bf16, fragment-packed transpose of its weight)) and returns no gradient for
    h; the producer's backward then runs GEMM and LayerNorm backward as ONE launch (``g2048_linear_add_ln_bwd``): the bf16 [T, 256]
    gradient between them never exists.  ``armed``: set by the producer's forward when its backward can do that."""

    __slots__ = ("armed", "pending")

    def __init__(self):
        self.armed, self.pending = False, None

    def offer(self, dy2: torch.Tensor, wt_packed, tile_stride: int = 0, extra=None, extra_period: int = 1) -> bool:
        """Called by the consumer's backward: True if the producer will compute the input gradient itself.  ``tile_stride``: the packed
        weight is K columns of a wider packed matrix (``native.rowgemm_ok``); ``extra`` bf16 [T / extra_period, 256]: added to the
        gradient on the rows t % extra_period == 0 (the CLS rows' share of the last layer's query projection)."""
        from ..g2048 import native as nv

        if (not self.armed or wt_packed is None or self.pending is not None or not _rowgemm_on()
                or not nv.rowgemm_ok(dy2, wt_packed, tile_stride)):
            return False
        self.pending = (dy2, wt_packed, dict(tile_stride=int(tile_stride), g_h_extra=extra, extra_period=int(extra_period)))
        return True

    def take(self):
        p, self.pending = self.pending, None
        return p


class FFNLink:
    """Shared by the two autograd nodes of one feed-forward block (``_LinearReluDropout`` -> ``_LinearAddLayerNorm``): the
    backward of the second computes linear2's input gradient already masked by the saved activation and with linear1's
    bias gradient (``g2048_linear_mask_bwd_bf16``) and leaves both here for the backward of the first, which then has
    nothing left to launch for the activation."""

    __slots__ = ("p_drop", "db", "masked", "bias_param", "db_sunk", "bits")

    def __init__(self, p_drop: float):
        self.p_drop, self.db, self.masked = float(p_drop), None, False
        self.bits = None  # the forward kernel's bit mask (output non-zero or not), read by the masked-gradient GEMM
        self.bias_param, self.db_sunk = None, False  # linear1's bias; True when a GradSink took its gradient



# the wide projections of the update (in_proj 256 -> 768, the last layer's K/V 256 -> 512) go through g2048_linear_bf16 instead of
# hipBLASLt: 2.73 -> 2.69 ms per minibatch in the pipeline (isolated and cache-warm hipBLASLt is the faster one, tools/probe_linear.py)
_SINK_SLICES = 16  # split-K slices of a weight gradient when a GradSink sums them


def _dweight_parts(dy2: torch.Tensor, x2: torch.Tensor) -> torch.Tensor:
    """The first stage of ``_dweight``: bf16 [parts, out, in] whose sum over parts is dW (parts = 16 split-K slices for a
    long token axis, else 1)."""
    T, S = x2.shape[0], _SINK_SLICES
    if T >= 16384:  # the minibatch's token axis: g2048_dweight_bf16, [128 x 128] blocks x slices = 128-256 workgroups
        from ..g2048 import native as nv

        slices = 16 if dy2.shape[1] * x2.shape[1] > 256 * 256 else 32
        if nv.dweight_ok(dy2, x2, slices):
            return nv.dweight_parts(dy2, x2, slices, block_rows=128)
    # also for the 2048-row GEMMs of the CLS-only layer and the heads: their [out, in] results are a handful of tiles with a
    # 2048-long reduction each (17 us per GEMM in the pipeline); the sink adds the slices at no extra launch
    if T % S == 0 and T // S >= 64:
        return torch.bmm(dy2.view(S, T // S, -1).transpose(1, 2), x2.view(S, T // S, -1))
    return (dy2.t() @ x2).unsqueeze(0)


def _deferred_dweight(sink, weight, bias, dy2, x2, w_offset: int = 0, b_offset: int = 0) -> bool:
    """A minibatch-sized token axis: the product joins the sink's grouped launch.  8 token slices (one per XCD): the launch as a whole
    fills the chip (~1 200 workgroups at minibatch 2048), so a product needs no more - measured 16 / 32 slices per product: 259 us for
    the launch + 40 us for the reduction of the partials, 8 slices: 250 + 32."""
    # (from 2 048 rows on: the 2048-row products of the CLS-only layer's query projection and of the MLP policy's 512 x 512 layers
    # join the launch as well - each was a batched GEMM node of 8-14 us of its own)
    if x2.shape[0] < 2048:
        return False
    from ..g2048 import native as nv

    slices, dtype = _dweight_parts_config()
    if not nv.dweight_ok(dy2, x2, slices):
        return False
    sink.add_dweight(weight, bias, dy2, x2, slices, w_offset, b_offset, parts_dtype=dtype)
    return True


def _dweight_parts_config():
    """(token slices, dtype of the partial products) of the deferred weight gradients.  Default 8 bf16 slices; G2048_DWEIGHT_PARTS =
    bf16x16 / f32x8 are the other two arms of round 4's multi-seed A/B (profiles/round4_dweight_slices_seeds.txt), read per call so a
    sweep can flip it between trainers of one process."""
    import os

    v = os.environ.get("G2048_DWEIGHT_PARTS", "bf16x8").strip().lower()
    table = {"bf16x8": (8, torch.bfloat16), "bf16x16": (16, torch.bfloat16), "f32x8": (8, torch.float32), "f32x16": (16, torch.float32)}
    if v not in table:
        raise ValueError(f"G2048_DWEIGHT_PARTS must be one of {sorted(table)}, got {v!r}")
    return table[v]


def _sink_weight_bias(sink, weight, bias, dy2, x2, w_offset: int = 0, b_offset: int = 0):
    """Weight and bias gradient of one Linear into the sink; for a minibatch-sized token axis ONE (deferred, grouped) launch produces
    the first stage of both (``g2048_dweight_jobs`` with column sums)."""
    if bias is not None and _deferred_dweight(sink, weight, bias, dy2, x2, w_offset, b_offset):
        return
    _sink_weight(sink, weight, dy2, x2, w_offset)
    if bias is not None:
        _sink_bias(sink, bias, dy2, b_offset)


def _sink_weight(sink, param, dy2, x2, dst_offset: int = 0):
    if _deferred_dweight(sink, param, None, dy2, x2, dst_offset):
        return
    parts = _dweight_parts(dy2, x2)
    n = parts.shape[1] * parts.shape[2]
    sink.add(param, parts, n, n, parts.shape[0], dst_offset)


def _colsum_sinkable(t: torch.Tensor) -> bool:
    N = t.shape[-1]
    return (t.is_cuda and t.dim() == 2 and t.stride(1) == 1 and N % 4 == 0 and N <= 1024 and t.stride(0) % 4 == 0
            and t.dtype in (torch.bfloat16, torch.float32) and t.data_ptr() % 16 == 0)


def _sink_bias(sink, param, dy2, dst_offset: int = 0):
    """Bias gradient = column sums of dy2, second stage left to the sink (first stage: g2048_colsum's partial kernel)."""
    from ..g2048 import native as nv

    N = dy2.shape[-1]
    if _colsum_sinkable(dy2):
        ws = nv.colsum_partial(dy2)
        sink.add(param, ws, N, N, ws.shape[0], dst_offset)
    else:  # widths the partial kernel does not take: a finished column sum, copied into place by the sink
        sink.add(param, _colsum(dy2).contiguous(), N, N, 1, dst_offset)


def _dweight(dy2: torch.Tensor, x2: torch.Tensor) -> torch.Tensor:
    """dW = dY^T X in f32 from bf16 operands.  It reduces over every token of the minibatch (34 816 at minibatch 2048)
    into a tiny [out, in] matrix; hipBLASLt runs that as 16-64 workgroups on 256 CUs (177 us per GEMM).  Cutting the
    token axis into 16 slices turns it into a batched GEMM with 16x the workgroups plus an f32 sum of the partials:
    42 us, 4x faster, and the f32 sum is at least as accurate as the single bf16-output GEMM it replaces."""
    T, S = x2.shape[0], _LinearSplitK.SLICES
    if T % S == 0 and T // S >= 1024:
        return _sum_f32(_dweight_parts(dy2, x2))  # g2048_dweight_bf16 when it takes the operands, else the batched GEMM
    return (dy2.t() @ x2).float()


class _LinearSplitK(torch.autograd.Function):
    """``F.linear`` in bf16 (what autocast does) with a split-K weight gradient (``_dweight``), f32 gradients for the f32
    master weight/bias produced directly by the reductions, and optional pre-cast bf16 shadows ``wb``/``bb`` of the
    masters (``Bf16Shadow``) so that no cast kernels run per call."""

    SLICES = 16

    @staticmethod
    def forward(ctx, x, weight, bias, wb=None, bb=None, h_link=None, wt_packed=None):
        ctx.h_link, ctx.wt_packed = h_link, wt_packed  # (HLink: the producer of x may run this node's input-gradient GEMM itself)
        with torch.autocast("cuda", enabled=False):
            xb = x.to(torch.bfloat16)
            if wb is None:
                wb = weight.to(torch.bfloat16)
            if bias is not None and bb is None:
                bb = bias.to(torch.bfloat16)
            y = None
            if bias is not None and bias.dtype == torch.float32:
                y = _hip_linear(xb.reshape(-1, xb.shape[-1]), wb, bias.detach())
                if y is not None:
                    y = y.view(*xb.shape[:-1], wb.shape[0])
            if y is None:
                y = F.linear(xb, wb, bb)
        ctx.save_for_backward(xb, wb)
        ctx.meta = (x.dtype, weight.dtype, None if bias is None else bias.dtype)
        ctx.params = (weight, bias)
        return y

    @staticmethod
    def backward(ctx, dy):
        xb, wb = ctx.saved_tensors
        x_dtype, w_dtype, b_dtype = ctx.meta
        with torch.autocast("cuda", enabled=False):
            dy2 = dy.reshape(-1, dy.shape[-1]).to(torch.bfloat16)
            x2 = xb.reshape(-1, xb.shape[-1])
            dx = None
            if ctx.needs_input_grad[0] and not (ctx.h_link is not None and ctx.h_link.offer(dy2.contiguous(), ctx.wt_packed)):
                dx = (dy2 @ wb).view(xb.shape).to(x_dtype)
            weight, bias = ctx.params
            sink = _sink_for(weight, bias) if w_dtype == torch.float32 and b_dtype in (None, torch.float32) else None
            if sink is not None:
                _sink_weight_bias(sink, weight, bias, dy2, x2)
                return dx, None, None, None, None, None, None
            dw = _dweight(dy2, x2).to(w_dtype)
            db = None if b_dtype is None else _colsum(dy2).to(b_dtype)
        return dx, dw, db, None, None, None, None


class _LinearRelu(torch.autograd.Function):
    """``relu(F.linear(x))`` for the hidden layers of the actor / critic heads (2048 rows per minibatch: every kernel here
    is launch-latency-bound, so the count is what matters).  Forward: bias + ReLU in the GEMM's epilogue
    (``torch._addmm_activation``).  Backward: threshold and the bias gradient's partial column sums in ONE launch
    (``g2048_relu_dropout_bwd`` with p = 0, reading the saved output as the mask) instead of threshold_backward + column sum."""

    @staticmethod
    def ok(x, weight, bias, wb, bb) -> bool:
        return (bias is not None and bias.dtype == torch.float32 and weight.dtype == torch.float32 and wb is not None
                and bb is not None and wb.shape[0] % 8 == 0 and wb.shape[0] // 8 <= 256 and wb.is_contiguous())

    @staticmethod
    def forward(ctx, x, weight, bias, wb, bb):
        with torch.autocast("cuda", enabled=False):
            x2 = x.to(torch.bfloat16).reshape(-1, x.shape[-1])
            y = torch._addmm_activation(bb, x2, wb.t())
        ctx.save_for_backward(x2, wb, y)
        ctx.params, ctx.meta = (weight, bias), (x.shape, x.dtype)
        return y.view(*x.shape[:-1], wb.shape[0])

    @staticmethod
    def backward(ctx, dy):
        from ..g2048 import native as nv

        x2, wb, y = ctx.saved_tensors
        weight, bias = ctx.params
        shape, x_dtype = ctx.meta
        sink = _sink_for(weight, bias)
        with torch.autocast("cuda", enabled=False):
            dy2 = dy.reshape(-1, dy.shape[-1]).to(torch.bfloat16).contiguous()
            dz = torch.empty_like(y)
            db = None if sink is not None else torch.empty(y.shape[-1], dtype=torch.float32, device=y.device)
            ws = nv.relu_dropout_bwd(dy2, y, dz, db, 0.0)
            dx = (dz @ wb).view(shape).to(x_dtype) if ctx.needs_input_grad[0] else None
            if sink is not None:
                sink.add(bias, ws, ws.shape[1], ws.shape[1], ws.shape[0])
                _sink_weight(sink, weight, dz, x2)
                return dx, None, None, None, None
            dw = _dweight(dz, x2)
        return dx, dw, db, None, None


class _InProjCls(torch.autograd.Function):
    """in_proj of the LAST layer when only the CLS row is wanted: q = h[:, :1] Wq^T + bq, kv = h Wkv^T + bkv, with
    the gradients of the whole in_proj weight and bias assembled in one buffer (slicing the parameter instead costs a
    zero-fill, a copy and an accumulate per slice in the backward).  h, wb, bb bf16; weight/bias the f32 masters.
    The query projection reads the CLS rows in place (row stride S * D) through ``g2048_gemm_jobs``; where that kernel does not apply
    they are first gathered into a contiguous [B, D] matrix, so that no LIBRARY GEMM of the update takes the [B, 1, D] view with row
    stride 17 * D as an operand (NOTES.md 3, "the 02:59 fault")."""

    @staticmethod
    def forward(ctx, h, weight, bias, wb, bb, h_link=None, wt_packed=None):
        from ..g2048 import native as nv

        ctx.h_link, ctx.wt_packed = h_link, wt_packed  # (HLink: the producer of h may run the K/V input-gradient GEMM itself)
        B, S, D = h.shape
        h_cls = None
        if (h.is_cuda and h.dtype == torch.bfloat16 and h.is_contiguous() and wb.dtype == torch.bfloat16 and D % 64 == 0
                and bias.dtype == torch.float32 and h.data_ptr() % 16 == 0):
            q = torch.empty((B, D), dtype=torch.bfloat16, device=h.device)
            nv.gemm_jobs([dict(segs=[(h[:, 0], wb[:D])], bias=bias.detach()[:D], y=q)], B)
            q = q.view(B, 1, D)
        else:
            h_cls = h[:, 0].contiguous()
            q = F.linear(h_cls, wb[:D], bb[:D]).view(B, 1, D)
        kv = None
        if bias.dtype == torch.float32:
            kv = _hip_linear(h.reshape(B * S, D), wb[D:], bias.detach()[D:])
            if kv is not None:
                kv = kv.view(B, S, 2 * D)
        if kv is None:
            kv = F.linear(h, wb[D:], bb[D:])
        ctx.save_for_backward(h, wb)
        ctx.params = (weight, bias)
        return q, kv

    @staticmethod
    def backward(ctx, dq, dkv):
        from ..g2048 import native as nv

        h, wb = ctx.saved_tensors
        B, S, D = h.shape
        dq2, dkv2 = dq.reshape(B, D).contiguous(), dkv.reshape(B * S, 2 * D).contiguous()
        dh = None
        link, wtp = ctx.h_link, ctx.wt_packed
        if link is not None and wtp is not None and D == 256:
            # the node that produced h runs dkv . W_kv inside its LayerNorm backward (g2048_linear_add_ln_bwd): W_kv^T = columns
            # D .. 3 D of the packed in_proj^T [256][3 D] (k-steps D / 16 onwards, tile stride 3 D / 16 fragments), and the CLS rows' share
            # of the query projection travels as an extra term on every S-th row
            if link.offer(dkv2, wtp[(D // 16) * 512:], tile_stride=(3 * D // 16) * 512, extra=(dq2 @ wb[:D]).contiguous(), extra_period=S):
                dh = None
            else:
                link = None
        else:
            link = None
        if link is None:
            dh = (dkv2 @ wb[D:]).view(B, S, D)
            dh0 = dh[:, 0]  # [B, D] with row stride S * D: the CLS rows receive the query's gradient in the GEMM's epilogue
            torch.addmm(dh0, dq2, wb[:D], out=dh0)
        weight, bias = ctx.params
        sink = _sink_for(weight, bias)
        # the CLS rows in place for our own weight-gradient GEMM (the grouped launch, which also sums dq's columns = the query bias
        # gradient); a contiguous copy for everything else
        h_cls = h[:, 0]
        if not (sink is not None and B >= 2048 and nv.dweight_ok(dq2, h_cls, _dweight_parts_config()[0])):
            h_cls = h_cls.contiguous()
        if sink is not None:  # the four pieces land in their slices of the in_proj gradients
            _sink_weight_bias(sink, weight, bias, dq2, h_cls, 0, 0)
            _sink_weight_bias(sink, weight, bias, dkv2, h.view(B * S, D), D * D, D)
            return dh, None, None, None, None, None, None

        def weight_grads():
            dw = torch.empty((3 * D, D), dtype=torch.float32, device=h.device)
            dw[:D] = dq2.t() @ h_cls
            dw[D:] = _dweight(dkv2, h.view(B * S, D))
            db = torch.empty(3 * D, dtype=torch.float32, device=h.device)
            _colsum(dq2, db[:D])
            _colsum(dkv2, db[D:])
            return dw, db

        dw, db = weight_grads()
        return dh, dw, db, None, None, None, None


def _seed() -> int:
    return int(torch.randint(0, 2 ** 62, (1,)).item())  # CPU generator: no device sync


_GRAPH_SEED = {}  # device -> int64[1] word the dropout kernels of a captured region mix into their seed
_capture_site = [0]


def graph_seed_state(device) -> torch.Tensor:
    """The device-resident seed word for hipGraph-captured dropout: whoever replays a captured update advances it
    (``.add_(1)``, inside or outside the graph) so that every replay draws new masks."""
    device = torch.device(device)
    if device not in _GRAPH_SEED:
        _GRAPH_SEED[device] = torch.randint(0, 2 ** 62, (1,), dtype=torch.int64).to(device)
    return _GRAPH_SEED[device]


def _seed_pair(t: torch.Tensor, p_drop: float):
    """(seed, seed_state address) for a dropout kernel launch: a fresh host-side seed in eager mode; while a hipGraph is
    being captured the launch arguments are frozen, so the seed is a per-call-site constant and the randomness comes
    from ``graph_seed_state`` read by the kernel at run time."""
    if p_drop <= 0:
        return 0, 0
    if torch.cuda.is_current_stream_capturing():
        _capture_site[0] += 1
        return (_capture_site[0] * 0x9E3779B97F4A7C15) & (2 ** 62 - 1), graph_seed_state(t.device).data_ptr()
    return _seed(), 0


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
        seed = _seed_pair(qkv, p_drop)
        base = qkv.data_ptr()
        strides = (S * W, W) * 3
        nv.attn_fwd(base, base + 2 * hw, base + 4 * hw, o, lse, B, nhead, S, strides, (hw // nhead) ** -0.5, p_drop, *seed)
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
                    nhead, S, (S * W, W) * 3, (hw // nhead) ** -0.5, p_drop, *seed)
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
        seed = _seed_pair(kv, p_drop)
        kb = kv.data_ptr()
        nv.attn_fwd(q.data_ptr(), kb, kb + 2 * hw, o, lse, B, nhead, 1, (hw, 0, S * W, W, S * W, W),
                    (hw // nhead) ** -0.5, p_drop, *seed)
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
                    (hw, 0, S * W, W, S * W, W), (hw // nhead) ** -0.5, p_drop, *seed)
        return dq, dkv, None, None


class _AddLayerNorm(torch.autograd.Function):
    """``x_new = x + dropout(a); h = LayerNorm(x_new).bfloat16()`` in one HIP kernel each way (``g2048_add_ln_fwd/bwd``):
    the tail of one pre-norm sub-layer fused with the head of the next.  ``a`` None: ``h = LayerNorm(x)`` only.
    x f32 [..., 256] (a [B, 1, 256] slice of the residual stream is read in place), a bf16; returns (x_new, h)."""

    @staticmethod
    def forward(ctx, x, a, gamma, beta, eps, p_drop, h_link=None, pre=None):
        from ..g2048 import native as nv

        ctx.set_materialize_grads(False)  # an unused output's gradient arrives as None, not as a zero-filled tensor
        ctx.h_link = h_link
        if h_link is not None:
            h_link.armed = _rowgemm_on()
        x_in = x
        x, row_stride = _residual_rows(x)
        T = x.numel() // 256
        gamma, beta = gamma.contiguous(), beta.contiguous()
        x_new = None
        seed = (0, 0)
        if a is None and pre is not None and pre.h is not None and pre.x is not None and pre.x.data_ptr() == x_in.data_ptr() \
                and x is x_in and pre.h.shape == x.shape:
            h, stats = pre.h, pre.stats  # the producer of x normalised its rows already (LNPre)
            pre.x = pre.h = pre.stats = None
        else:
            h = torch.empty(x.shape, dtype=torch.bfloat16, device=x.device)
            stats = torch.empty((2, T), dtype=torch.float32, device=x.device)
            if a is not None:
                a = a.contiguous()
                x_new = torch.empty(x.shape, dtype=torch.float32, device=x.device)
                seed = _seed_pair(x, p_drop)
            nv.add_ln_fwd(x.data_ptr(), row_stride, a, gamma, beta, x_new, h, stats[0], stats[1], T, eps, p_drop, *seed)
        ctx.params = (gamma, beta)
        if a is None:
            # x itself is the first output: the residual stream continues from THIS node, so the gradient of the stream and
            # the LayerNorm gradient meet in one backward kernel instead of an accumulate kernel over [B, 17, 256]
            ctx.save_for_backward(x, gamma, stats)
            ctx.meta = (row_stride, 0.0, (0, 0), False)
            return x, h
        ctx.save_for_backward(x_new, gamma, stats)
        ctx.meta = (256, p_drop, seed, True)
        return x_new, h

    @staticmethod
    def backward(ctx, g_x, g_h):
        from ..g2048 import native as nv

        xn, gamma, stats = ctx.saved_tensors
        row_stride, p_drop, seed, has_a = ctx.meta
        T = xn.numel() // 256
        dx = torch.empty(xn.shape, dtype=torch.float32, device=xn.device)
        da = torch.empty(xn.shape, dtype=torch.bfloat16, device=xn.device) if has_a else None
        g_x = g_x.contiguous() if g_x is not None else None
        sink = _sink_for(*ctx.params)
        pend = ctx.h_link.take() if ctx.h_link is not None else None
        if pend is not None:  # the consumer of h left its input-gradient GEMM to this node: GEMM + LayerNorm backward in one launch
            if g_h is not None:
                raise RuntimeError("the normalised activation has a second consumer besides the linked Linear")
            ws = nv.linear_add_ln_bwd(pend[0], pend[1], xn.data_ptr(), row_stride, g_x, stats[0], stats[1], gamma, dx, da, p_drop, *seed,
                                      **pend[2])
            dparams = None if sink is not None else ws.sum(0).view(3, 256)
        else:
            if g_h is None:
                g_h = torch.zeros(xn.shape, dtype=torch.bfloat16, device=xn.device)
            dparams = None if sink is not None else torch.empty((3, 256), dtype=torch.float32, device=xn.device)
            ws = nv.add_ln_bwd(xn.data_ptr(), row_stride, g_x, g_h.contiguous(), stats[0], stats[1], gamma, dx, da, dparams, T, p_drop,
                               *seed)
        if sink is not None:
            sink.add(ctx.params[0], ws, 768, 256, ws.shape[0])
            sink.add(ctx.params[1], ws[:, 256:], 768, 256, ws.shape[0])
            return dx, da, None, None, None, None, None, None
        return dx, da, dparams[0], dparams[1], None, None, None, None


def _residual_rows(x: torch.Tensor):
    """(x, row stride): a [B, 1, 256] slice of the residual stream is read in place by the add+LayerNorm kernels."""
    if x.dim() == 3 and x.shape[1] == 1 and x.stride(2) == 1 and x.stride(0) % 4 == 0:
        return x, x.stride(0)
    return x.contiguous(), x.shape[-1]


class ClsLink:
    """Joins the node that PRODUCES the residual stream entering the CLS-only last layer with the node that takes its CLS
    rows: the gradient of those rows ([B, 1, 256]) is handed over as it is and enters the producer's backward kernel with
    ``g_x_period`` = sequence length, instead of being scattered into a zero-filled [B, 17, 256] tensor first."""

    __slots__ = ("g", "attached")

    def __init__(self):
        self.g, self.attached = None, False


class _ClsRows(torch.autograd.Function):
    """``x[:, :1]`` of the residual stream; with an attached ``ClsLink`` the backward passes its gradient through the link."""

    @staticmethod
    def forward(ctx, x, link):
        ctx.link = link if (link is not None and link.attached) else None
        ctx.shape = x.shape
        return x[:, :1]

    @staticmethod
    def backward(ctx, g):
        if ctx.link is not None:
            ctx.link.g = g.contiguous()
            return None, None
        gx = g.new_zeros(ctx.shape)
        gx[:, :1] = g
        return gx, None


class _LinearAddLayerNorm(torch.autograd.Function):
    """``a = Linear(u); x_new = x + dropout(a); h = LayerNorm(x_new).bfloat16()``: the closing Linear of a sub-layer
    (out_proj / linear2) fused with ``_AddLayerNorm``.  The Linear's bias gradient (column sums of da) comes out of
    ``g2048_add_ln_bwd`` for free, its weight gradient is the split-K product.  u bf16, wb/bb bf16 shadows of the f32
    masters weight/bias, x f32; returns (x_new, h)."""

    @staticmethod
    def forward(ctx, u, weight, bias, wb, bb, x, gamma, beta, eps, p_drop, wbT=None, link=None, cls_link=None, h_link=None,
                w_packed=None):
        from ..g2048 import native as nv

        ctx.set_materialize_grads(False)  # an unused output's gradient arrives as None, not as a zero-filled tensor
        ctx.wbT, ctx.link = wbT, link
        ctx.params = (weight, bias, gamma, beta)
        ctx.cls_link = cls_link
        ctx.h_link = h_link
        if h_link is not None:
            h_link.armed = _rowgemm_on()
        if cls_link is not None:
            cls_link.attached = x.dim() == 3  # [B, S, 256]: the period of the CLS rows is S
        x, row_stride = _residual_rows(x)
        T = x.numel() // 256
        gamma, beta = gamma.contiguous(), beta.contiguous()
        h = torch.empty(x.shape, dtype=torch.bfloat16, device=x.device)
        x_new = torch.empty(x.shape, dtype=torch.float32, device=x.device)
        stats = torch.empty((2, T), dtype=torch.float32, device=x.device)
        seed = _seed_pair(x, p_drop)
        u2 = u.reshape(-1, u.shape[-1])
        if (w_packed is not None and _rowgemm_on() and u2.shape[0] == T and T >= 4096 and (bias is None or bias.dtype == torch.float32)
                and nv.rowgemm_ok(u2, w_packed)):
            # the Linear, dropout, the residual add and the LayerNorm in one launch: the Linear's bf16 output stays in LDS
            nv.linear_add_ln_fwd(u2, w_packed, None if bias is None else bias.detach(), x.data_ptr(), row_stride, gamma, beta, x_new, h,
                                 stats[0], stats[1], eps, p_drop, *seed)
        else:
            with torch.autocast("cuda", enabled=False):
                a = _hip_linear(u2, wb, bias.detach()) if bias is not None and u.is_contiguous() else None
                a = F.linear(u, wb, bb) if a is None else a.view(*u.shape[:-1], wb.shape[0])
            nv.add_ln_fwd(x.data_ptr(), row_stride, a, gamma, beta, x_new, h, stats[0], stats[1], T, eps, p_drop, *seed)
        ctx.save_for_backward(u, wb, x_new, gamma, stats)
        ctx.meta = (p_drop, seed)
        return x_new, h

    @staticmethod
    def backward(ctx, g_x, g_h):
        from ..g2048 import native as nv

        u, wb, xn, gamma, stats = ctx.saved_tensors
        p_drop, seed = ctx.meta
        T = xn.numel() // 256
        dx = torch.empty(xn.shape, dtype=torch.float32, device=xn.device)
        da = torch.empty(xn.shape, dtype=torch.bfloat16, device=xn.device)
        weight, bias, gamma_p, beta_p = ctx.params
        sink = _sink_for(weight, bias, gamma_p, beta_p)
        dparams = None if sink is not None else torch.empty((3, 256), dtype=torch.float32, device=xn.device)
        period = 1
        cl = ctx.cls_link
        if cl is not None and cl.g is not None:  # the stream's only gradient: the CLS rows, from _ClsRows
            if g_x is not None:
                raise RuntimeError("the residual stream entering the CLS-only layer has a second consumer")
            g_x, period, cl.g = cl.g, xn.shape[1], None
        pend = ctx.h_link.take() if ctx.h_link is not None else None
        if pend is not None:  # the consumer of h left its input-gradient GEMM to this node: GEMM + LayerNorm backward in one launch
            if g_h is not None:
                raise RuntimeError("the normalised activation has a second consumer besides the linked Linear")
            ws = nv.linear_add_ln_bwd(pend[0], pend[1], xn.data_ptr(), 256, None if g_x is None else g_x.contiguous(), stats[0],
                                      stats[1], gamma, dx, da, p_drop, *seed, g_x_period=period, **pend[2])
            if sink is None:
                dparams = ws.sum(0).view(3, 256)
        else:
            if g_h is None:
                g_h = torch.zeros(xn.shape, dtype=torch.bfloat16, device=xn.device)
            ws = nv.add_ln_bwd(xn.data_ptr(), 256, None if g_x is None else g_x.contiguous(), g_h.contiguous(), stats[0], stats[1],
                               gamma, dx, da, dparams, T, p_drop, *seed, g_x_period=period)
        if sink is not None:
            sink.add(gamma_p, ws, 768, 256, ws.shape[0])
            sink.add(beta_p, ws[:, 256:], 768, 256, ws.shape[0])
            sink.add(bias, ws[:, 512:], 768, 256, ws.shape[0])
        with torch.autocast("cuda", enabled=False):
            da2, u2 = da.view(T, 256), u.reshape(T, -1)
            du = None
            link = ctx.link
            if ctx.needs_input_grad[0]:
                if link is not None and link.bits is not None and ctx.wbT is not None and _stationary_ok(da2, ctx.wbT):
                    # u = dropout(relu(.)) of the same block: mask + 1/keep + linear1's bias gradient in the GEMM epilogue
                    sink1 = _sink_for(link.bias_param)
                    if sink1 is not None:
                        du, ws1 = nv.linear_mask_bwd(da2, ctx.wbT, link.bits, link.p_drop, final=False)
                        sink1.add(link.bias_param, ws1, ws1.shape[1], ws1.shape[1], ws1.shape[0])
                        link.db, link.db_sunk = None, True
                    else:
                        du, link.db = nv.linear_mask_bwd(da2, ctx.wbT, link.bits, link.p_drop)
                    link.masked, link.bits = True, None
                    du = du.view(u.shape)
                else:
                    du = _hip_linear(da2, ctx.wbT) if ctx.wbT is not None else None
                    du = (da2 @ wb if du is None else du).view(u.shape)
            if sink is not None:
                _sink_weight(sink, weight, da2, u2)
                return du, None, None, None, None, dx, None, None, None, None, None, None, None, None, None
            dw = _dweight(da2, u2)
        return du, dw, dparams[2], None, None, dx, dparams[0], dparams[1], None, None, None, None, None, None, None


class _LinearAddCast(torch.autograd.Function):
    """``bf16(x + dropout(Linear(u)))``: the closing Linear of the LAST sub-layer (linear2 of the last encoder layer, no
    LayerNorm follows), whose output the heads read in bf16.  ``g2048_add_ln_fwd/bwd`` with gamma = NULL: one launch instead
    of dropout + add + cast (+ a second cast for the critic head), and in the backward one launch instead of two casts, an
    add, masked_scale and the bias gradient's column sum.  u bf16, wb/bb bf16 shadows of the f32 masters weight/bias, x f32
    rows of 256; returns the bf16 tensor."""

    @staticmethod
    def forward(ctx, u, weight, bias, wb, bb, x, p_drop):
        from ..g2048 import native as nv

        with torch.autocast("cuda", enabled=False):
            a = F.linear(u, wb, bb)
        ctx.params = (weight, bias)
        xr, row_stride = _residual_rows(x)
        T = xr.numel() // 256
        h = torch.empty(xr.shape, dtype=torch.bfloat16, device=x.device)
        seed = _seed_pair(xr, p_drop)
        nv.add_ln_fwd(xr.data_ptr(), row_stride, a, None, None, None, h, None, None, T, 0.0, p_drop, *seed)
        ctx.save_for_backward(u, wb)
        ctx.meta = (p_drop, seed, tuple(xr.shape))
        return h

    @staticmethod
    def backward(ctx, g_h):
        from ..g2048 import native as nv

        u, wb = ctx.saved_tensors
        p_drop, seed, shape = ctx.meta
        T = g_h.numel() // 256
        dx = torch.empty(shape, dtype=torch.float32, device=g_h.device)
        da = torch.empty(shape, dtype=torch.bfloat16, device=g_h.device)
        weight, bias = ctx.params
        sink = _sink_for(weight, bias)
        dparams = None if sink is not None else torch.empty((3, 256), dtype=torch.float32, device=g_h.device)
        ws = nv.add_ln_bwd(0, 256, None, g_h.contiguous(), None, None, None, dx, da, dparams, T, p_drop, *seed)
        with torch.autocast("cuda", enabled=False):
            da2, u2 = da.view(T, 256), u.reshape(T, -1)
            du = (da2 @ wb).view(u.shape) if ctx.needs_input_grad[0] else None
            if sink is not None:
                sink.add(bias, ws[:, 512:], 768, 256, ws.shape[0])
                _sink_weight(sink, weight, da2, u2)
                return du, None, None, None, None, dx, None
            dw = _dweight(da2, u2)
        return du, dw, dparams[2], None, None, dx, None


class _LinearReluDropout(torch.autograd.Function):
    """``dropout(relu(Linear(h)))`` (linear1 of the feed-forward block).  At the update's shapes the whole thing is one
    GEMM with a fused epilogue (``g2048_linear_relu_dropout_bf16``), otherwise a GEMM plus ``g2048_relu_dropout_fwd``.  Only
    the OUTPUT is saved (it is non-zero exactly where the unit was active and kept).  Backward: when the consumer
    (``_LinearAddLayerNorm`` of the same block, via ``link``) already delivered the masked gradient and the bias
    gradient, nothing is launched for the activation; else ``g2048_relu_dropout_bwd``."""

    @staticmethod
    def forward(ctx, h, weight, bias, wb, bb, p_drop, link=None, h_link=None, wt_packed=None):
        from ..g2048 import native as nv

        ctx.h_link, ctx.wt_packed = h_link, wt_packed  # (HLink: the producer of h may run this node's input-gradient GEMM itself)
        h2 = h.reshape(-1, h.shape[-1])
        with torch.autocast("cuda", enabled=False):
            if _stationary_ok(h2, wb) and bias.dtype == torch.float32:
                y = nv.linear_relu_dropout(h2, wb, bias.detach(), p_drop, *_seed_pair(h2, p_drop), want_mask=link is not None)
                if link is not None:
                    y, link.bits = y
                y = y.view(*h.shape[:-1], wb.shape[0])
            else:
                z = _hip_linear(h2, wb, bias)
                z = F.linear(h, wb, bb) if z is None else z.view(*h.shape[:-1], wb.shape[0])
                y = torch.empty_like(z)
                nv.relu_dropout_fwd(z, y, p_drop, *_seed_pair(z, p_drop))
        ctx.save_for_backward(h, wb, y)
        ctx.p_drop, ctx.link = p_drop, link
        ctx.params = (weight, bias)
        if link is not None:
            link.bias_param = bias
        return y

    @staticmethod
    def backward(ctx, dy):
        from ..g2048 import native as nv

        h, wb, y = ctx.saved_tensors
        link = ctx.link
        weight, bias = ctx.params
        sink = _sink_for(weight, bias)
        db_done = False
        if link is not None and link.masked:
            dz, db, db_done = dy.contiguous(), link.db, link.db_sunk
            link.db, link.masked, link.db_sunk = None, False, False
        else:
            dz = torch.empty_like(y)
            db = None if sink is not None else torch.empty(y.shape[-1], dtype=torch.float32, device=y.device)
            ws = nv.relu_dropout_bwd(dy.contiguous(), y, dz, db, ctx.p_drop)
            if sink is not None:
                sink.add(bias, ws, ws.shape[1], ws.shape[1], ws.shape[0])
                db_done = True
        with torch.autocast("cuda", enabled=False):
            dz2, h2 = dz.view(-1, dz.shape[-1]), h.reshape(-1, h.shape[-1])
            dh = None
            if ctx.needs_input_grad[0] and not (ctx.h_link is not None and ctx.h_link.offer(dz2, ctx.wt_packed)):
                dh = (dz2 @ wb).view(h.shape)
            if sink is not None:
                _sink_weight(sink, weight, dz2, h2)
                if not db_done:  # the link delivered a finished bias gradient: let the sink copy it into place
                    sink.add(bias, db.contiguous(), db.numel(), db.numel(), 1)
                return dh, None, None, None, None, None, None, None, None
            dw = _dweight(dz2, h2)
        return dh, dw, (None if db_done else db), None, None, None, None, None, None


def _fused_norm_ok(x: torch.Tensor, a) -> bool:
    return (x.is_cuda and x.dtype == torch.float32 and x.shape[-1] == 256 and torch.is_grad_enabled()
            and torch.is_autocast_enabled() and torch.get_autocast_dtype("cuda") == torch.bfloat16
            and (a is None or a.dtype == torch.bfloat16))


def _add_norm(x: torch.Tensor, a, norm: nn.LayerNorm, p: float, training: bool, h_link=None, pre=None):
    """(x + dropout(a), LayerNorm(x + dropout(a))); ``a`` None: (x, LayerNorm(x)).  ``h_link``: see ``HLink``; ``pre``: see ``LNPre``."""
    if _fused_norm_ok(x, a):
        if a is None and _residual_rows(x)[0] is not x:  # a copy would be made: keep the caller's x as the stream
            return x, _AddLayerNorm.apply(x, a, norm.weight, norm.bias, norm.eps, 0.0)[1]
        return _AddLayerNorm.apply(x, a, norm.weight, norm.bias, norm.eps, p if training else 0.0, h_link, pre)
    if a is not None:
        x = x + F.dropout(a, p, training)
    return x, F.layer_norm(x, (x.shape[-1],), norm.weight, norm.bias, norm.eps)


class LNPre:
    """Lets the node that PRODUCES the token matrix (``_EmbedBoards``) also run the first LayerNorm of the encoder on the rows it
    holds (``g2048_embed_ln_fwd``): it leaves h and the row statistics here, and the ``_AddLayerNorm`` node of that LayerNorm (``a``
    None) takes them instead of launching ``g2048_add_ln_fwd`` over the same 36 MB.  The backward is untouched (two nodes)."""

    __slots__ = ("norm", "x", "h", "stats")

    def __init__(self, norm: nn.LayerNorm):
        self.norm, self.x, self.h, self.stats = norm, None, None, None

    def usable(self) -> bool:
        n = self.norm
        return (n.weight is not None and n.bias is not None and n.weight.dtype == torch.float32 and n.bias.dtype == torch.float32
                and tuple(n.weight.shape) == (256,))


class _EmbedBoards(torch.autograd.Function):
    """Packed boards u8 [M, 16] -> token matrix f32 [M, 17, 256]: CLS row + dropout(embedding + positional code), one
    gather kernel (``g2048_embed_fwd``); the backward is a segmented sum by cell value (``g2048_embed_bwd``) instead of
    a one-hot^T x gradient GEMM whose reduction runs over every token of the minibatch.
    emb_weight: the bias-free input Linear's weight [256, 31]; pe f32 [16, 256]; cls [1, 1, 256]."""

    @staticmethod
    def forward(ctx, boards, emb_weight, pe, cls, p_drop, pre=None):
        from ..g2048 import native as nv

        boards = boards.contiguous()
        M = boards.shape[0]
        w = emb_weight.detach()
        # the f32 nn.Linear weight [256, 31] is read in place (no transposed copy per minibatch)
        wt = w if (w.dtype == torch.float32 and w.is_contiguous()) else w.float().t().contiguous()
        x0 = torch.empty((M, 17, 256), dtype=torch.float32, device=boards.device)
        seed = _seed_pair(x0, p_drop)
        ln = None
        if pre is not None and pre.usable():
            n = pre.norm
            pre.h = torch.empty((M, 17, 256), dtype=torch.bfloat16, device=boards.device)
            pre.stats = torch.empty((2, M * 17), dtype=torch.float32, device=boards.device)
            pre.x = x0
            ln = (n.weight.detach().contiguous(), n.bias.detach().contiguous(), n.eps, pre.h, pre.stats[0], pre.stats[1])
        nv.embed_fwd(boards, wt, pe, cls.detach().float().reshape(256).contiguous(), x0, p_drop, *seed, ln=ln)
        ctx.save_for_backward(boards)
        ctx.meta = (p_drop, seed, emb_weight.dtype, cls.dtype)
        ctx.params = (emb_weight, cls)
        return x0

    @staticmethod
    def backward(ctx, g):
        from ..g2048 import native as nv

        (boards,) = ctx.saved_tensors
        p_drop, seed, w_dtype, c_dtype = ctx.meta
        emb_weight, cls = ctx.params
        sink = _sink_for(emb_weight, cls) if (w_dtype == torch.float32 and c_dtype == torch.float32) else None
        if sink is not None:  # per-workgroup [32][256] images: 31 class rows -> the [256, 31] weight gradient, row 31 -> cls
            ws = nv.embed_bwd(boards, g.contiguous(), None, p_drop, *seed)
            sink.add(emb_weight, ws, ws.shape[1], 31 * 256, ws.shape[0], transpose_rows=31)
            sink.add(cls, ws[:, 31 * 256:], ws.shape[1], 256, ws.shape[0])
            return None, None, None, None, None, None
        out = torch.empty((32, 256), dtype=torch.float32, device=g.device)
        nv.embed_bwd(boards, g.contiguous(), out, p_drop, *seed)
        return None, out[:31].t().to(w_dtype), None, out[31].view(1, 1, 256).to(c_dtype), None, None


def _fused_attention_ok(t: torch.Tensor, S: int, head_dim: int) -> bool:
    return (t.is_cuda and t.dtype == torch.bfloat16 and S == 17 and head_dim == 32 and torch.is_grad_enabled()
            and t.requires_grad)


def _train_bf16(t: torch.Tensor, weight: torch.Tensor) -> bool:
    """The update path: HIP device, gradients wanted, bf16 autocast."""
    return (t.is_cuda and torch.is_grad_enabled() and weight.requires_grad and torch.is_autocast_enabled()
            and torch.get_autocast_dtype("cuda") == torch.bfloat16)


def _linear(x: torch.Tensor, weight: torch.Tensor, bias, wb=None, bb=None, relu: bool = False, h_link=None, wt_packed=None) -> torch.Tensor:
    """``F.linear`` (``relu``: followed by ReLU), through the update path's autograd nodes under bf16 autocast with gradients.
    ``h_link`` / ``wt_packed``: see ``HLink`` (the producer of x may take over the input-gradient GEMM)."""
    if _train_bf16(x, weight):
        if relu and _LinearRelu.ok(x, weight, bias, wb, bb):
            return _LinearRelu.apply(x, weight, bias, wb, bb)
        y = _LinearSplitK.apply(x, weight, bias, wb, bb, h_link, wt_packed)
    else:
        y = F.linear(x, weight, bias)
    return F.relu(y) if relu else y


# ---------------------------------------------------------------------------------------------------------------------
# the 2048-row tail of the update as one autograd node (csrc/g2048_tail.hip)
# ---------------------------------------------------------------------------------------------------------------------
TAIL_PARAM_ORDER = ("wo", "bo", "ln_g", "ln_b", "w1", "b1", "w2", "b2", "a1", "ab1", "a2", "ab2", "a3", "c1", "cb1", "c2", "cb2", "c3")
_TAIL_BIAS_OF = dict(wo="bo", w1="b1", w2="b2", a1="ab1", a2="ab2", c1="cb1", c2="cb2")
_TAIL_REAL_ROWS = dict(a3=4, c3=1)  # Linears whose dY^T buffer is padded to 32 rows


class TailBufferCache(dict):
    """Per-module cache of ``native.TailBuffers`` (device buffers + ctypes structs holding raw pointers): never copied or
    pickled with the module -- a copy of an agent starts with an empty cache."""

    def __deepcopy__(self, memo):
        return TailBufferCache()

    def __reduce__(self):
        return (TailBufferCache, ())


class TailPlan:
    """Everything ``_ClsTailHeads`` needs besides its tensor inputs: the f32 master parameters by field name (``TAIL_PARAM_ORDER``),
    their bf16 shadows (fragment-packed: the weight for the forward, its transpose for the backward) and the buffer cache of the owning
    module."""

    def __init__(self, params: dict, dense: dict, transposed: dict, buffer_cache: dict, eps: float, p_drop: float):
        # dense / transposed: FRAGMENT-PACKED bf16 copies of the weights / of their transposes (flat tensors), except a3 / c3 in
        # ``dense``, which the kernels read row-major
        self.params, self.dense, self.transposed, self.cache = params, dense, transposed, buffer_cache
        self.eps, self.p_drop = float(eps), float(p_drop)

    def buffers(self, M: int, device):
        from ..g2048 import native as nv

        key = (int(M), str(device))
        buf = self.cache.get(key)
        if buf is None or getattr(buf, "busy", False):
            # busy: a forward whose autograd graph is still alive and whose backward has not run owns the cached set (two forwards
            # before one backward): this call gets a private one instead of overwriting what that backward would read.  The cached
            # set itself is never replaced -- a captured hipGraph holds its raw pointers -- and it is released when its backward
            # runs OR when the graph that references it is dropped (``_ClsTailHeads.forward`` ties ``busy`` to the node's lifetime)
            fresh = nv.TailBuffers(M, device)
            if buf is None:
                self.cache[key] = fresh
            buf = fresh
        return buf

    def fwd_struct(self):
        from ..g2048 import native as nv

        t = {k: self.dense[k] for k in ("wo", "w1", "w2", "a1", "a2", "a3", "c1", "c2", "c3")}
        t.update({k: self.params[k].detach() for k in ("bo", "b1", "b2", "ab1", "ab2", "cb1", "cb2", "ln_g", "ln_b")})
        return nv.tail_weights(t), t

    def bwd_struct(self):
        from ..g2048 import native as nv

        t = {k + "T": self.transposed[k] for k in ("wo", "w1", "w2", "a1", "a2", "c1", "c2")}
        t.update(a3=self.dense["a3"], c3=self.dense["c3"], ln_g=self.params["ln_g"].detach())
        return nv.tail_weights_t(t), t


def _release_tail_buffers(buf_ref, gen: int):
    buf = buf_ref()
    if buf is not None and getattr(buf, "gen", 0) == gen:
        buf.busy = False


class _ClsTailHeads(torch.autograd.Function):
    """``(logits, values)`` from the CLS rows after the last layer's attention: out_proj + dropout + residual + LayerNorm, the
    feed-forward block + dropout + residual, actor and critic heads -- ``g2048_cls_tail_fwd`` forward, ``g2048_cls_tail_bwd`` +
    ``g2048_dweight_t`` backward: 3 launches where the unfused nodes needed ~55.  o bf16 [M, 1, 256] (attention output), x f32
    [M, 1, 256] (CLS rows of the residual stream, read in place through their row stride); ``params``: the f32 masters in
    ``TAIL_PARAM_ORDER`` (inputs only so that autograd can take their gradients when no ``GradSink`` is active).
    One minibatch in flight per module: the buffers between forward and backward are cached per minibatch size.  The returned
    ``logits`` / ``values`` are tensors OF that cached set: the next forward at the same minibatch size overwrites them (unless this
    one's backward is still pending, see ``TailPlan.buffers``) -- a caller that keeps them across minibatches clones them."""

    @staticmethod
    def forward(ctx, o, x, plan, *params):
        from ..g2048 import native as nv

        M = o.shape[0]
        o2 = o.reshape(M, 256).contiguous()
        xr, row_stride = _residual_rows(x)
        buf = plan.buffers(M, o.device)
        W, keep = plan.fwd_struct()
        seed = _seed_pair(o2, plan.p_drop)
        logits, values = nv.cls_tail_fwd(o2, xr.data_ptr(), row_stride, W, buf, plan.eps, plan.p_drop, *seed)
        buf.busy = True
        # a forward that is never followed by a backward (a diagnostic evaluate_actions with gradients on, an exception in the
        # loss) must not hold the cached set forever: when this node dies without having run, the set is free again.  The
        # generation keeps a late finaliser of an OLD node from releasing a set that a newer forward owns by then.
        buf.gen = gen = getattr(buf, "gen", 0) + 1
        weakref.finalize(ctx, _release_tail_buffers, weakref.ref(buf), gen)
        ctx.plan, ctx.buf, ctx.seed, ctx.keep = plan, buf, seed, (keep, xr, o2)
        ctx.x_shape, ctx.o_shape = tuple(x.shape), tuple(o.shape)
        ctx.set_materialize_grads(False)
        return logits, values.view(M, 1)

    @staticmethod
    def backward(ctx, dlogits, dvalues):
        from ..g2048 import native as nv

        plan, buf = ctx.plan, ctx.buf
        M = buf.M
        dev = buf.logits.device
        dlogits = torch.zeros((M, 4), dtype=torch.float32, device=dev) if dlogits is None else dlogits.float().contiguous()
        dvalues = torch.zeros(M, dtype=torch.float32, device=dev) if dvalues is None else dvalues.float().reshape(M).contiguous()
        WT, keep_t = plan.bwd_struct()
        d_o, dx = nv.cls_tail_bwd(dlogits, dvalues, WT, buf, plan.p_drop, *ctx.seed)
        jobs = []
        for k, (dy, xt, _n, _k, has_b) in buf.dw_spec.items():
            jobs.append((buf.grads[dy], buf.saved[xt], buf.dw[k], buf.db[k] if has_b else None))
        nv.dweight_t(jobs, buf.ld, buf.ld, buf.slices)
        buf.busy = False
        P = plan.params
        sink = _sink_for(*[P[k] for k in TAIL_PARAM_ORDER])
        S = buf.slices
        grads = {}
        for k, (_dy, _xt, n_pad, kk, has_b) in buf.dw_spec.items():
            n_real = _TAIL_REAL_ROWS.get(k, n_pad)
            if sink is not None:
                sink.add(P[k], buf.dw[k], n_pad * kk, n_real * kk, S)
                if has_b:
                    sink.add(P[_TAIL_BIAS_OF[k]], buf.db[k], n_pad, n_pad, S)
            else:
                grads[k] = buf.dw[k].sum(0)[:n_real].to(P[k].dtype)
                if has_b:
                    grads[_TAIL_BIAS_OF[k]] = buf.db[k].sum(0).to(P[_TAIL_BIAS_OF[k]].dtype)
        lnp = buf.grads["ln_partial"]
        if sink is not None:
            sink.add(P["ln_g"], lnp, 512, 256, buf.blocks)
            sink.add(P["ln_b"], lnp[:, 256:], 512, 256, buf.blocks)
        else:
            s = lnp.sum(0)
            grads["ln_g"], grads["ln_b"] = s[:256].to(P["ln_g"].dtype), s[256:].to(P["ln_b"].dtype)
        if sink is not None:
            sink.early_complete()  # everything this node produces is registered: an early all-reduce bucket may go now
        # (views of the cached buffers: the nodes that consume them run in this same backward pass)
        do_out = d_o.view(ctx.o_shape) if ctx.needs_input_grad[0] else None
        dx_out = dx.view(ctx.x_shape) if ctx.needs_input_grad[1] else None
        if sink is not None:
            return (do_out, dx_out, None) + (None,) * len(TAIL_PARAM_ORDER)
        return (do_out, dx_out, None) + tuple(grads[k] for k in TAIL_PARAM_ORDER)

