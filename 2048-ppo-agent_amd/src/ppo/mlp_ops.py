"""The MLP policy's update path (BASELINE.json configs[1]) as ONE autograd node over a handful of HIP launches (csrc/g2048_mlp.hip).

At minibatch 2048 this policy's forward + backward was ~45 nodes of the replayed hipGraph, every one of them a few microseconds of work
under a ~4.5 us node floor.  Here: 5 launches forward (trunk_in on packed boards; trunk_hidden; both heads' first layers; both second
layers; both output layers) and 5 backward (output layers + their weight-gradient partials; second layers; first layers into the shared
trunk gradient; trunk_hidden; the bias gradient of trunk_in), with every 512-wide weight gradient - trunk_in's included, its input being
the one-hot matrix the forward left behind - in the GradSink's one grouped ``g2048_dweight_jobs`` launch.  The reference has no MLP
policy; the heads are the reference's (src/ppo/ppo_agent.py:72-87).
"""
import torch
import torch.nn as nn

from .hip_ops import Bf16Shadow, _dweight_parts_config, _sink_for

# parameter order of the node (and of its shadow): trunk_in w, b | trunk_hidden w, b | actor.0 w, b | critic.0 w, b | actor.2 w, b |
# critic.2 w, b | actor.4 w | critic.4 w  (the two output weights adjacent: the kernels read them as one [5][512] matrix)
MLP_PARAMS = ("win", "bin", "wh", "bh", "a1w", "a1b", "c1w", "c1b", "a2w", "a2b", "c2w", "c2b", "a3w", "c3w")
_T = dict(win=0, wh=2, a1w=4, c1w=6, a2w=8, c2w=10)  # weights whose [in][out] copy the kernels multiply with


class MLPPlan:
    """Parameters and bf16 shadows of an ``MLPAgent`` with the standard shapes (16 x 31 one-hot -> 512 -> 512, heads 512 -> 512 -> 512 ->
    4 / 1) for ``_MLPUpdate``."""

    def __init__(self, agent):
        a, c = list(agent.actor), list(agent.critic)
        ps = dict(win=agent.trunk_in.weight, bin=agent.trunk_in.bias, wh=agent.trunk_hidden.weight, bh=agent.trunk_hidden.bias,
                  a1w=a[0].weight, a1b=a[0].bias, c1w=c[0].weight, c1b=c[0].bias, a2w=a[2].weight, a2b=a[2].bias,
                  c2w=c[2].weight, c2b=c[2].bias, a3w=a[4].weight, c3w=c[4].weight)
        self.params = [ps[k] for k in MLP_PARAMS]
        self.shadow = Bf16Shadow(self.params, transposed=sorted(_T.values()))

    @staticmethod
    def supports(agent) -> bool:
        def head(h, n_out):
            m = list(h)
            return (len(m) == 5 and isinstance(m[0], nn.Linear) and type(m[1]) is nn.ReLU and isinstance(m[2], nn.Linear)
                    and type(m[3]) is nn.ReLU and isinstance(m[4], nn.Linear) and m[4].bias is None and m[0].bias is not None
                    and m[2].bias is not None and tuple(m[0].weight.shape) == (512, 512) and tuple(m[2].weight.shape) == (512, 512)
                    and tuple(m[4].weight.shape) == (n_out, 512))

        return (tuple(agent.trunk_in.weight.shape) == (512, 496) and tuple(agent.trunk_hidden.weight.shape) == (512, 512)
                and agent.trunk_in.bias is not None and agent.trunk_hidden.bias is not None and head(agent.actor, 4)
                and head(agent.critic, 1) and all(p.dtype == torch.float32 for p in agent.parameters()))

    def views(self, refresh: bool = True):
        """(dense bf16 views by name, transposed bf16 copies by name, w3 = [actor.4 ; critic.4] as one [5, 512] view).  ``refresh``
        False: the buffers as they are (a rollout forward being captured: whoever replays that graph refreshes first)."""
        v = self.shadow() if (refresh or self.shadow.views is None) else self.shadow.views
        dense = dict(zip(MLP_PARAMS, v))
        tv = {k: self.shadow.tviews[i] for k, i in _T.items()}
        a3 = dense["a3w"]
        # actor.4.weight (4 x 512 = 2048 elements, a multiple of the shadow's 64-element alignment) is followed directly by critic.4.weight
        w3 = torch.as_strided(a3, (5, 512), (512, 1))
        assert dense["c3w"].data_ptr() == a3.data_ptr() + 2 * 4 * 512, "shadow layout: critic.4.weight must follow actor.4.weight"
        return dense, tv, w3


class _MLPUpdate(torch.autograd.Function):
    """``(logits [M, 4], values [M, 1])`` of the MLP policy from packed boards u8 [M, 16]; ``params`` (the f32 masters in ``MLP_PARAMS``
    order) are inputs only so that autograd can take their gradients when no ``GradSink`` is active."""

    @staticmethod
    def forward(ctx, boards, plan, *params):
        from ..g2048 import native as nv

        M = boards.shape[0]
        dev, bf = boards.device, torch.bfloat16
        dense, tv, w3 = plan.views()
        P = dict(zip(MLP_PARAMS, params))
        new = lambda *s: torch.empty(s, dtype=bf, device=dev)
        t1, onehot, t2, h1, h2 = new(M, 512), new(M, 512), new(M, 512), new(M, 1024), new(M, 1024)
        nv.mlp_embed_fwd(boards.contiguous(), tv["win"], P["bin"].detach(), t1, onehot)
        nv.gemm_jobs([dict(segs=[(t1, dense["wh"])], bias=P["bh"].detach(), relu=True, y=t2)], M)
        nv.gemm_jobs([dict(segs=[(t2, dense["a1w"])], bias=P["a1b"].detach(), relu=True, y=h1[:, :512]),
                      dict(segs=[(t2, dense["c1w"])], bias=P["c1b"].detach(), relu=True, y=h1[:, 512:])], M)
        nv.gemm_jobs([dict(segs=[(h1[:, :512], dense["a2w"])], bias=P["a2b"].detach(), relu=True, y=h2[:, :512]),
                      dict(segs=[(h1[:, 512:], dense["c2w"])], bias=P["c2b"].detach(), relu=True, y=h2[:, 512:])], M)
        logits = torch.empty((M, 4), dtype=torch.float32, device=dev)
        values = torch.empty(M, dtype=torch.float32, device=dev)
        nv.mlp_out_fwd(h2, w3, logits, values)
        ctx.plan, ctx.keep = plan, (t1, onehot, t2, h1, h2, tv, w3)
        ctx.set_materialize_grads(False)
        return logits, values.view(M, 1)

    @staticmethod
    def backward(ctx, dlogits, dvalues):
        from ..g2048 import native as nv

        t1, onehot, t2, h1, h2, tv, w3 = ctx.keep
        M, dev, bf = t1.shape[0], t1.device, torch.bfloat16
        dlogits = torch.zeros((M, 4), dtype=torch.float32, device=dev) if dlogits is None else dlogits.float().contiguous()
        dvalues = torch.zeros(M, dtype=torch.float32, device=dev) if dvalues is None else dvalues.float().reshape(M).contiguous()
        new = lambda *s: torch.empty(s, dtype=bf, device=dev)
        d2, d1, dt2, dt1 = new(M, 1024), new(M, 1024), new(M, 512), new(M, 512)
        ws3 = nv.mlp_out_bwd(dlogits, dvalues, h2, w3, d2)
        nv.gemm_jobs([dict(segs=[(d2[:, :512], tv["a2w"])], act=h1[:, :512], y=d1[:, :512]),
                      dict(segs=[(d2[:, 512:], tv["c2w"])], act=h1[:, 512:], y=d1[:, 512:])], M)
        nv.gemm_jobs([dict(segs=[(d1[:, :512], tv["a1w"]), (d1[:, 512:], tv["c1w"])], act=t2, y=dt2)], M)
        nv.gemm_jobs([dict(segs=[(dt2, tv["wh"])], act=t1, y=dt1)], M)
        P = dict(zip(MLP_PARAMS, ctx.plan.params))
        products = (("a2w", "a2b", d2[:, :512], h1[:, :512]), ("c2w", "c2b", d2[:, 512:], h1[:, 512:]),
                    ("a1w", "a1b", d1[:, :512], t2), ("c1w", "c1b", d1[:, 512:], t2), ("wh", "bh", dt2, t1))
        sink = _sink_for(*ctx.plan.params)
        if sink is not None and M % 512 == 0:
            slices, dtype = _dweight_parts_config()
            for wk, bk, dy, x in products:
                sink.add_dweight(P[wk], P[bk], dy, x, slices, parts_dtype=dtype)
            # trunk_in: dW^T [496][512] = onehot^T dt1 as one more product of the grouped launch (N = 512 padded classes, K = 512), stored
            # transposed by the reduction; its bias gradient = the column sums of dt1 = row 496 of the same product (the one-hot matrix
            # carries a column of ones there)
            parts = torch.empty((slices, 512, 512), dtype=dtype, device=dev)
            sink.dw_jobs.append((onehot, dt1, parts, None))
            sink.add(P["win"], parts, 512 * 512, 496 * 512, slices, transpose_rows=496)
            sink.add(P["bin"], parts.view(slices, 512 * 512)[:, 496 * 512:], 512 * 512, 512, slices)
            rows = ws3.shape[0]
            flat3 = ws3.view(rows, 5 * 512)
            sink.add(P["a3w"], flat3, 5 * 512, 4 * 512, rows)
            sink.add(P["c3w"], flat3[:, 4 * 512:], 5 * 512, 512, rows)
            return (None, None) + (None,) * len(MLP_PARAMS)
        # no sink (plain autograd): the same gradients with PyTorch's operators
        g = {}
        for wk, bk, dy, x in products:
            g[wk] = (dy.float().t() @ x.float())
            g[bk] = dy.float().sum(0)
        g["win"] = (dt1.float().t() @ onehot.float())[:, :496].contiguous()  # (column 496 of the one-hot matrix: ones, the bias gradient)
        g["bin"] = dt1.float().sum(0)
        s3 = ws3.sum(0)
        g["a3w"], g["c3w"] = s3[:4].contiguous(), s3[4:5].contiguous()
        return (None, None) + tuple(g[k].to(P[k].dtype) for k in MLP_PARAMS)


def forward_nograd(boards: torch.Tensor, plan: MLPPlan, refresh: bool = True):
    """The same five launches as ``_MLPUpdate.forward`` without anything kept for a backward: the bf16 rollout forward of the MLP
    policy on packed boards -> (logits f32 [M, 4], values f32 [M, 1])."""
    from ..g2048 import native as nv

    M = boards.shape[0]
    dev, bf = boards.device, torch.bfloat16
    dense, tv, w3 = plan.views(refresh)
    P = dict(zip(MLP_PARAMS, plan.params))
    new = lambda *s: torch.empty(s, dtype=bf, device=dev)
    t1, t2, h1, h2 = new(M, 512), new(M, 512), new(M, 1024), new(M, 1024)
    nv.mlp_embed_fwd(boards.contiguous(), tv["win"], P["bin"].detach(), t1, None)
    nv.gemm_jobs([dict(segs=[(t1, dense["wh"])], bias=P["bh"].detach(), relu=True, y=t2)], M)
    nv.gemm_jobs([dict(segs=[(t2, dense["a1w"])], bias=P["a1b"].detach(), relu=True, y=h1[:, :512]),
                  dict(segs=[(t2, dense["c1w"])], bias=P["c1b"].detach(), relu=True, y=h1[:, 512:])], M)
    nv.gemm_jobs([dict(segs=[(h1[:, :512], dense["a2w"])], bias=P["a2b"].detach(), relu=True, y=h2[:, :512]),
                  dict(segs=[(h1[:, 512:], dense["c2w"])], bias=P["c2b"].detach(), relu=True, y=h2[:, 512:])], M)
    logits = torch.empty((M, 4), dtype=torch.float32, device=dev)
    values = torch.empty(M, dtype=torch.float32, device=dev)
    nv.mlp_out_fwd(h2, w3, logits, values)
    return logits, values.view(M, 1)
