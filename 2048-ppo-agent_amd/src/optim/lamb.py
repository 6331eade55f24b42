"""LAMB optimizer (constructor-compatible with the reference src/optim/lamb.py:56-209).

Layer-wise Adaptive Moments (You et al., "Large Batch Optimization for Deep Learning", arXiv:1904.00962) in
the NVLAMB flavour the reference uses: global gradient-norm pre-clipping, bias-corrected Adam direction,
decoupled-into-the-update weight decay, then a per-tensor trust ratio ||w|| / ||update||.
Written on ``torch._foreach_*`` so a 60-tensor model costs a handful of launches per step.
"""
import math

import torch
from torch.optim import Optimizer


class Lamb(Optimizer):
    def __init__(self, params, lr=1e-3, bias_correction=True, betas=(0.9, 0.999), eps=1e-6, weight_decay=0.01,
                 grad_averaging=True, max_grad_norm=1.0, trust_clip=False, always_adapt=False):
        defaults = dict(lr=lr, bias_correction=bias_correction, betas=betas, eps=eps, weight_decay=weight_decay,
                        grad_averaging=grad_averaging, max_grad_norm=max_grad_norm, trust_clip=trust_clip,
                        always_adapt=always_adapt)
        super().__init__(params, defaults)

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        all_grads = [p.grad for g in self.param_groups for p in g["params"] if p.grad is not None]
        if not all_grads:
            return loss
        if any(gr.is_sparse for gr in all_grads):
            raise RuntimeError("Lamb does not support sparse gradients")
        # one global norm over every gradient; scale so that it is at most max_grad_norm
        gnorm = torch.linalg.vector_norm(torch.stack(torch._foreach_norm(all_grads)))
        max_norm = self.defaults["max_grad_norm"]
        clip = (gnorm / max_norm).clamp(min=1.0) if max_norm is not None else torch.ones_like(gnorm)

        for group in self.param_groups:
            params = [p for p in group["params"] if p.grad is not None]
            if not params:
                continue
            beta1, beta2 = group["betas"]
            beta3 = 1 - beta1 if group["grad_averaging"] else 1.0
            group["step"] = group.get("step", 0) + 1
            if group["bias_correction"]:
                bc1, bc2 = 1 - beta1 ** group["step"], 1 - beta2 ** group["step"]
            else:
                bc1 = bc2 = 1.0
            grads = torch._foreach_div([p.grad for p in params], clip)
            for p in params:
                st = self.state[p]
                if not st:
                    st["exp_avg"] = torch.zeros_like(p)
                    st["exp_avg_sq"] = torch.zeros_like(p)
            m = [self.state[p]["exp_avg"] for p in params]
            v = [self.state[p]["exp_avg_sq"] for p in params]
            torch._foreach_mul_(m, beta1)
            torch._foreach_add_(m, grads, alpha=beta3)
            torch._foreach_mul_(v, beta2)
            torch._foreach_addcmul_(v, grads, grads, value=1 - beta2)
            denom = torch._foreach_sqrt(v)
            torch._foreach_div_(denom, math.sqrt(bc2))
            torch._foreach_add_(denom, group["eps"])
            update = torch._foreach_div(m, denom)
            torch._foreach_div_(update, bc1)
            wd = group["weight_decay"]
            if wd != 0:
                torch._foreach_add_(update, params, alpha=wd)
            if wd != 0 or group["always_adapt"]:
                w_norm = torch._foreach_norm(params)
                u_norm = torch._foreach_norm(update)
                ratios = []
                for wn, un in zip(w_norm, u_norm):
                    r = torch.where((wn > 0) & (un > 0), wn / un, torch.ones_like(wn))
                    ratios.append(r.clamp(max=1.0) if group["trust_clip"] else r)
                torch._foreach_mul_(update, ratios)
            torch._foreach_add_(params, update, alpha=-group["lr"])
        return loss
