"""Optimizer + LR schedule factory (API of the reference src/optim/configure_optimizers.py:16-260)."""
import logging

import torch
from torch.optim.lr_scheduler import ConstantLR, CosineAnnealingLR, LinearLR, SequentialLR

from .lamb import Lamb

logger = logging.getLogger(__name__)

_OPTIMIZERS = {"adam": torch.optim.Adam, "adamw": torch.optim.AdamW, "lamb": Lamb}


def split_decay_groups(model: torch.nn.Module, blacklist_weight_modules):
    """Names of parameters that do / do not receive weight decay.

    No decay: every ``*bias`` and every ``*weight`` whose full name contains one of the blacklist strings
    (e.g. "norm", "embedding").  Everything else decays -- including parameters that are neither
    (``transformer.cls_token``).  Mirrors reference configure_optimizers.py:160-180.
    """
    decay, no_decay = [], []
    for name, _ in model.named_parameters():
        leaf = name.rsplit(".", 1)[-1]
        if leaf.endswith("bias"):
            no_decay.append(name)
        elif leaf.endswith("weight") and any(tag in name for tag in blacklist_weight_modules):
            no_decay.append(name)
        else:
            decay.append(name)
    return sorted(decay), sorted(no_decay)


def _warmup(optimizer, n_warm: int, kind: str):
    if kind == "linear":
        return LinearLR(optimizer, start_factor=1e-3, end_factor=1, total_iters=n_warm)
    if kind == "constant":
        return ConstantLR(optimizer, factor=1, total_iters=n_warm)
    raise TypeError(f"Invalid schedule type: {kind}")


def _main(optimizer, n_main: int, kind: str):
    if kind == "linear":
        return LinearLR(optimizer, start_factor=1, end_factor=1e-4, total_iters=n_main)
    if kind == "cosine":
        return CosineAnnealingLR(optimizer, T_max=n_main)
    if kind == "constant":
        return ConstantLR(optimizer, factor=1)
    return None  # the reference leaves this undefined; SequentialLR then fails below with a clear error


def configure_bert_optimizers(model: torch.nn.Module, opt_name: str, max_lr: float, betas: tuple, eps: float,
                              weight_decay: float, steps: int, warmup_steps_ratio: float, scheduler_names: list,
                              blacklist_weight_modules: list = []) -> dict:
    """-> ``{"optimizer": opt, "lr_scheduler": {"scheduler": SequentialLR[warmup, main], "interval": "step"}}``.

    ``opt_name`` in {"adam", "adamw", "lamb"}; ``scheduler_names = [warmup_kind, main_kind]`` with warmup in
    {"linear", "constant"} and main in {"linear", "cosine", "constant"}; the switch happens after
    ``int(warmup_steps_ratio * steps)`` scheduler steps.
    """
    if opt_name not in _OPTIMIZERS:
        raise TypeError(f"Invalid optimizer name: {opt_name}")
    opt_cls = _OPTIMIZERS[opt_name]
    betas = tuple(betas)
    extra = {}
    if opt_name in ("adam", "adamw") and all(p.is_cuda for p in model.parameters()):
        extra["fused"] = True  # one multi-tensor kernel per step, and GradScaler hands it found_inf without a sync
    if blacklist_weight_modules:
        decay, no_decay = split_decay_groups(model, list(blacklist_weight_modules))
        params = dict(model.named_parameters())
        logger.info("Weight decay will not be applied to: %s", no_decay)
        groups = [{"params": [params[n] for n in decay], "weight_decay": weight_decay},
                  {"params": [params[n] for n in no_decay], "weight_decay": 0.0}]
        optimizer = opt_cls(groups, lr=max_lr, betas=betas, eps=eps, **extra)
    else:
        optimizer = opt_cls(model.parameters(), lr=max_lr, betas=betas, eps=eps, weight_decay=weight_decay, **extra)
    n_warm = int(warmup_steps_ratio * steps)
    warm = _warmup(optimizer, n_warm, scheduler_names[0])
    main = _main(optimizer, int(steps - warmup_steps_ratio * steps), scheduler_names[1])
    if main is None:
        raise TypeError(f"Invalid schedule type: {scheduler_names[1]}")
    scheduler = SequentialLR(optimizer, [warm, main], milestones=[n_warm])
    return {"optimizer": optimizer, "lr_scheduler": {"scheduler": scheduler, "interval": "step"}}
