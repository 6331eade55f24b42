"""The PPO update's optimiser step in three HIP launches (``g2048_opt_step``, csrc/g2048_optim.hip).

The reference runs ``scaler.unscale_(opt); clip_grad_norm_(params, max_norm); scaler.step(opt); scaler.update()``
(src/ppo/ppo_trainer.py:413-434) with ``opt = torch.optim.AdamW`` over two weight-decay groups
(src/optim/configure_optimizers.py:16-127); PyTorch executes that as about a dozen multi-tensor launches per minibatch.
``FlatAdamWStep`` keeps the SAME ``torch.optim.AdamW`` object (its ``param_groups`` drive the LR schedule, its
``state_dict()`` is what checkpoints store) but owns the storage behind it: gradients, ``exp_avg`` and ``exp_avg_sq`` of all
parameters live in three flat f32 buffers, ``optimizer.state[p]`` holds views into them, and ``step()`` is one call into the
kernel sequence.  The flat gradient buffer doubles as the all-reduce bucket of a multi-GPU run.
"""
from __future__ import annotations

import torch

from ..g2048 import native as nv


class FlatAdamWStep:
    @staticmethod
    def supports(optimizer, device) -> bool:
        """A plain AdamW over contiguous f32 parameters on the HIP device, at most ``OPT_MAX_GROUPS`` groups."""
        if type(optimizer) is not torch.optim.AdamW or torch.device(device).type != "cuda":
            return False
        if len(optimizer.param_groups) > nv.OPT_MAX_GROUPS:
            return False
        for g in optimizer.param_groups:
            if g.get("amsgrad") or g.get("maximize") or isinstance(g["lr"], torch.Tensor):
                return False
            for p in g["params"]:
                if not p.is_cuda or p.dtype != torch.float32 or not p.is_contiguous():
                    return False
        return True

    def __init__(self, optimizer: torch.optim.AdamW, device, first=()):
        """``first``: parameters whose slices come FIRST in the flat buffers, in optimiser order (the trainer's early all-reduce
        bucket: gradients that are final before the rest of the backward has run); ``self.n_first`` = elements they occupy."""
        if not self.supports(optimizer, device):
            raise ValueError("FlatAdamWStep needs a torch.optim.AdamW over contiguous f32 parameters on the HIP device")
        self.optimizer, self.device = optimizer, torch.device(device)
        entries = [(p, gi) for gi, g in enumerate(optimizer.param_groups) for p in g["params"] if p.requires_grad]
        first_ids = {id(p) for p in first}
        entries = [e for e in entries if id(e[0]) in first_ids] + [e for e in entries if id(e[0]) not in first_ids]
        self.params, self.group_of = [e[0] for e in entries], [e[1] for e in entries]
        n_first_params = sum(1 for p in self.params if id(p) in first_ids)
        self.offsets, n = [], 0
        for p in self.params:
            self.offsets.append(n)
            n += (p.numel() + 3) // 4 * 4  # every tensor starts on a 16-byte boundary of the flat buffers
        self.numel = n
        self.n_first = self.offsets[n_first_params] if n_first_params < len(self.params) else n
        if n_first_params == 0:
            self.n_first = 0
        z = lambda: torch.zeros(n, dtype=torch.float32, device=self.device)
        self.grad, self.exp_avg, self.exp_avg_sq = z(), z(), z()
        self.steps = torch.zeros(len(self.params), dtype=torch.float32, device=self.device)
        view = lambda flat: [flat[o:o + p.numel()].view_as(p) for o, p in zip(self.offsets, self.params)]
        self.grad_views, self.m_views, self.v_views = view(self.grad), view(self.exp_avg), view(self.exp_avg_sq)
        self.info = torch.zeros(2, dtype=torch.float32, device=self.device)  # [grad norm before clipping, found_inf]
        self._table_key, self._table, self._n_chunks, self._ws = None, None, 0, None
        self._shadows = []  # Bf16Shadow-like objects this step keeps up to date (see adopt_shadows)
        self.adopt_state()

    # ------------------------------------------------------------------ state <-> torch.optim
    def adopt_state(self):
        """Take over whatever ``optimizer.state`` holds (fresh: nothing; after ``load_state_dict``: loaded tensors) and
        re-point it at the flat buffers."""
        st = self.optimizer.state
        for i, p in enumerate(self.params):
            s = st.get(p, {})
            m, v, t = s.get("exp_avg"), s.get("exp_avg_sq"), s.get("step")
            with torch.no_grad():
                if m is not None and m.data_ptr() != self.m_views[i].data_ptr():
                    self.m_views[i].copy_(m)
                elif m is None:
                    self.m_views[i].zero_()
                if v is not None and v.data_ptr() != self.v_views[i].data_ptr():
                    self.v_views[i].copy_(v)
                elif v is None:
                    self.v_views[i].zero_()
                if t is None:
                    self.steps[i] = 0.0
                elif not (torch.is_tensor(t) and t.data_ptr() == self.steps[i].data_ptr()):
                    self.steps[i] = float(t)
            st[p] = {"step": self.steps[i], "exp_avg": self.m_views[i], "exp_avg_sq": self.v_views[i]}
        # the kernel derives the bias corrections of every group from steps[0]: a loaded state whose parameters disagree on the
        # step count (parameters frozen for part of a run under a plain AdamW) cannot be continued by this step
        if self.steps.numel() and float(self.steps.min()) != float(self.steps.max()):
            raise ValueError("FlatAdamWStep: the optimizer state holds different step counts per parameter "
                             f"({float(self.steps.min()):.0f}..{float(self.steps.max()):.0f}); g2048_opt_step keeps one count for all")

    def reset_state(self):
        """A fresh optimiser state (zero moments, step 0) in the flat buffers ``optimizer.state`` already points at."""
        with torch.no_grad():
            self.exp_avg.zero_()
            self.exp_avg_sq.zero_()
            self.steps.zero_()

    def adopt_shadows(self, shadows):
        """Keep bf16 shadow copies of parameters up to date from inside the optimiser kernel.  ``shadows``: objects with
        ``params`` (list), ``views`` (bf16 tensors, same order), ``tviews`` ({index: transposed bf16 tensor}), optionally ``pviews`` /
        ``ptviews`` ({index: fragment-packed bf16 copy of the tensor / of its transpose}), a ``key``
        attribute and ``current_key()`` (what ``hip_ops.Bf16Shadow`` offers); only those whose parameters all belong to this
        optimiser and whose buffers exist are taken.  Each adopted shadow is refreshed once here (a copy) and from then on by
        ``step()``; its ``maintainer`` is set so that it stops copying on its own while its key is current."""
        mine = {id(p) for p in self.params}
        for sh in shadows:
            if sh in self._shadows or getattr(sh, "views", None) is None or not all(id(p) in mine for p in sh.params):
                continue
            # ONE maintained shadow per parameter: the chunk table holds one set of shadow pointers per parameter, so a second shadow
            # over a parameter that an adopted one already covers would be marked maintained and never rewritten (stale weights in
            # whatever reads it).  It stays un-adopted instead and keeps refreshing itself by copy when its key is stale.
            taken = {id(p) for other in self._shadows for p in other.params}
            if any(id(p) in taken for p in sh.params):
                continue
            sh.key = None
            sh()  # one eager refresh: from here on the kernel writes the same values in place
            sh.maintainer = self
            self._shadows.append(sh)
            self._table_key = None  # rebuild the chunk table with the shadow pointers
        return len(self._shadows)

    def _chunk_table(self, skip=()):
        key = tuple(p.data_ptr() for p in self.params) + tuple(id(sh) for sh in self._shadows) + ("skip",) + tuple(skip)
        if key != self._table_key:  # a parameter was re-allocated (module.to(), load with assign=True, ...) or shadows joined
            shadow_of = {}
            for sh in self._shadows:
                for i, p in enumerate(sh.params):
                    shadow_of[id(p)] = (sh.views[i], sh.tviews.get(i), getattr(sh, "pviews", {}).get(i),
                                        getattr(sh, "ptviews", {}).get(i))
            keep = [i for i in range(len(self.params)) if i not in set(skip)]
            self._table = nv.opt_chunk_table([self.params[i] for i in keep], [self.offsets[i] for i in keep],
                                             [self.group_of[i] for i in keep], self.device, shadow_of)
            self._n_chunks = self._table.numel() // nv.OPT_CHUNK_BYTES
            self._ws = nv.opt_workspace(self._n_chunks, self.device)
            self._table_key = key
        return self._table

    # ------------------------------------------------------------------ the step
    def step(self, max_grad_norm, scaler=None, skip=()):
        """Gradients are read from ``self.grad`` (bind ``p.grad`` to ``grad_views`` or copy into them first).
        ``skip``: indices (into ``self.params``) of parameters that received NO gradient this step: like torch.optim.AdamW they are
        left alone entirely -- no weight decay, no moment update, out of the gradient norm (their slice of ``self.grad`` is not
        read).  Their step counter still advances with the others (the kernel keeps one count for all parameters)."""
        table = self._chunk_table(tuple(sorted(skip)))
        groups = []
        for g in self.optimizer.param_groups:
            b1, b2 = g["betas"]
            groups.append((g["lr"], b1, b2, g["eps"], g["weight_decay"]))
        scale = tracker = None
        growth, backoff, interval = 2.0, 0.5, 2000
        if scaler is not None and scaler.is_enabled():
            if scaler._scale is None:
                scaler._lazy_init_scale_growth_tracker(self.device)
            scale, tracker = scaler._scale, scaler._growth_tracker
            growth, backoff, interval = scaler.get_growth_factor(), scaler.get_backoff_factor(), scaler.get_growth_interval()
        # shadows that were current BEFORE this step stay current through it (the kernel rewrites them with the parameters, or
        # leaves both untouched on a skipped step); one that was stale (a parameter changed behind the optimiser's back) stays
        # stale, so that its next use copies for real
        was_current = [sh.key == sh.current_key() for sh in self._shadows]
        nv.opt_step(table, self._n_chunks, self.grad, self.exp_avg, self.exp_avg_sq, groups, max_grad_norm, self.steps, scale,
                    tracker, growth, backoff, interval, self._ws, self.info)
        # the kernel wrote the parameters through raw pointers: tell autograd's version counters, which is what everything
        # that caches derived weights keys on (the bf16 shadows of the update path, the packed weights of the fused rollout
        # encoder, torch's own saved-tensor checks).  No launch.
        torch.autograd.graph.increment_version(self.params)
        for sh, ok in zip(self._shadows, was_current):  # rewritten together with the parameters: the key follows the new versions
            sh.key = sh.current_key() if ok else None
        self.optimizer._opt_called = True  # the LR scheduler checks that a step preceded scheduler.step()
