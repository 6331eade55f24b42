"""The private / semi-private PyTorch interfaces the device update path leans on, checked in one place.

They are pinned by the image (torch 2.10); none of them is covered by PyTorch's compatibility promise.  ``check()`` verifies
that each exists with the call shape used here and raises ONE error naming the torch version and everything that is
missing, instead of an AttributeError deep inside a captured update.  ``PPOTrainer`` calls it when it takes the device
path; ``tests/test_host_logic.py`` calls it on every CPU run, so a torch upgrade fails loudly in CI first.

| interface | used by | for |
|---|---|---|
| ``GradScaler._scale``, ``._growth_tracker``, ``._lazy_init_scale_growth_tracker(device)`` | optim/flat_step.py, ppo_trainer.py | the device-resident loss scale read by ``g2048_ppo_loss`` / ``g2048_opt_step`` |
| ``torch.autograd.graph.increment_version(tensors)`` | optim/flat_step.py, ppo_trainer.py | parameters written through raw pointers / by fused optimisers |
| ``torch._addmm_activation(bias, x, w_t, use_gelu=False)`` | hip_ops._LinearRelu, MLPAgent.forward | bias + ReLU in the GEMM epilogue |
| ``torch._foreach_copy_(dsts, srcs)`` | hip_ops.Bf16Shadow, ppo_trainer._collect_grads | one multi-tensor copy |
"""
import inspect

import torch


class TorchInterfaceError(RuntimeError):
    pass


def _has_params(fn, names) -> bool:
    try:
        sig = inspect.signature(fn)
    except (TypeError, ValueError):  # builtins without an introspectable signature: presence is all we can check
        return True
    return all(n in sig.parameters for n in names)


def problems() -> list:
    """Human-readable list of what is missing (empty: all interfaces present)."""
    out = []
    from torch.amp import GradScaler

    scaler = GradScaler("cpu", enabled=True)
    for attr in ("_scale", "_growth_tracker"):
        if not hasattr(scaler, attr):
            out.append(f"torch.amp.GradScaler.{attr} (attribute)")
    lazy = getattr(GradScaler, "_lazy_init_scale_growth_tracker", None)
    if lazy is None or not _has_params(lazy, ("dev",)):
        out.append("torch.amp.GradScaler._lazy_init_scale_growth_tracker(self, dev)")
    else:
        try:
            scaler._lazy_init_scale_growth_tracker(torch.device("cpu"))
            if not (torch.is_tensor(scaler._scale) and scaler._scale.numel() == 1 and scaler._scale.dtype == torch.float32
                    and torch.is_tensor(scaler._growth_tracker) and scaler._growth_tracker.dtype == torch.int32):
                out.append("GradScaler._scale / _growth_tracker are no longer a float32[1] / int32[1] tensor pair")
        except Exception as e:  # noqa: BLE001
            out.append(f"GradScaler._lazy_init_scale_growth_tracker(cpu) raised {e!r}")
    for name in ("get_growth_factor", "get_backoff_factor", "get_growth_interval"):
        if not callable(getattr(scaler, name, None)):
            out.append(f"torch.amp.GradScaler.{name}()")
    inc = getattr(getattr(torch.autograd, "graph", None), "increment_version", None)
    if inc is None:
        out.append("torch.autograd.graph.increment_version(tensors)")
    else:
        try:
            t = torch.zeros(2)
            v = t._version
            inc([t])  # must accept a list of tensors
            if t._version != v + 1:
                out.append("torch.autograd.graph.increment_version no longer bumps Tensor._version by one")
        except Exception as e:  # noqa: BLE001
            out.append(f"torch.autograd.graph.increment_version([tensor]) raised {e!r}")
    if not hasattr(torch, "_addmm_activation"):
        out.append("torch._addmm_activation(bias, mat1, mat2)")
    else:
        try:
            y = torch._addmm_activation(torch.ones(3), -torch.ones(2, 4), torch.ones(4, 3))
            if not torch.equal(y, torch.zeros(2, 3)):  # relu(1 - 4) = 0: the default activation must still be ReLU
                out.append("torch._addmm_activation's default epilogue is no longer bias + ReLU")
        except Exception as e:  # noqa: BLE001
            out.append(f"torch._addmm_activation(bias, mat1, mat2) raised {e!r}")
    if not hasattr(torch, "_foreach_copy_"):
        out.append("torch._foreach_copy_(dsts, srcs)")
    else:
        try:
            d, s = [torch.zeros(2), torch.zeros(3)], [torch.ones(2), torch.ones(3)]
            torch._foreach_copy_(d, s)
            if not all(bool((x == 1).all()) for x in d):
                out.append("torch._foreach_copy_(dsts, srcs) no longer copies srcs into dsts")
        except Exception as e:  # noqa: BLE001
            out.append(f"torch._foreach_copy_(dsts, srcs) raised {e!r}")
    return out


_checked = False


def check() -> None:
    """Raise ``TorchInterfaceError`` naming the torch version and every missing private interface; cached once green."""
    global _checked
    if _checked:
        return
    bad = problems()
    if bad:
        raise TorchInterfaceError(
            f"torch {torch.__version__} lacks private interfaces the device update path of this engine was written against "
            f"(torch 2.10): " + "; ".join(bad) + ".  See src/ppo/torch_compat.py for what each is used for.")
    _checked = True
