"""hipGraph capture helper shared by every capture site of the engine (the update in ``ppo_trainer._GraphedFwdBwd``, the
MLP policy's rollout forward in ``torch_action_wrapper.TorchActionFunction._graphed``).

Why a guard: Python's cyclic garbage collector may run at ANY allocation, also in the middle of a stream capture.  If that
collection finalises device objects of earlier work -- in the recorded case (NOTES.md 3, "The 10:48 abort") a previous
``PPOTrainer`` kept alive only by reference cycles, i.e. its ``_GraphedFwdBwd`` objects: ``torch.cuda.CUDAGraph`` instances
with their private allocator pools and the static input / gradient tensors allocated from those pools -- then the C++
destructors run on the capturing thread: ``CUDAGraph::~CUDAGraph`` destroys the graph / graph-exec handles and releases the
pool, whose segments the caching allocator returns with ``hipFree``.  Those calls are among the ones a thread with an
ongoing (thread-local or global) capture must not make; HIP answers with a capture error, c10 turns it into a C++ exception,
and an exception leaving a destructor that the interpreter called from ``tp_dealloc`` cannot propagate: ``std::terminate``,
SIGABRT ("Fatal Python error: Aborted" with "Garbage-collecting" on the stack).  No ``except`` clause sees it.

So: collect BEFORE the capture (whatever garbage exists is finalised while that is still legal), and keep the collector off
until the capture has ended.  Reference counting still frees acyclic temporaries during the capture; those are tensors of the
capture itself, which the allocator handles (they go back to the graph's private pool without any HIP call).
"""
import contextlib
import gc

import torch


@contextlib.contextmanager
def no_gc_during_capture():
    """``gc.collect()``, then the cyclic collector disabled for the duration of the block (restored afterwards)."""
    gc.collect()
    was_enabled = gc.isenabled()
    gc.disable()
    try:
        yield
    finally:
        if was_enabled:
            gc.enable()


@contextlib.contextmanager
def capture(graph: "torch.cuda.CUDAGraph", **kwargs):
    """``torch.cuda.graph(graph, capture_error_mode="thread_local", **kwargs)`` under ``no_gc_during_capture``.

    thread_local: only this thread's calls are checked during the capture.  Other threads (the RCCL watchdog of a multi-GPU
    run polling its events) must not be able to invalidate it; the autograd worker's launches are captured either way because
    capture is a property of the stream."""
    kwargs.setdefault("capture_error_mode", "thread_local")
    with no_gc_during_capture():
        with torch.cuda.graph(graph, **kwargs):
            yield graph


class CollectionsWhileCapturing:
    """Test instrument: counts garbage collections that START while the current stream is capturing (``gc.callbacks``)."""

    def __init__(self):
        self.during_capture, self.total = 0, 0

    def _cb(self, phase, info):
        if phase != "start":
            return
        self.total += 1
        if torch.cuda.is_available() and torch.cuda.is_current_stream_capturing():
            self.during_capture += 1

    def __enter__(self):
        gc.callbacks.append(self._cb)
        return self

    def __exit__(self, *exc):
        gc.callbacks.remove(self._cb)
        return False
