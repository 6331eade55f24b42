"""g2048_linear_bf16 vs F.linear (hipBLASLt): correctness and time for the update's shapes."""
import os
import sys
import time

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, os.path.join(ROOT, "2048-ppo-agent_amd"))
sys.path.insert(0, ROOT)
import torch
import torch.nn.functional as F

from src.g2048 import native as nv

dev = torch.device("cuda:0")
torch.manual_seed(0)


def bench(fn, n=30):
    """mean DEVICE time per call (kernel durations from the profiler; the host wrapper is slower than the kernel)"""
    from torch.profiler import ProfilerActivity, profile

    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    with profile(activities=[ProfilerActivity.CUDA]) as prof:
        for _ in range(n):
            fn()
        torch.cuda.synchronize()
    return sum(k.self_device_time_total for k in prof.key_averages()) / n


T = int(sys.argv[1]) if len(sys.argv) > 1 else 34816
for K, N in ((256, 256), (256, 768), (256, 1024), (1024, 256), (768, 256), (256, 512), (512, 256), (512, 512)):
    x = torch.randn(T, K, device=dev).to(torch.bfloat16)
    w = (torch.randn(N, K, device=dev) / K ** 0.5).to(torch.bfloat16)
    b = torch.randn(N, device=dev)
    bb = b.to(torch.bfloat16)
    ref = F.linear(x.float(), w.float(), b)
    y = nv.linear_bf16(x, w, b)
    yt = F.linear(x, w, bb)
    rel = lambda a: ((a.float() - ref).norm() / ref.norm()).item()
    out = torch.empty(T, N, dtype=torch.bfloat16, device=dev)
    t_ours = bench(lambda: nv.linear_bf16(x, w, b, out))
    t_torch = bench(lambda: F.linear(x, w, bb))
    gf = 2 * T * K * N / 1e6  # MFLOP; / us = TFLOP/s
    print(f"T={T} K={K} N={N}: rel err ours {rel(y):.2e} torch {rel(yt):.2e} | ours {t_ours:6.1f} us ({gf / t_ours:5.0f} TF/s) "
          f"torch {t_torch:6.1f} us ({gf / t_torch:5.0f} TF/s)", flush=True)
