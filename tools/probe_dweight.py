"""g2048_dweight_bf16 against the 16-slice batched hipBLASLt GEMM it replaces, at the update's shapes.  usage: probe_dweight.py [slices]"""
import os
import sys

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, os.path.join(ROOT, "2048-ppo-agent_amd"))
import torch

from src.g2048 import native as nv

dev = torch.device("cuda:0")
T = 34816
S = int(sys.argv[1]) if len(sys.argv) > 1 else 16
BR = int(sys.argv[2]) if len(sys.argv) > 2 else 0


def timed(fn, reps=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3


for N, K in ((1024, 256), (256, 1024), (768, 256), (256, 256)):
    dys = [(torch.randn(T, N, device=dev) / 8).to(torch.bfloat16) for _ in range(6)]
    xs = [torch.randn(T, K, device=dev).to(torch.bfloat16) for _ in range(6)]
    ref = dys[0].float().t() @ xs[0].float()
    ours = nv.dweight_parts(dys[0], xs[0], S, block_rows=BR).float().sum(0)
    blas = torch.bmm(dys[0].view(16, T // 16, -1).transpose(1, 2), xs[0].view(16, T // 16, -1)).float().sum(0)
    rel = lambda a: ((a - ref).norm() / ref.norm()).item()
    it = [0]

    def f_ours():
        it[0] += 1
        nv.dweight_parts(dys[it[0] % 6], xs[it[0] % 6], S, block_rows=BR)

    def f_blas():
        it[0] += 1
        torch.bmm(dys[it[0] % 6].view(16, T // 16, -1).transpose(1, 2), xs[it[0] % 6].view(16, T // 16, -1))

    print(f"dW [{N} x {K}] over {T} tokens: rel err ours {rel(ours):.2e} hipBLASLt {rel(blas):.2e} | ours ({S} slices, block rows {BR}) {timed(f_ours):6.1f} us"
          f"  hipBLASLt (16 slices) {timed(f_blas):6.1f} us")
