"""g2048_linear_bf16 [34816, 256] x [256 -> N] with COLD operands (rotating through more buffers than the 256 MB MALL holds),
for rocprofv3 passes (tools/pmc_linear.sh).  usage: prof_linear.py [N]"""
import os
import sys

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, os.path.join(ROOT, "2048-ppo-agent_amd"))
import torch

from src.g2048 import native as nv

N = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
dev = torch.device("cuda:0")
T, K = 34816, 256
xs = [torch.randn(T, K, device=dev).to(torch.bfloat16) for _ in range(8)]
ys = [torch.empty(T, N, device=dev, dtype=torch.bfloat16) for _ in range(8)]
w = (torch.randn(N, K, device=dev) / 16).to(torch.bfloat16)
b = torch.randn(N, device=dev)
for it in range(24):
    nv.linear_bf16(xs[it % 8], w, b, ys[it % 8])
torch.cuda.synchronize()
print("done")
