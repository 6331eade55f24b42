"""Time g2048_attn_fwd / g2048_attn_bwd alone at the update's shape (2048 boards x 8 heads x 17 tokens, packed in_proj
output, dropout 0.1) - for A/B runs of kernel variants: `G2048_LIB=tools/_build/<variant>.so python tools/time_attn.py`.
Prints one line: both times and two checksums of dq/dk/dv (equal checksums across libraries on the same inputs = bitwise-equal arithmetic
as far as a sum can tell; the parity tests are tests/test_gpu_ppo.py)."""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "2048-ppo-agent_amd"))
import torch

from src.g2048 import native as nv

B, H, S, D = 2048, 8, 17, 256
dev = torch.device("cuda:0")
g = torch.Generator(device="cpu").manual_seed(3)
qkv = (torch.randn(B, S, 3 * D, generator=g) * 0.7).to(dev, torch.bfloat16)
dout = (torch.randn(B, S, D, generator=g) * 0.01).to(dev, torch.bfloat16)
o = torch.empty(B, S, D, dtype=torch.bfloat16, device=dev)
lse = torch.empty(B * H * S, dtype=torch.float32, device=dev)
dqkv = torch.zeros_like(qkv)
st = (S * 3 * D, 3 * D) * 3
p0 = qkv.data_ptr()
d0 = dqkv.data_ptr()
scale, p, seed = 32 ** -0.5, 0.1, 1234


def fwd():
    nv.attn_fwd(p0, p0 + 2 * D, p0 + 4 * D, o, lse, B, H, S, st, scale, p, seed)


def bwd():
    nv.attn_bwd(p0, p0 + 2 * D, p0 + 4 * D, dout, lse, d0, d0 + 2 * D, d0 + 4 * D, B, H, S, st, scale, p, seed)


def timed(fn, n=200):
    for _ in range(20):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3


fwd()
tf, tb = timed(fwd), timed(bwd)
print(f"{os.path.basename(nv.LIB_PATH)}: attn fwd {tf:.1f} us  bwd {tb:.1f} us  checksum {dqkv.float().abs().sum().item():.6e} "
      f"{dqkv.float().sum().item():.6e}")
