#!/usr/bin/env python3
"""g2048_linear_add_ln_fwd / _bwd against the launches they replace, at the update's shape (34 816 tokens), cold operands (three
rotating operand sets, ~0.5 GB: more than the 256 MB Infinity Cache holds).

    python tools/probe_rowgemm.py [T]

Per K in {256, 768, 1024}: forward fused vs (g2048_linear_bf16 or torch's bf16 GEMM) + g2048_add_ln_fwd; backward fused vs torch's bf16 GEMM
(hipBLASLt) + g2048_add_ln_bwd; microseconds per call by HIP events on the launch stream, and the algorithmic bytes over that time.
"""
import os
import sys

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, os.path.join(ROOT, "2048-ppo-agent_amd"))
import torch  # noqa: E402

from src.g2048 import native as nv  # noqa: E402

T = int(sys.argv[1]) if len(sys.argv) > 1 else 34816
dev = torch.device("cuda:0")
SETS, REPS = 3, 12
bf = torch.bfloat16


def timed(fn):
    for i in range(SETS):
        fn(i)
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for i in range(REPS):
        fn(i % SETS)
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) * 1e3 / REPS


for K in (256, 768, 1024):
    torch.manual_seed(K)
    W = (torch.randn(256, K, device=dev) / K ** 0.5).to(bf)
    Wp = nv.pack_fragments(W)
    Wt_dense = W.t().contiguous()  # [K][256]: what `dy @ w` multiplies with in the unfused backward
    bias, gamma, beta = torch.randn(256, device=dev) * 0.1, torch.ones(256, device=dev), torch.zeros(256, device=dev)
    S = []
    for _ in range(SETS):
        S.append(dict(u=(torch.randn(T, K, device=dev) * 0.5).to(bf), x=torch.randn(T, 256, device=dev), gx=torch.randn(T, 256, device=dev),
                      x_new=torch.empty(T, 256, device=dev), h=torch.empty(T, 256, dtype=bf, device=dev), a=torch.empty(T, 256, dtype=bf, device=dev),
                      mean=torch.zeros(T, device=dev), rstd=torch.ones(T, device=dev), dx=torch.empty(T, 256, device=dev),
                      da=torch.empty(T, 256, dtype=bf, device=dev)))
    p, seed = float(os.environ.get("P_DROP", "0.1")), 12345

    def fwd_fused(i):
        s = S[i]
        nv.linear_add_ln_fwd(s["u"], Wp, bias, s["x"].data_ptr(), 256, gamma, beta, s["x_new"], s["h"], s["mean"], s["rstd"], 1e-5, p, seed)

    def fwd_unfused(i):
        s = S[i]
        a = nv.linear_bf16(s["u"], W, bias) if (K <= 256 and nv.linear_ok(s["u"], W)) else torch.nn.functional.linear(s["u"], W, bias.to(bf))
        nv.add_ln_fwd(s["x"].data_ptr(), 256, a, gamma, beta, s["x_new"], s["h"], s["mean"], s["rstd"], T, 1e-5, p, seed)

    def gemm_only(i):
        s = S[i]
        if K <= 256 and nv.linear_ok(s["u"], W):
            nv.linear_bf16(s["u"], W, bias)
        else:
            torch.nn.functional.linear(s["u"], W, bias.to(bf))

    def bwd_fused(i):
        s = S[i]
        nv.linear_add_ln_bwd(s["u"], Wp, s["x"].data_ptr(), 256, s["gx"], s["mean"], s["rstd"], gamma, s["dx"], s["da"], p, seed)

    def bwd_unfused(i):
        s = S[i]
        g_h = s["u"] @ Wt_dense
        nv.add_ln_bwd(s["x"].data_ptr(), 256, s["gx"], g_h, s["mean"], s["rstd"], gamma, s["dx"], s["da"], None, T, p, seed)

    fwd_bytes = T * (2 * K + 1024 + 1024 + 512)
    bwd_bytes = T * (2 * K + 1024 + 1024 + 1024 + 512)
    tf, tu, tg = timed(fwd_fused), timed(fwd_unfused), timed(gemm_only)
    print(f"K {K:4d} forward : fused {tf:6.1f} us ({fwd_bytes / tf / 1e6:5.2f} TB/s of its algorithmic bytes)   unfused pair {tu:6.1f} us "
          f"(its GEMM alone {tg:5.1f})", flush=True)
    tf, tu = timed(bwd_fused), timed(bwd_unfused)
    print(f"K {K:4d} backward: fused {tf:6.1f} us ({bwd_bytes / tf / 1e6:5.2f} TB/s)   unfused pair {tu:6.1f} us", flush=True)
