"""Streaming write / read / copy rates of this MI355X through plain PyTorch kernels on buffers far larger than the 256 MB Infinity
Cache: the reference point for the update's write-heavy kernels (NOTES.md 3).  usage: python tools/hbm_rw_rates.py"""
import json

import torch

dev = torch.device("cuda:0")
n = 1 << 30  # 1 GiB per buffer
a = torch.empty(n, dtype=torch.uint8, device=dev).view(torch.float32)
b = torch.empty(n, dtype=torch.uint8, device=dev).view(torch.float32)
a.normal_()


def timed(fn, reps=10):
    fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / reps * 1e-3


out = {}
t = timed(lambda: b.zero_())
out["write_only_fill"] = {"GBps": round(n / t / 1e9, 1)}
t = timed(lambda: b.copy_(a))
out["copy"] = {"GBps_read_plus_write": round(2 * n / t / 1e9, 1), "GBps_each_way": round(n / t / 1e9, 1)}
t = timed(lambda: a.sum())
out["read_only_sum"] = {"GBps": round(n / t / 1e9, 1)}
t = timed(lambda: torch.add(a, 1.0, out=b))
out["add_scalar_out"] = {"GBps_read_plus_write": round(2 * n / t / 1e9, 1)}
print(json.dumps(out))
