"""Where k_linear_ws's cycles go, phase by phase (diagnostic build, never the product library).

Builds the library with -DG2048_WS_STAMPS into tools/_build/libg2048_ws_stamps.so (s_memtime stamps at the phase boundaries, summed
per phase by one fetching and one storing wave of every 16th workgroup row of N-slice 0), points the binding at it via G2048_LIB and
prints cycles per 64-token tile for both roles.  usage: stamps_linear.py [--build] [N]"""
import ctypes
import glob
import json
import os
import subprocess
import sys

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
PKG = os.path.join(ROOT, "2048-ppo-agent_amd")
OUT = os.path.join(ROOT, "tools", "_build", "libg2048_ws_stamps.so")
NAMES = ["vmcnt(0): next tile's fetch + previous tile's stores", "barrier 1 (other buffer free)", "issue next fetch", "bias + 32 MFMAs",
         "barrier 2 (X buffer free)", "epilogue -> staging tile", "barrier 3 (tile staged, next published)", "copy-out"]


def build():
    os.makedirs(os.path.dirname(OUT), exist_ok=True)
    src = sorted(glob.glob(os.path.join(PKG, "csrc", "*.hip")))
    subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC", "-shared",
                           "-DG2048_WS_STAMPS", "-o", OUT, *src])


if __name__ == "__main__":
    if "--build" in sys.argv:
        build()
        sys.exit(0)
    os.environ["G2048_LIB"] = OUT
    sys.path.insert(0, PKG)
    import torch

    from src.g2048 import native as nv

    N = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
    dev = torch.device("cuda:0")
    T, K = 34816, 256
    xs = [torch.randn(T, K, device=dev).to(torch.bfloat16) for _ in range(8)]
    ys = [torch.empty(T, N, device=dev, dtype=torch.bfloat16) for _ in range(8)]
    w = (torch.randn(N, K, device=dev) / 16).to(torch.bfloat16)
    b = torch.randn(N, device=dev)
    lib = nv.load()
    lib.g2048_debug_ws_stamps.argtypes = [ctypes.c_void_p, ctypes.c_int]
    buf = (ctypes.c_ulonglong * 16)()
    for it in range(8):
        nv.linear_bf16(xs[it % 8], w, b, ys[it % 8])
    torch.cuda.synchronize()
    lib.g2048_debug_ws_stamps(buf, 1)
    n = 16
    for it in range(n):
        nv.linear_bf16(xs[it % 8], w, b, ys[it % 8])
    torch.cuda.synchronize()
    lib.g2048_debug_ws_stamps(buf, 0)
    groups = (256 // (N // 256)) if N % 256 == 0 else 512 // (N // 128)
    groups -= groups % 8
    tiles = (T + 63) // 64
    stamped_rows = (groups + 15) // 16
    tiles_seen = n * sum(len(range(g, tiles, groups)) for g in range(0, groups, 16))
    out = {}
    for role, name in enumerate(("wave 0", "wave 1")):
        v = [int(buf[8 * role + i]) for i in range(8)]
        tot = sum(v)
        print(f"{name}: {tot / tiles_seen:.0f} cycles per tile (s_memtime ticks)")
        for i in range(8):
            print(f"   {v[i] / tiles_seen:8.0f}  {100 * v[i] / tot:5.1f} %  {NAMES[i]}")
        out[name] = {NAMES[i]: round(v[i] / tiles_seen) for i in range(8)}
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    json.dump({"N": N, "tiles_per_launch": tiles, "workgroup_rows": groups, "per_tile_ticks": out},
              open(os.path.join(ROOT, "gpurun_out", "linear_ws_stamps.json"), "w"), indent=1)
