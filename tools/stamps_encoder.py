"""Where the rollout encoder's cycles go, phase by phase (diagnostic build, never the product library).

Builds csrc/g2048_policy.hip with -DG2048_STAMPS into tools/_build/libg2048_stamps.so (s_memtime stamps at the phase
boundaries of k_encoder_main, summed per phase by wave 0 of every 64th workgroup), points the binding at it via
G2048_LIB and prints the share of each phase.  The stamps fence the instruction stream, so read SHARES, not totals.
"""
import ctypes
import json
import os
import subprocess
import sys

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
PKG = os.path.join(ROOT, "2048-ppo-agent_amd")
OUT = os.path.join(ROOT, "tools", "_build", "libg2048_stamps.so")
NAMES = ["layer prologue: stage + LN1 + wait + barrier", "project(0) + store + barrier", "head: K reads + fetch issue", "head: S MFMAs",
         "head: projections(h+1) || softmax(h)", "head: P.V + scale + pack", "head: wait Wo + barrier 1", "head: store K/V + out-proj",
         "head: vmcnt(0)", "head: barrier 2", "ffn: +bo, LN2", "ffn: linear1(0) + barrier", "chunk: (entry)",
         "(unused)", "chunk: linear1(c+1) || pack(c), linear2(c)", "chunk: vmcnt(0) (+b2)", "chunk: barrier", "(exit layers)", "last layer K/V part"]


def build():
    os.makedirs(os.path.dirname(OUT), exist_ok=True)
    src = [os.path.join(PKG, "csrc", f) for f in ("g2048.hip", "g2048_policy.hip", "g2048_attention.hip", "g2048_layernorm.hip",
                                                  "g2048_ppo_loss.hip", "g2048_linear.hip")]
    subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fno-honor-nans", "-fPIC",
                           "-shared", "-DG2048_STAMPS", "-o", OUT, *src])


if __name__ == "__main__":
    if "--build" in sys.argv:
        build()
        sys.exit(0)
    os.environ["G2048_LIB"] = OUT
    sys.path.insert(0, PKG)
    sys.path.insert(0, ROOT)
    import torch

    import bench
    from src.g2048 import native as nv
    from src.ppo import PPOAgent
    from src.ppo.fused_policy import FusedPolicy

    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    agent = PPOAgent(**bench.MODEL_CFG).to(dev).eval()
    fp = FusedPolicy(agent)
    B = 65536
    boards = torch.randint(0, 12, (B, 16), dtype=torch.uint8, device=dev)
    lib = nv.load()
    buf = (ctypes.c_ulonglong * 24)()
    for _ in range(3):
        fp.features(boards, split=True)
    torch.cuda.synchronize()
    lib.g2048_debug_stamps.argtypes = [ctypes.c_void_p, ctypes.c_int]
    lib.g2048_debug_stamps(buf, 1)
    n = 5
    for _ in range(n):
        fp.features(boards, split=True)
    torch.cuda.synchronize()
    lib.g2048_debug_stamps(buf, 0)
    v = [int(x) for x in buf]
    tot = sum(v)
    waves = n * ((B + 6) // 7 + 63) // 64
    rows = [{"phase": NAMES[i], "cycles_per_tile": round(v[i] / max(waves, 1)), "share": round(v[i] / tot, 4)} for i in range(len(NAMES))]
    for r in rows:
        print(f"{r['share']*100:6.2f} %  {r['cycles_per_tile']:>9}  {r['phase']}")
    print("total cycles per tile (stamped build):", round(tot / max(waves, 1)))
    json.dump(rows, open(os.path.join(ROOT, "gpurun_out", "encoder_stamps.json"), "w"), indent=1)
