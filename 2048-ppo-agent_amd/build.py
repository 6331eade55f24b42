"""Build libg2048.so (HIP, gfx950) in-tree: 2048-ppo-agent_amd/lib/libg2048.so.

hipcc cross-compiles without a GPU.  The .so is git-ignored but travels with the working tree.
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, "csrc", "g2048.hip")
SRC_POLICY = os.path.join(HERE, "csrc", "g2048_policy.hip")
SRC_ATTN = os.path.join(HERE, "csrc", "g2048_attention.hip")
SRC_LN = os.path.join(HERE, "csrc", "g2048_layernorm.hip")
SRC_LOSS = os.path.join(HERE, "csrc", "g2048_ppo_loss.hip")
SRC_LIN = os.path.join(HERE, "csrc", "g2048_linear.hip")
DEPS = [SRC, SRC_POLICY, SRC_ATTN, SRC_LN, SRC_LOSS, SRC_LIN, os.path.join(HERE, "csrc", "g2048_device.h"), os.path.join(HERE, "..", "include", "g2048.h")]
OUT = os.path.join(HERE, "lib", "libg2048.so")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC", "-shared"]


def build(force: bool = False, verbose: bool = False) -> str:
    os.makedirs(os.path.dirname(OUT), exist_ok=True)
    if not force and os.path.exists(OUT) and all(os.path.getmtime(OUT) >= os.path.getmtime(d) for d in DEPS):
        return OUT
    cmd = ["hipcc", *FLAGS, "-o", OUT, SRC, SRC_POLICY, SRC_ATTN, SRC_LN, SRC_LOSS, SRC_LIN]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    return OUT


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
