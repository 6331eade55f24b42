"""Build libg2048.so (HIP, gfx950) in-tree: 2048-ppo-agent_amd/lib/libg2048.so.

hipcc cross-compiles without a GPU.  Every csrc/*.hip becomes an object under build/ (compiled in parallel, rebuilt only
when it or a header changed), then one link.  The .so is git-ignored but travels with the working tree.
"""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
SOURCES = ["g2048.hip", "g2048_policy.hip", "g2048_attention.hip", "g2048_layernorm.hip", "g2048_ppo_loss.hip",
           "g2048_linear.hip", "g2048_optim.hip", "g2048_reduce.hip", "g2048_tail.hip", "g2048_dweight.hip", "g2048_rowgemm.hip", "g2048_mlp.hip"]
HEADERS = [os.path.join(CSRC, "g2048_device.h"), os.path.join(CSRC, "g2048_colsum_final.h"), os.path.join(CSRC, "g2048_mfma.h"),
           os.path.join(HERE, "..", "include", "g2048.h")]
OBJDIR = os.path.join(HERE, "build")
OUT = os.path.join(HERE, "lib", "libg2048.so")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC"]
# per-file additions.  g2048_policy.hip: no NaN handling -- without it every fmaxf on an MFMA result gets a
# canonicalising v_max in front (32 extra vector instructions per attention head in the encoder's softmax)
EXTRA = {"g2048_policy.hip": ["-fno-honor-nans"]}


def _stale(target: str, deps) -> bool:
    return not os.path.exists(target) or any(os.path.getmtime(target) < os.path.getmtime(d) for d in deps)


def build(force: bool = False, verbose: bool = False) -> str:
    os.makedirs(os.path.dirname(OUT), exist_ok=True)
    os.makedirs(OBJDIR, exist_ok=True)
    me = os.path.abspath(__file__)
    jobs = []
    for name in SOURCES:
        src, obj = os.path.join(CSRC, name), os.path.join(OBJDIR, name.replace(".hip", ".o"))
        if force or _stale(obj, [src, me] + HEADERS):
            jobs.append(["hipcc", *FLAGS, *EXTRA.get(name, []), "-c", src, "-o", obj])

    def run(cmd):
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)

    if jobs:
        with ThreadPoolExecutor(max_workers=min(len(jobs), 6)) as pool:
            list(pool.map(run, jobs))
    objs = [os.path.join(OBJDIR, n.replace(".hip", ".o")) for n in SOURCES]
    if jobs or _stale(OUT, objs):
        run(["hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-o", OUT, *objs])
    return OUT


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
