"""Where does one PPO minibatch go?  wall ms/minibatch (eager and hipGraph), GPU-busy ms, kernels per minibatch,
and the kernel table by time and by launch count.  usage: python tools/probe_update.py [rows] [--mlp] [--eager] [--timeline] [--hostprof]"""
import os
import sys
import time

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, os.path.join(ROOT, "2048-ppo-agent_amd"))
sys.path.insert(0, ROOT)
import torch

import bench
from src.ppo import PPOAgent, PPOTrainer, RolloutBuffer
from src.runs import BatchRunner

rows = int(sys.argv[1]) if len(sys.argv) > 1 else 30
dev = torch.device("cuda:0")
torch.manual_seed(0)
from src.ppo import MLPAgent
agent = MLPAgent() if "--mlp" in sys.argv else PPOAgent(**bench.MODEL_CFG)
tr = PPOTrainer(agent, BatchRunner(0, device=dev), RolloutBuffer(31, 16, 4), bench.OPTIM_CFG, max_steps=500000, device=dev,
                rollout_amp=True, log_dir="/tmp/lg", **bench.TRAINER_CFG)
tr.collect_rollouts(8192, 1)
tr.max_samples_per_epoch = 40000


def upd():
    torch.cuda.synchronize()
    t = time.time()
    m = tr.update_policy(batch_size=2048, n_epochs=2)
    t_cpu = time.time() - t
    torch.cuda.synchronize()
    n = max(m["n_updates"], 1)
    return round((time.time() - t) / n * 1e3, 3), round(t_cpu / n * 1e3, 3), n


print("hipGraph update:", tr.use_hip_graph)
print("warm", upd())
print("graph (wall ms/minibatch, host-side ms/minibatch, minibatches)", upd())
if "--eager" in sys.argv:
    tr.use_hip_graph = False
    print("eager warm", upd())
    print("eager", upd())
if "--hostprof" in sys.argv:  # where the HOST side of a minibatch goes (matters when the GPU work per minibatch is ~0.15 ms: the MLP policy)
    import cProfile
    import pstats

    pr = cProfile.Profile()
    pr.enable()
    _, _, n_h = upd()
    pr.disable()
    st = pstats.Stats(pr)
    st.sort_stats("cumulative")
    print(f"host profile over {n_h} minibatches (cumulative seconds; divide by {n_h} for per-minibatch):")
    st.print_stats(28)
from torch.profiler import ProfilerActivity, profile

with profile(activities=[ProfilerActivity.CUDA]) as prof:
    _, _, n = upd()
ka = prof.key_averages()
tot = sum(k.self_device_time_total for k in ka)
cnt = sum(k.count for k in ka)
print(f"GPU busy {tot / n / 1e3:.3f} ms/minibatch in {cnt / n:.0f} kernels/minibatch")
print(ka.table(sort_by="cuda_time_total", row_limit=rows, max_name_column_width=90))
print("by launch count:")
for k in sorted(ka, key=lambda k: -k.count)[:rows]:
    print(f"{k.count / n:7.1f}/mb {k.self_device_time_total / n:9.1f} us/mb  {k.key[:150]}")
if "--timeline" in sys.argv:  # one minibatch in launch order: offset, duration, idle gap before the kernel
    ev = sorted((e for e in prof.events() if e.device_type == torch.autograd.DeviceType.CUDA and e.time_range is not None),
                key=lambda e: e.time_range.start)
    marks = [i for i, e in enumerate(ev) if "k_opt_adamw" in e.name]  # (the last kernel of a minibatch)
    if len(marks) >= 3:
        lo, hi = marks[len(marks) // 2 - 1] + 1, marks[len(marks) // 2] + 1
        t0, prev_end, gaps = ev[lo].time_range.start, ev[lo].time_range.start, 0.0
        print(f"timeline of one minibatch ({hi - lo} kernels):")
        for e in ev[lo:hi]:
            gap = e.time_range.start - prev_end
            gaps += max(gap, 0.0)
            print(f"{e.time_range.start - t0:9.1f} us  {e.time_range.end - e.time_range.start:7.1f} us  gap {gap:6.1f}  {e.name[:110]}")
            prev_end = max(prev_end, e.time_range.end)
        print(f"span {prev_end - t0:.1f} us, idle gaps {gaps:.1f} us")
