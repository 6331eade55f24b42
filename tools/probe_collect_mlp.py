"""Where does a lock-step of the MLP policy's rollout go (4 096 boards)?  Kernel table + launch-order timeline of one step."""
import os, sys, time
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, os.path.join(ROOT, "2048-ppo-agent_amd")); sys.path.insert(0, ROOT)
import torch
import bench
from src.ppo import MLPAgent, PPOTrainer, RolloutBuffer
from src.runs import BatchRunner
dev = torch.device("cuda:0")
torch.manual_seed(0)
tr = PPOTrainer(MLPAgent(), BatchRunner(0, device=dev), RolloutBuffer(31, 16, 4), bench.OPTIM_CFG, max_steps=500000, device=dev,
                rollout_amp=True, log_dir="/tmp/lg", **bench.TRAINER_CFG)
for _ in range(2):
    tr.collect_rollouts(4096, 1)
torch.cuda.synchronize(); t = time.time(); tr.collect_rollouts(4096, 1); torch.cuda.synchronize()
dt = time.time() - t
T = int(tr.last_rollout_stats["timesteps"])
print("collect", round(dt, 4), "s, env-steps", T, "graphs", list(tr._rollout_graphs.keys())[:2])
from torch.profiler import ProfilerActivity, profile
with profile(activities=[ProfilerActivity.CUDA, ProfilerActivity.CPU]) as prof:
    tr.collect_rollouts(4096, 1)
    torch.cuda.synchronize()
ka = [k for k in prof.key_averages() if k.self_device_time_total > 0]
tot = sum(k.self_device_time_total for k in ka)
print(f"GPU busy {tot / 1e3:.1f} ms")
for k in sorted(ka, key=lambda k: -k.self_device_time_total)[:25]:
    print(f"{k.count:6d} x {k.self_device_time_total / max(k.count, 1):8.1f} us = {k.self_device_time_total / 1e3:8.2f} ms  {k.key[:110]}")
ev = sorted((e for e in prof.events() if e.device_type == torch.autograd.DeviceType.CUDA), key=lambda e: e.time_range.start)
marks = [i for i, e in enumerate(ev) if "k_policy_step" in e.name]
if len(marks) > 40:
    lo, hi = marks[30] + 1, marks[32] + 1
    t0 = ev[lo].time_range.start; prev = t0
    for e in ev[lo:hi]:
        print(f"{e.time_range.start - t0:9.1f} {e.time_range.end - e.time_range.start:7.1f} gap {e.time_range.start - prev:6.1f} {e.name[:90]}")
        prev = max(prev, e.time_range.end)
