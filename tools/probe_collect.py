"""Kernel breakdown of the collect phase (policy-in-the-loop rollout of 65536 boards with the bf16 Transformer).
usage: python tools/probe_collect.py [boards] [--fp32]   (--fp32: the reference's rollout precision, the PyTorch fp32 module forward)"""
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

B = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 65536
FP32 = "--fp32" in sys.argv
dev = torch.device("cuda:0")
torch.manual_seed(0)
agent = PPOAgent(**bench.MODEL_CFG)
tr = PPOTrainer(agent, BatchRunner(0, device=dev), RolloutBuffer(31, 16, 4), bench.OPTIM_CFG, max_steps=500000, device=dev,
                rollout_amp=not FP32, log_dir="/tmp/lg", **bench.TRAINER_CFG)
tr.collect_rollouts(B, 1)
tr.rollout_buffer.reset()
torch.cuda.synchronize()
t = time.time()
tr.collect_rollouts(B, 1)
torch.cuda.synchronize()
print("collect wall s", round(time.time() - t, 3), tr.last_rollout_stats)
tr.rollout_buffer.reset()
from torch.profiler import ProfilerActivity, profile

with profile(activities=[ProfilerActivity.CUDA]) as prof:
    tr.collect_rollouts(B, 1)
    torch.cuda.synchronize()
ka = prof.key_averages()
tot = sum(k.self_device_time_total for k in ka)
print(f"GPU busy {tot / 1e6:.3f} s in {sum(k.count for k in ka)} kernels")
for k in sorted(ka, key=lambda k: -k.self_device_time_total)[:25]:
    print(f"{k.self_device_time_total / 1e3:9.1f} ms {k.count:7d} x {k.self_device_time_total / max(k.count, 1):8.1f} us  {k.key[:110]}")
