"""Which GEMM shapes cost what inside the update (eager mode, torch profiler grouped by input shape)."""
import os
import sys

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, os.path.join(ROOT, "2048-ppo-agent_amd"))
sys.path.insert(0, ROOT)
import torch
from torch.profiler import ProfilerActivity, profile

import bench
from src.ppo import PPOAgent, PPOTrainer, RolloutBuffer
from src.runs import BatchRunner

dev = torch.device("cuda:0")
torch.manual_seed(0)
agent = PPOAgent(**bench.MODEL_CFG)
tr = PPOTrainer(agent, BatchRunner(0, device=dev), RolloutBuffer(31, 16, 4), bench.OPTIM_CFG, max_steps=500000, device=dev,
                rollout_amp=True, log_dir="/tmp/lg", use_hip_graph=False, **bench.TRAINER_CFG)
tr.collect_rollouts(8192, 1)
tr.max_samples_per_epoch = 40000
tr.update_policy(batch_size=2048, n_epochs=1)
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True) as prof:
    m = tr.update_policy(batch_size=2048, n_epochs=1)
    torch.cuda.synchronize()
n = m["n_updates"]
rows = [k for k in prof.key_averages(group_by_input_shape=True) if k.key in ("aten::mm", "aten::addmm", "aten::bmm", "aten::linear")]
rows.sort(key=lambda k: -k.device_time_total)
tot = 0
for k in rows:
    if k.key == "aten::linear":
        continue
    tot += k.device_time_total
    print(f"{k.key:12s} {k.count / n:5.1f}/mb {k.device_time_total / k.count:8.1f} us  {k.device_time_total / n:8.1f} us/mb  {k.input_shapes}")
print("GEMM total us/mb", tot / n)
