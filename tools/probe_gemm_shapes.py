"""Which GEMM of one PPO minibatch costs what IN THE PIPELINE (cold operands, real neighbours): eager forward+backward
under the profiler with input shapes recorded; prints aten GEMM ops grouped by (op, shapes) with device time per call."""
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
tr.max_samples_per_epoch = 20000
tr.update_policy(batch_size=2048, n_epochs=1)  # warm
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True) as prof:
    m = tr.update_policy(batch_size=2048, n_epochs=1)
    torch.cuda.synchronize()
n = m["n_updates"]
rows = []
for k in prof.key_averages(group_by_input_shape=True):
    if k.key in ("aten::mm", "aten::addmm", "aten::bmm", "aten::linear", "aten::matmul", "aten::_addmm_activation") \
            and k.key in ("aten::mm", "aten::addmm", "aten::bmm"):
        rows.append((k.device_time_total / n, k.count / n, k.key, str(k.input_shapes)))
rows.sort(reverse=True)
print(f"{n} minibatches; GEMM ops per minibatch, device us per minibatch (sum over calls), calls, op, shapes")
tot = 0.0
for us, c, key, shp in rows:
    tot += us
    print(f"{us:9.1f} us  {c:5.1f}x  {us / c:7.1f} us/call  {key:12s} {shp}")
print(f"total {tot:.1f} us/minibatch")
print("\nother aten ops by device time (self), per minibatch:")
others = []
for k in prof.key_averages(group_by_input_shape=True):
    if k.key.startswith("aten::") and k.key not in ("aten::mm", "aten::addmm", "aten::bmm") and k.self_device_time_total > 0:
        others.append((k.self_device_time_total / n, k.count / n, k.key, str(k.input_shapes)[:150]))
others.sort(reverse=True)
for us, c, key, shp in others[:45]:
    print(f"{us:9.1f} us  {c:5.1f}x  {key:28s} {shp}")
print(f"total {sum(o[0] for o in others):.1f} us/minibatch in {sum(o[1] for o in others):.0f} ops")
