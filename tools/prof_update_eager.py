"""A few eager (no hipGraph) minibatches of the PPO update at the bench shape, for rocprofv3: every dispatch is a separate
kernel-trace / counter record.  Used by tools/pmc_update.sh."""
import os
import sys

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, os.path.join(ROOT, "2048-ppo-agent_amd"))
sys.path.insert(0, ROOT)
import torch

import bench
from src.ppo import PPOAgent, PPOTrainer, RolloutBuffer
from src.runs import BatchRunner

dev = torch.device("cuda:0")
torch.manual_seed(0)
tr = PPOTrainer(PPOAgent(**bench.MODEL_CFG), BatchRunner(0, device=dev), RolloutBuffer(31, 16, 4), bench.OPTIM_CFG,
                max_steps=500000, device=dev, rollout_amp=True, log_dir="/tmp/lg", use_hip_graph=False, **bench.TRAINER_CFG)
tr.collect_rollouts(2048, 1)
tr.max_samples_per_epoch = 8 * 2048
m = tr.update_policy(batch_size=2048, n_epochs=1)
torch.cuda.synchronize()
print("minibatches", m["n_updates"], "hip_graph", m["hip_graph"])
