"""Host-side cost of one PPO minibatch: cProfile of update_policy (the update is launch-bound in eager mode)."""
import cProfile
import os
import pstats
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
agent = PPOAgent(**bench.MODEL_CFG)
tr = PPOTrainer(agent, BatchRunner(0, device=dev), RolloutBuffer(31, 16, 4), bench.OPTIM_CFG, max_steps=500000, device=dev,
                rollout_amp=True, log_dir="/tmp/lg", **bench.TRAINER_CFG)
tr.collect_rollouts(8192, 1)
tr.max_samples_per_epoch = 40000
tr.update_policy(batch_size=2048, n_epochs=2)
pr = cProfile.Profile()
pr.enable()
m = tr.update_policy(batch_size=2048, n_epochs=2)
torch.cuda.synchronize()
pr.disable()
print("minibatches", m["n_updates"])
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(45)
st.sort_stats("cumtime").print_stats(60)
