"""Capture the update graph for one model config (argv: layers dff hidden envs) - segfault bisection helper."""
import os
import sys

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, os.path.join(ROOT, "2048-ppo-agent_amd"))
sys.path.insert(0, ROOT)
import torch

import bench
from src.ppo import PPOAgent, PPOTrainer, RolloutBuffer
from src.ppo.data_loader import DeviceBatches, PPODataset
from src.ppo.ppo_trainer import _GraphedFwdBwd
from src.runs import BatchRunner

L, dff, hid, envs = [int(v) for v in sys.argv[1:5]]
dev = torch.device("cuda:0")
torch.manual_seed(0)
agent = PPOAgent(d_model=256, nhead=8, num_layers=L, dim_feedforward=dff, hidden_dim=hid, dropout=0.0, reduction="cls")
tr = PPOTrainer(agent, BatchRunner(0, device=dev), RolloutBuffer(31, 16, 4), bench.OPTIM_CFG, max_steps=1000, device=dev,
                rollout_amp=True, log_dir="/tmp/lg", max_samples_per_epoch=100000)
tr.collect_rollouts(envs, 1)
data = tr.rollout_buffer.device_data(dev)
ds = PPODataset(data, gamma=tr.gamma, lambda_gae=tr.lambda_gae, max_samples_per_epoch=20000, shuffle_on_reset=False)
b = list(DeviceBatches(ds, 2048, drop_last=True).epoch())[0]
agent.train()
obs, actions, masks, old_lp, adv, ret = tr._unpack_batch(b, packed=True)
print("capturing", sys.argv[1:], flush=True)
gr = _GraphedFwdBwd(tr, 2048, dict(obs=obs, actions=actions, masks=masks, old_lp=old_lp, adv=adv, ret=ret))
gr.run(dict(obs=obs, actions=actions, masks=masks, old_lp=old_lp, adv=adv, ret=ret))
torch.cuda.synchronize()
print("OK", sys.argv[1:], flush=True)
