"""Time of the fused rollout encoder (g2048_policy_encoder) at 65 536 boards, HIP events on its stream: bench.py's
policy_encoder object on its own.  usage: python tools/time_encoder.py [boards]"""
import json
import os
import sys

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, os.path.join(ROOT, "2048-ppo-agent_amd"))
sys.path.insert(0, ROOT)
import torch

import bench
from src.ppo import PPOAgent

dev = torch.device("cuda:0")
torch.manual_seed(0)
agent = PPOAgent(**bench.MODEL_CFG).to(dev).eval()
boards = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
best = None
for _ in range(3):
    r = bench.policy_encoder_roofline(agent, dev, boards, launches=10)
    if best is None or r["launch_ms"] < best["launch_ms"]:
        best = r
print(json.dumps(best))
