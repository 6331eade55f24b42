"""Throughput mode (per-lane auto-reset): env-steps/s of full PPO iterations as a function of the rollout horizon, with the
reference's update config unchanged (5 epochs x 300 000 samples in minibatches of 2048).  usage: sweep_horizon.py 128 256 ..."""
import json
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

dev = torch.device("cuda:0")
horizons = [int(a) for a in sys.argv[1:]] or [128, 256, 512]
for h in horizons:
    torch.manual_seed(0)
    tr = PPOTrainer(PPOAgent(**bench.MODEL_CFG), BatchRunner(init_seed=0, rng_mode="partitionable", device=dev),
                    RolloutBuffer(31, 16, 4), bench.OPTIM_CFG, max_steps=500000, device=dev, rollout_amp=True,
                    log_dir="/tmp/g2048_sweep", **dict(bench.TRAINER_CFG, rollout_mode="fixed_horizon", rollout_horizon=h))
    phases = {"collect": 0.0, "update": 0.0}
    steps = 0
    for it in range(3):  # 1 warm-up + 2 timed
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        tr.collect_rollouts(65536, 1)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        m = tr.update_policy(batch_size=2048, n_epochs=5)
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        if it:
            phases["collect"] += t1 - t0
            phases["update"] += t2 - t1
            steps += tr.last_rollout_stats["timesteps"]
    s = phases["collect"] + phases["update"]
    print(json.dumps({"horizon": h, "env_steps_per_sec": round(steps / s, 1), "env_steps_per_iteration": steps // 2,
                      "collect_s": round(phases["collect"] / 2, 3), "update_s": round(phases["update"] / 2, 3),
                      "n_updates": m["n_updates"], "hip_graph": m["hip_graph"]}), flush=True)
    del tr
    torch.cuda.empty_cache()
