#!/usr/bin/env python3
"""Multi-seed A/B of the deferred weight gradients' partial format (VERDICT r3 "next" 2(ii)): does the learning speed of the
config-5 protocol depend on how the minibatch-sized weight gradients are split and rounded before they are summed?

    python tools/seeds_dweight_ab.py --seeds 1 2 3 4 5 --iterations 40 --out gpurun_out/ab/dweight_slices_seeds.txt

Arms (G2048_DWEIGHT_PARTS, read by hip_ops._dweight_parts_config per backward pass): bf16x8 (the tree's default since round 3),
bf16x16 (what round 3's `v6` tree ran), f32x8 (partials not rounded at all).  Per (arm, seed): a fresh default Transformer agent,
65 536 envs, the reference's trainer config (the protocol of profiles/round2_seed_variance.txt = run/train_to_2048.py), a fixed number
of PPO iterations; recorded per iteration: mean episode length of the rollout, entropy loss, KL.  Reported per arm: mean +- sd over the
seeds of the episode length at the last iteration and of the iteration at which the entropy starts to fall (first iteration with
entropy_loss > -1.20; the untrained policy sits at -1.237).  One process for all runs (a fresh trainer per run).
"""
import argparse
import gc
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "2048-ppo-agent_amd"))

import numpy as np  # noqa: E402
import torch  # noqa: E402

from src.ppo import PPOAgent, PPOTrainer, RolloutBuffer  # noqa: E402
from src.runs import BatchRunner  # noqa: E402

TRAINER = dict(gamma=0.99, lambda_gae=0.95, clip_epsilon=0.2, value_loss_coef=0.5, entropy_coef=0.01, max_grad_norm=0.5,
               target_kl=0.25, use_action_mask=True, mixed_precision="bfloat16", max_samples_per_epoch=300000,
               shuffle_on_reset=True)
OPTIM = dict(opt_name="adamw", max_lr=4e-4, betas=(0.9, 0.999), eps=1e-6, weight_decay=0.01, warmup_steps_ratio=0.025,
             scheduler_names=["constant", "constant"], blacklist_weight_modules=["norm", "embedding"])
MODEL = dict(observation_dim=31, action_dim=4, hidden_dim=512, d_model=256, nhead=8, num_layers=4, dim_feedforward=1024,
             dropout=0.1, reduction="cls")


def run(arm: str, seed: int, iterations: int, envs: int, dev):
    os.environ["G2048_DWEIGHT_PARTS"] = arm
    torch.manual_seed(seed)
    agent = PPOAgent(**MODEL)
    tr = PPOTrainer(agent, BatchRunner(seed, device=dev), RolloutBuffer(31, 16, 4), OPTIM, max_steps=500000, device=dev,
                    rollout_amp=True, log_dir="/tmp/g2048_seeds_ab", **TRAINER)
    rows = []
    t0 = time.perf_counter()
    for it in range(1, iterations + 1):
        tr.collect_rollouts(envs, 1)
        m = tr.update_policy(batch_size=2048, n_epochs=5)
        rows.append((it, tr.last_rollout_stats["mean_episode_length"], m["entropy_loss"], m["kl_divergence"], m["n_updates"]))
    torch.cuda.synchronize()
    assert m.get("hip_graph"), "the update fell back to eager mode"
    secs = time.perf_counter() - t0
    del tr, agent
    gc.collect()
    torch.cuda.empty_cache()
    return rows, secs


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--arms", nargs="+", default=["bf16x8", "bf16x16", "f32x8"])
    ap.add_argument("--seeds", nargs="+", type=int, default=[1, 2, 3, 4, 5])
    ap.add_argument("--iterations", type=int, default=40)
    ap.add_argument("--envs", type=int, default=65536)
    ap.add_argument("--out", default=None)
    a = ap.parse_args()
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    lines = [f"# {' '.join(sys.argv)}",
             f"# {a.envs} envs, default Transformer, reference trainer config, {a.iterations} PPO iterations per run; per run: entropy loss per "
             "iteration, episode length at the last iteration, first iteration with entropy_loss > -1.20 (fall)"]
    summary = {}
    for seed in a.seeds:  # seeds outermost: an interrupted sweep still has whole seeds for every arm
        for arm in a.arms:
            rows, secs = run(arm, seed, a.iterations, a.envs, dev)
            fall = next((it for it, _, e, _, _ in rows if e > -1.20), a.iterations + 1)
            summary.setdefault(arm, []).append((rows[-1][1], fall, rows[min(len(rows), 20) - 1][1]))
            ent = " ".join(f"{e:.3f}" for _, _, e, _, _ in rows)
            line = (f"seed {seed} {arm:8s}: len@{a.iterations} {rows[-1][1]:6.1f}  len@20 {rows[min(len(rows), 20) - 1][1]:6.1f}  fall@{fall:2d}  "
                    f"kl@{a.iterations} {rows[-1][3]:.4f}  {secs:5.1f} s | {ent}")
            print(line, flush=True)
            lines.append(line)
            if a.out:
                os.makedirs(os.path.dirname(os.path.abspath(a.out)) or ".", exist_ok=True)
                open(a.out, "w").write("\n".join(lines) + "\n")
    lines.append("")
    lines.append(f"# per arm over seeds {a.seeds}: mean +- sd (n - 1)")
    for arm, v in summary.items():
        v = np.array(v, dtype=np.float64)
        sd = v.std(axis=0, ddof=1) if len(v) > 1 else np.zeros(3)
        line = (f"{arm:8s}: episode length at iteration {a.iterations}: {v[:, 0].mean():6.1f} +- {sd[0]:5.1f}   at iteration 20: "
                f"{v[:, 2].mean():6.1f} +- {sd[2]:5.1f}   entropy-fall iteration: {v[:, 1].mean():5.1f} +- {sd[1]:4.1f}")
        print(line, flush=True)
        lines.append(line)
    if a.out:
        open(a.out, "w").write("\n".join(lines) + "\n")
        json.dump({k: [list(map(float, r)) for r in v] for k, v in summary.items()}, open(a.out + ".json", "w"))


if __name__ == "__main__":
    main()
