"""Does the engine learn?  Train for a fixed wall-clock budget and evaluate with the reference's 1000-episode
protocol (seed 42, greedy + mask) next to the random (109.17) and DRUL (189.44) baselines."""
import argparse, json, os, sys, time
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, os.path.join(ROOT, "2048-ppo-agent_amd")); sys.path.insert(0, ROOT)
import torch
import bench
from src.ppo import MLPAgent, PPOAgent, PPOTrainer, RolloutBuffer
from src.runs import BatchRunner, evaluate_agent

ap = argparse.ArgumentParser()
ap.add_argument("--minutes", type=float, default=5.0)
ap.add_argument("--envs", type=int, default=4096)
ap.add_argument("--model", default="transformer")
ap.add_argument("--train-batch", type=int, default=2048)
ap.add_argument("--out", default=None)
a = ap.parse_args()
dev = torch.device("cuda:0"); torch.manual_seed(0)
agent = PPOAgent(**bench.MODEL_CFG) if a.model == "transformer" else MLPAgent(hidden_dim=512, trunk_dim=512)
tr = PPOTrainer(agent, BatchRunner(0, device=dev), RolloutBuffer(31, 16, 4), bench.OPTIM_CFG, max_steps=500000,
                device=dev, rollout_amp=True, log_dir="/tmp/g2048_demo_logs", **bench.TRAINER_CFG)
log = []
ev = evaluate_agent(agent, dev, 1000); ev["minutes"] = 0.0; ev["timesteps"] = 0; log.append(ev)
print("untrained:", ev["mean_max_tile"], ev["percent"], flush=True)
t0 = time.time(); it = 0; next_eval = a.minutes / 4
while (time.time() - t0) / 60 < a.minutes:
    it += 1
    tr.collect_rollouts(a.envs, 1)
    m = tr.update_policy(batch_size=a.train_batch, n_epochs=5)
    el = (time.time() - t0) / 60
    print(f"iter {it} {el:.2f} min  timesteps {tr.total_timesteps}  mean len {tr.last_rollout_stats['mean_episode_length']:.1f} "
          f"mean max step-reward {tr.last_rollout_stats['mean_max_episode_reward']:.1f}  kl {m['kl_divergence']:.4f}", flush=True)
    if el >= next_eval:
        ev = evaluate_agent(agent, dev, 1000); ev["minutes"] = round(el, 2); ev["timesteps"] = tr.total_timesteps; log.append(ev)
        print("eval:", ev["mean_max_tile"], ev["percent"], flush=True); next_eval += a.minutes / 4
ev = evaluate_agent(agent, dev, 1000); ev["minutes"] = round((time.time() - t0) / 60, 2); ev["timesteps"] = tr.total_timesteps; log.append(ev)
print("final:", ev["mean_max_tile"], ev["percent"], flush=True)
res = {"model": a.model, "envs": a.envs, "train_minutes": a.minutes, "evals": log,
       "baselines": {"random": 109.17, "drul": 189.44, "reference_ppo_readme": 383}}
print(json.dumps(res))
if a.out:
    json.dump(res, open(a.out, "w"), indent=1)
