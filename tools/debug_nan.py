import sys, os
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, os.path.join(ROOT, "2048-ppo-agent_amd")); sys.path.insert(0, ROOT)
import torch, bench
from src.ppo import PPOAgent, PPOTrainer, RolloutBuffer
from src.runs import BatchRunner
dev = torch.device("cuda:0"); torch.manual_seed(0)
cfg = dict(bench.MODEL_CFG)
if '--nodropout' in sys.argv: cfg['dropout'] = 0.0
agent = PPOAgent(**cfg)
tc = dict(bench.TRAINER_CFG)
if '--noamp' in sys.argv: tc['mixed_precision'] = None
tr = PPOTrainer(agent, BatchRunner(0, device=dev), RolloutBuffer(31,16,4), bench.OPTIM_CFG, max_steps=500000, device=dev, rollout_amp=True, log_dir="/tmp/lg", use_hip_graph=("--graph" in sys.argv), **tc)
import time
for it in range(3):
    tr.collect_rollouts(4096, 1)
    print("iter", it, "steps", tr.last_rollout_stats, flush=True)
    torch.cuda.synchronize(); t0=time.time(); m = tr.update_policy(batch_size=2048, n_epochs=2); torch.cuda.synchronize(); print(' ms/minibatch', (time.time()-t0)/max(m['n_updates'],1)*1e3)
    bad = [n for n,p in agent.named_parameters() if not torch.isfinite(p).all()]
    print(" metrics", {k: round(v,4) for k,v in m.items()}, "nonfinite params:", bad[:5], "scale", tr.scaler.get_scale() if tr.scaler else None, flush=True)
