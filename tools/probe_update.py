import sys, os, time
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, os.path.join(ROOT, "2048-ppo-agent_amd")); sys.path.insert(0, ROOT)
import torch
import bench
from src.ppo import PPOAgent, PPOTrainer, RolloutBuffer
from src.runs import BatchRunner
dev = torch.device("cuda:0"); torch.manual_seed(0)
agent = PPOAgent(**bench.MODEL_CFG)
tr = PPOTrainer(agent, BatchRunner(0, device=dev), RolloutBuffer(31,16,4), bench.OPTIM_CFG, max_steps=500000, device=dev, rollout_amp=True, log_dir="/tmp/lg", **bench.TRAINER_CFG)
tr.collect_rollouts(8192, 1)
tr.max_samples_per_epoch = 40000
def upd():
    torch.cuda.synchronize(); t=time.time(); m=tr.update_policy(batch_size=2048, n_epochs=2); torch.cuda.synchronize(); return (time.time()-t)/max(m["n_updates"],1)*1e3, m["n_updates"]
print("warm", upd()); print("graph ms/minibatch", upd())
tr.use_hip_graph=False
print("eager ms/minibatch", upd())
tr.use_hip_graph=True
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CUDA]) as prof:
    upd()
print(prof.key_averages().table(sort_by="cuda_time_total", row_limit=25, max_name_column_width=70))
