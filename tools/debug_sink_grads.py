"""Per-step, per-parameter gradient comparison of the captured update with and without the GradSink (same seeds)."""
import os, sys
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, os.path.join(ROOT, "2048-ppo-agent_amd")); sys.path.insert(0, ROOT)
import torch
from torch.amp import GradScaler
import bench
from src.ppo import PPOAgent, PPOTrainer, RolloutBuffer
from src.runs import BatchRunner
dev = torch.device("cuda:0")
GRAPH = os.environ.get("DBG_GRAPH", "1") == "1"
def run(sink):
    os.environ["G2048_GRAD_SINK"] = "1" if sink else "0"
    from src.ppo import hip_ops
    hip_ops._capture_site[0] = 0; hip_ops._GRAPH_SEED.clear()
    torch.manual_seed(0)
    cfg = dict(bench.MODEL_CFG); cfg["dropout"] = float(os.environ.get("DBG_DROPOUT", cfg.get("dropout", 0.1)))
    agent = PPOAgent(**cfg)
    tr = PPOTrainer(agent, BatchRunner(0, device=dev), RolloutBuffer(31, 16, 4), bench.OPTIM_CFG, max_steps=500000, device=dev,
                    rollout_amp=True, log_dir="/tmp/lg", use_hip_graph=GRAPH, **bench.TRAINER_CFG)
    tr.scaler = GradScaler(init_scale=65536.0, growth_interval=int(os.environ.get("DBG_GROWTH", "4")))
    tr.collect_rollouts(2048, 1)
    tr.max_samples_per_epoch = 2048 * 12
    snaps = []
    global sums
    sums = []
    orig = tr._allreduce_grads
    def hook(*a, **k):
        r = orig(*a, **k)
        snaps.append(tr._flat_grad.clone() / tr.scaler.get_scale())
        try:
            st = list(tr._graphs.values())[0].static
            vals = st.values() if isinstance(st, dict) else (st if isinstance(st, (list, tuple)) else [st])
            sums.append(tuple(round(float(v.double().sum()), 3) for v in vals if torch.is_tensor(v)))
        except Exception as e:
            sums.append(repr(e)[:60])
        return r
    tr._allreduce_grads = hook
    torch.manual_seed(1)
    tr.update_policy(batch_size=2048, n_epochs=1)
    names = [n for n, _ in agent.named_parameters()]
    byid = {id(p): n for n, p in agent.named_parameters()}
    return snaps, [(byid[id(p)], o, p.numel()) for p, o in zip(tr._flat_step.params, tr._flat_step.offsets)]
a, lay = run(True); sa = sums
b, _ = run(False); sb = sums
print("inputs equal:", sa == sb, sa[:2], sb[:2])
print("steps", len(a), len(b), "graph", GRAPH)
for s in range(len(a)):
    worst = []
    for n, o, k in lay:
        x, y = a[s][o:o + k], b[s][o:o + k]
        worst.append(((x - y).norm().item() / max(y.norm().item(), 1e-30), n, x.norm().item(), y.norm().item()))
    worst.sort(reverse=True)
    tot = (a[s] - b[s]).norm().item() / b[s].norm().item()
    print(f"step {s}: total rel {tot:.4f} |g| {a[s].norm().item():.4f} {b[s].norm().item():.4f}  worst:",
          "; ".join(f"{n} {e:.3f} ({x:.2e} vs {y:.2e})" for e, n, x, y in worst[:4]))
