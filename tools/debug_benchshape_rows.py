"""Per-row comparison of the HIP update forward with an fp32 PyTorch forward on the minibatches of
tests/test_gpu_reference_vectors.py::test_hip_graph_update_at_bench_shape_matches_fp32_backward."""
import copy, sys, os, tempfile
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "2048-ppo-agent_amd"))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tests"))
import torch
from src.ppo import PPOAgent, PPOTrainer, RolloutBuffer
from src.ppo.data_loader import DeviceBatches, PPODataset
from src.runs import BatchRunner
from test_gpu_reference_vectors import OPTIM

dev = torch.device("cuda:0")
torch.manual_seed(0)
agent = PPOAgent(d_model=256, nhead=8, num_layers=4, dim_feedforward=1024, hidden_dim=512, dropout=0.0, reduction="cls")
tr = PPOTrainer(agent, BatchRunner(0, device=dev), RolloutBuffer(31, 16, 4), dict(OPTIM), max_steps=1000, device=dev,
                rollout_amp=True, log_dir=tempfile.mkdtemp(), max_samples_per_epoch=100000, use_action_mask=True)
tr.collect_rollouts(256, 1)
M = 2048
data = tr.rollout_buffer.device_data(dev)
ds = PPODataset(data, gamma=tr.gamma, lambda_gae=tr.lambda_gae, max_samples_per_epoch=4 * M, shuffle_on_reset=False)
db = DeviceBatches(ds, M, drop_last=True)
samples = [db.gather_packed(idx) for idx in list(db.indices())[:3]]
agent.train()
ref = copy.deepcopy(agent).float().train()
ref.transformer._shadow, ref._head_shadow = None, None
for i, s in enumerate(samples):
    with torch.no_grad():
        l32, v32 = ref(s["obs"], None)
    with torch.autocast("cuda", dtype=torch.bfloat16):
        l16, v16 = agent(s["obs"], None)  # grad mode + autocast: the HIP update forward
    l16, v16 = l16.float().detach(), v16.float().detach().flatten()
    v32 = v32.flatten()
    ev = (v16 - v32).abs()
    el = (l16 - l32).abs().amax(1)
    res = (v32 - s["ret"])
    print(f"sample {i}: |v|~{v32.abs().mean():.4f} residual rms {res.pow(2).mean().sqrt():.4f} value err rms {ev.pow(2).mean().sqrt():.5f} max {ev.max():.5f}"
          f" | logit err rms {el.pow(2).mean().sqrt():.5f} max {el.max():.5f} | value noise/residual {(v16 - v32).norm() / res.norm():.5f}")
    top = ev.topk(5).indices
    print("   worst rows", top.tolist(), "err", ev[top].tolist(), "max tile", s["obs"][top].amax(1).tolist())
    print("   adv rms", s["adv"].pow(2).mean().sqrt().item(), "adv max", s["adv"].abs().max().item(), "ret rms", s["ret"].pow(2).mean().sqrt().item())

from test_gpu_reference_vectors import _fp32_reference_grads
scale = tr.scaler.get_scale()
for i, s in enumerate(samples):
    want, _ = _fp32_reference_grads(agent, tr, s)
    tr._loss_backward(**s)
    g = torch.cat([(p.grad / scale).flatten() for p in agent.parameters()])
    w = torch.cat([x.flatten() for x in want])
    print(f"sample {i}: |g32| {w.norm():.5f} |g_hip - g32| {(g - w).norm():.5f} rel {(g - w).norm() / w.norm():.4f} mean ret {s['ret'].mean():.4f} mean adv {s['adv'].mean():.4f}")
