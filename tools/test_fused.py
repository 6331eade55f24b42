import sys, os, time
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, os.path.join(ROOT, "2048-ppo-agent_amd")); sys.path.insert(0, ROOT)
import torch, bench
from src.ppo import PPOAgent
from src.ppo.fused_policy import FusedPolicy
dev = torch.device("cuda:0"); torch.manual_seed(0)
nl = int(sys.argv[1]) if len(sys.argv) > 1 else 4
cfg = dict(bench.MODEL_CFG); cfg["num_layers"] = nl
agent = PPOAgent(**cfg).to(dev).eval()
with torch.no_grad():
    for p in agent.parameters():
        if p.dim() == 1: p.add_(torch.randn_like(p) * 0.05)   # non-trivial biases / LN params
fp = FusedPolicy(agent)
for B in (7, 20, 1000):
    boards = torch.randint(0, 12, (B, 16), dtype=torch.uint8, device=dev)
    with torch.no_grad():
        ref32 = agent.features(boards)
        with torch.autocast("cuda", dtype=torch.bfloat16):
            ref16 = agent.features(boards).float()
    got = fp.features(boards); torch.cuda.synchronize()
    e16 = (got - ref16).abs(); e32 = (got - ref32).abs(); eb = (ref16 - ref32).abs()
    print(f"layers={nl} B={B}: |fused-autocast| max {e16.max():.4f} mean {e16.mean():.5f} ; |fused-fp32| max {e32.max():.4f} mean {e32.mean():.5f} ; |autocast-fp32| max {eb.max():.4f} mean {eb.mean():.5f} ; ref scale {ref32.abs().mean():.3f} finite {torch.isfinite(got).all().item()}")
B = 65536
boards = torch.randint(0, 12, (B, 16), dtype=torch.uint8, device=dev)
fp.features(boards); torch.cuda.synchronize(); t=time.time()
for _ in range(5): fp.features(boards)
torch.cuda.synchronize(); dt=(time.time()-t)/5
print(f"fused encoder B={B}: {dt*1e3:.2f} ms  ({B*110e6*nl/4/dt/1e12:.1f} TF/s)")
with torch.no_grad(), torch.autocast("cuda", dtype=torch.bfloat16):
    agent.features(boards); torch.cuda.synchronize(); t=time.time()
    for _ in range(5): agent.features(boards)
    torch.cuda.synchronize(); print(f"torch autocast encoder: {(time.time()-t)/5*1e3:.2f} ms")
