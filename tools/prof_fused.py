import sys, os, time
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, os.path.join(ROOT, "2048-ppo-agent_amd")); sys.path.insert(0, ROOT)
import torch, bench
from src.ppo import PPOAgent
from src.ppo.fused_policy import FusedPolicy
dev = torch.device("cuda:0"); torch.manual_seed(0)
agent = PPOAgent(**bench.MODEL_CFG).to(dev).eval()
fp = FusedPolicy(agent)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
boards = torch.randint(0, 12, (B, 16), dtype=torch.uint8, device=dev)
for split in (False, True):
    for _ in range(3): fp.features(boards, split=split)
    torch.cuda.synchronize(); t = time.time()
    for _ in range(10): fp.features(boards, split=split)
    torch.cuda.synchronize(); print(f"B={B} split={split}: {(time.time()-t)/10*1e3:.3f} ms per forward; tiles per CU {B/7/256:.2f}")
a, b = fp.features(boards, split=False), fp.features(boards, split=True)
print("split vs single kernel: max |diff|", (a - b).abs().max().item(), "mean |diff|", (a - b).abs().mean().item(), "scale", a.abs().mean().item())
