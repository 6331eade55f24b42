"""Policy forward (bf16 autocast) at 65 536 boards, a few iterations, for rocprofv3."""
import sys, os
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, os.path.join(ROOT, "2048-ppo-agent_amd")); sys.path.insert(0, ROOT)
import torch, bench
from src.ppo import PPOAgent
dev = torch.device("cuda:0"); torch.manual_seed(0)
agent = PPOAgent(**bench.MODEL_CFG).to(dev).eval()
B = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
boards = torch.randint(0, 12, (B, 16), dtype=torch.uint8, device=dev)
with torch.no_grad(), torch.autocast("cuda", dtype=torch.bfloat16):
    for _ in range(int(sys.argv[2]) if len(sys.argv) > 2 else 3):
        agent(boards)
torch.cuda.synchronize()
