import sys, os, time
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, os.path.join(ROOT, "2048-ppo-agent_amd")); sys.path.insert(0, ROOT)
import torch, bench
from torch.nn.attention import sdpa_kernel, SDPBackend
from src.ppo import PPOAgent
dev = torch.device("cuda:0"); torch.manual_seed(0)
agent = PPOAgent(**bench.MODEL_CFG).to(dev).train()
opt = torch.optim.AdamW(agent.parameters(), lr=1e-4, fused=True)
M = 2048
boards = torch.randint(0, 12, (M, 16), dtype=torch.uint8, device=dev); acts = torch.randint(0,4,(M,),device=dev)
def step():
    with torch.autocast("cuda", dtype=torch.bfloat16):
        lp, v, ent = agent.evaluate_actions(boards, acts)
        loss = (-lp.mean() + v.float().pow(2).mean() - 0.01*ent.mean())
    opt.zero_grad(); loss.backward(); opt.step()
def t(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); s=time.time()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.time()-s)/n*1e3
print("default", t(step))
for name, be in (("flash", SDPBackend.FLASH_ATTENTION), ("efficient", SDPBackend.EFFICIENT_ATTENTION), ("math", SDPBackend.MATH)):
    try:
        with sdpa_kernel(be):
            print(name, t(step))
    except Exception as e:
        print(name, "failed", str(e)[:100])
