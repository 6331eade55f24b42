import sys, os, time
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, os.path.join(ROOT, "2048-ppo-agent_amd")); sys.path.insert(0, ROOT)
import torch, bench
import src.ppo.transformer_encoder as te
from src.ppo import PPOAgent
dev = torch.device("cuda:0"); torch.manual_seed(0)
cfg = dict(bench.MODEL_CFG); cfg["dropout"] = 0.0
agent = PPOAgent(**cfg).to(dev).train()
M = 2048
boards = torch.randint(0, 12, (M, 16), dtype=torch.uint8, device=dev); acts = torch.randint(0,4,(M,),device=dev)
def grads(use_splitk):
    orig = te._linear
    if not use_splitk: te._linear = lambda x, w, b: torch.nn.functional.linear(x, w, b)
    agent.zero_grad(set_to_none=True)
    with torch.autocast("cuda", dtype=torch.bfloat16):
        lp, v, ent = agent.evaluate_actions(boards, acts)
        loss = (-lp.mean() + v.float().pow(2).mean() - 0.01*ent.mean())
    loss.backward(); te._linear = orig
    return loss.item(), {n: p.grad.detach().clone() for n, p in agent.named_parameters()}
l0, g0 = grads(False); l1, g1 = grads(True)
# fp32 reference
agent.zero_grad(set_to_none=True)
lp, v, ent = agent.evaluate_actions(boards, acts); (-lp.mean() + v.float().pow(2).mean() - 0.01*ent.mean()).backward()
g32 = {n: p.grad.detach().clone() for n, p in agent.named_parameters()}
print("loss", l0, l1)
worst = 0
for n in g0:
    d01 = (g0[n]-g1[n]).norm() / (g0[n].norm() + 1e-12); d0 = (g0[n]-g32[n]).norm()/(g32[n].norm()+1e-12); d1 = (g1[n]-g32[n]).norm()/(g32[n].norm()+1e-12)
    worst = max(worst, d1.item() / max(d0.item(), 1e-6))
    if "layers.0" in n or "actor.0" in n: print(f"{n:55s} |std-splitk| {d01:.2e}  std vs fp32 {d0:.2e}  splitk vs fp32 {d1:.2e}")
print("worst ratio (splitk err / std err) over params:", worst)
opt = torch.optim.AdamW(agent.parameters(), lr=1e-4, fused=True)
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
print("splitk step ms", t(step))
orig = te._linear; te._linear = lambda x, w, b: torch.nn.functional.linear(x, w, b)
print("standard step ms", t(step)); te._linear = orig
