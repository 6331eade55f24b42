import sys, os, time
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, os.path.join(ROOT, "2048-ppo-agent_amd"))
import torch
from src.ppo import PPOAgent
dev = torch.device("cuda:0")
torch.manual_seed(0)
agent = PPOAgent(hidden_dim=512, d_model=256, nhead=8, num_layers=4, dim_feedforward=1024, reduction="cls").to(dev).eval()
def t(fn, n=5):
    fn(); torch.cuda.synchronize(); s=time.time()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.time()-s)/n
for B in (4096, 16384, 65536):
    boards = torch.randint(0, 12, (B, 16), dtype=torch.uint8, device=dev)
    with torch.no_grad():
        f32 = t(lambda: agent(boards))
        with torch.autocast("cuda", dtype=torch.bfloat16):
            bf = t(lambda: agent(boards))
    print(f"B={B}: fwd fp32 {f32*1e3:.2f} ms ({B*110e6/f32/1e12:.1f} TF/s)  bf16 {bf*1e3:.2f} ms ({B*110e6/bf/1e12:.1f} TF/s)")
# train step at 2048
agent.train()
opt = torch.optim.AdamW(agent.parameters(), lr=1e-4)
for M in (2048, 8192, 32768):
    boards = torch.randint(0, 12, (M, 16), dtype=torch.uint8, device=dev); acts = torch.randint(0,4,(M,),device=dev)
    def step():
        with torch.autocast("cuda", dtype=torch.bfloat16):
            lp, v, ent = agent.evaluate_actions(boards, acts)
            loss = (-lp.mean() + v.float().pow(2).mean() - 0.01*ent.mean())
        opt.zero_grad(); loss.backward(); opt.step()
    dt = t(step)
    print(f"train M={M}: {dt*1e3:.2f} ms/step ({M*330e6/dt/1e12:.1f} TF/s)")
from torch.profiler import profile, ProfilerActivity
agent.eval(); boards = torch.randint(0, 12, (65536, 16), dtype=torch.uint8, device=dev)
with torch.no_grad(), torch.autocast("cuda", dtype=torch.bfloat16):
    with profile(activities=[ProfilerActivity.CUDA]) as prof:
        agent(boards); torch.cuda.synchronize()
print(prof.key_averages().table(sort_by="cuda_time_total", row_limit=14, max_name_column_width=60))
