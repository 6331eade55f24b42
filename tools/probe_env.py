"""Ad-hoc timing probe of the env kernels (not the bench): step kernel at large B, fused rollouts."""
import sys, os, time
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, os.path.join(ROOT, "2048-ppo-agent_amd"))
import numpy as np, torch
from src.g2048 import native as nv
from src.g2048.engine import RolloutEngine

dev = torch.device("cuda:0")
def time_step(B, mode=1, iters=20):
    eng = RolloutEngine(0, mode, dev)
    boards = torch.empty((B,16),dtype=torch.uint8,device=dev); masks=torch.empty(B,dtype=torch.uint8,device=dev)
    done=torch.empty(B,dtype=torch.uint8,device=dev); ep=torch.empty(B,dtype=torch.int32,device=dev)
    nv.reset_fused((1,2),boards,masks,done,ep,B,0,mode)
    actions=torch.randint(0,4,(B,),dtype=torch.int32,device=dev)
    keys=nv.split((5,6),B,mode,dev); rew=torch.empty(B,dtype=torch.float32,device=dev)
    # a few steps in to get mid-game boards
    for _ in range(30): nv.step(boards,masks,done,actions,keys,rew,mode)
    done.zero_()
    s=torch.cuda.Event(enable_timing=True); e=torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); s.record()
    for _ in range(iters): nv.step(boards,masks,done,actions,keys,rew,mode)
    e.record(); torch.cuda.synchronize()
    ms=s.elapsed_time(e)/iters
    print(f"k_step B={B} mode={mode}: {ms*1e3:.1f} us/launch  {B/ms/1e6:.2f} G steps/s  {50*B/ms/1e9:.3f} TB/s (50 B/step) live={(done==0).float().mean().item():.2f}")
for B in (65536, 1<<20, 1<<22, 1<<24): time_step(B)
time_step(1<<24, mode=0)
for B,pol in ((65536,1),(65536,0),(1<<20,1)):
    eng=RolloutEngine(0,1,dev)
    tr=eng.rollout_fused(B,pol,chunk=64)  # warm
    torch.cuda.synchronize(); t=time.time()
    tr=eng.rollout_fused(B,pol,chunk=64); n=tr.num_steps(); torch.cuda.synchronize(); dt=time.time()-t
    print(f"fused rollout B={B} policy={pol}: T={tr.T} live steps={n} {n/dt/1e9:.3f} G env-steps/s wall {dt*1e3:.1f} ms  mean max tile {float((2.0**tr.final_boards.max(1).values.float()).mean()):.1f}")
