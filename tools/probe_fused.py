import sys, os, time
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, os.path.join(ROOT, "2048-ppo-agent_amd"))
import numpy as np, torch
from src.g2048 import native as nv
dev = torch.device("cuda:0"); mode = 1
def run(B, policy, chunk, fill):
    key, subs = nv.chain_keys(np.array([0, 0], np.uint32), 1 + 2 * 1024, mode)
    boards = torch.empty((B,16),dtype=torch.uint8,device=dev); masks=torch.empty(B,dtype=torch.uint8,device=dev)
    done=torch.empty(B,dtype=torch.uint8,device=dev); ep=torch.empty(B,dtype=torch.int32,device=dev)
    nv.reset_fused(subs[0],boards,masks,done,ep,B,0,mode)
    cap=1024
    trb=torch.empty((cap,B,16),dtype=torch.uint8,device=dev); trm=torch.empty((cap,B),dtype=torch.uint8,device=dev)
    trr=torch.empty((cap,B),dtype=torch.float32,device=dev); trl=torch.empty((cap,B),dtype=torch.float32,device=dev)
    live=torch.zeros(1,dtype=torch.int32,device=dev)
    t=0; times=[]
    while t+chunk<=cap:
        live.zero_()
        s=torch.cuda.Event(enable_timing=True); e=torch.cuda.Event(enable_timing=True)
        s.record()
        nv.rollout_fused(subs[1+2*t:1+2*(t+chunk)].reshape(chunk,4),t,boards,masks,done,ep,trb,trm,trr,trl,B,0,policy,fill,mode,live)
        e.record(); torch.cuda.synchronize(); times.append(s.elapsed_time(e)); t+=chunk
        if int(live.item())==0: break
    n=int(ep.sum().item())
    print(f"B={B} policy={policy} chunk={chunk} fill={fill}: T={t} kernel ms per chunk {[round(x,3) for x in times]} total {sum(times):.2f} ms -> {n/sum(times)/1e6:.3f} G live steps/s")
for B in (65536, 1<<20):
    for pol in (0,1):
        run(B,pol,64,False)
run(65536,1,16,False); run(65536,1,64,True); run(65536,0,64,True)
