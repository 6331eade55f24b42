"""Launch g2048_step repeatedly at a saturating size (for rocprofv3)."""
import sys, os
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, os.path.join(ROOT, "2048-ppo-agent_amd"))
import torch
from src.g2048 import native as nv
dev = torch.device("cuda:0"); B = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 24; mode = 1
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 10
boards = torch.empty((B,16),dtype=torch.uint8,device=dev); masks=torch.empty(B,dtype=torch.uint8,device=dev)
done=torch.empty(B,dtype=torch.uint8,device=dev); ep=torch.empty(B,dtype=torch.int32,device=dev)
nv.reset_fused((1,2),boards,masks,done,ep,B,0,mode)
actions=torch.randint(0,4,(B,),dtype=torch.int32,device=dev); keys=nv.split((5,6),B,mode,dev); rew=torch.empty(B,dtype=torch.float32,device=dev)
for _ in range(iters): nv.step(boards,masks,done,actions,keys,rew,mode)
torch.cuda.synchronize()
