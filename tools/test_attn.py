import sys, os, time
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, os.path.join(ROOT, "2048-ppo-agent_amd")); sys.path.insert(0, ROOT)
import torch, torch.nn.functional as F
from src.ppo.hip_ops import _AttnPacked, _AttnCls
dev = torch.device("cuda:0"); torch.manual_seed(0)
H, hd, S = 8, 32, 17
def ref_packed(qkv):
    B = qkv.shape[0]
    q, k, v = qkv.view(B, S, 3, H, hd).unbind(2)
    return F.scaled_dot_product_attention(q.transpose(1,2), k.transpose(1,2), v.transpose(1,2)).transpose(1,2).reshape(B, S, H*hd)
for B in (1, 5, 2048):
    qkv = (torch.randn(B, S, 3*H*hd, device=dev) * 1.5).to(torch.bfloat16).requires_grad_(True)
    qkv32 = qkv.detach().float().requires_grad_(True)
    g = torch.randn(B, S, H*hd, device=dev).to(torch.bfloat16)
    o = _AttnPacked.apply(qkv, H, 0.0); o.backward(g)
    o32 = ref_packed(qkv32); o32.backward(g.float())
    qkvb = qkv.detach().clone().requires_grad_(True); ob = ref_packed(qkvb); ob.backward(g)
    e = lambda a, b: ((a.float()-b.float()).norm()/b.float().norm()).item()
    print(f"packed B={B}: fwd err vs fp32 {e(o,o32):.2e} (sdpa bf16 {e(ob,o32):.2e});  grad err {e(qkv.grad,qkv32.grad):.2e} (sdpa bf16 {e(qkvb.grad,qkv32.grad):.2e})")
    # cls variant
    q = (torch.randn(B, 1, H*hd, device=dev)).to(torch.bfloat16).requires_grad_(True)
    kv = (torch.randn(B, S, 2*H*hd, device=dev)).to(torch.bfloat16).requires_grad_(True)
    gc = torch.randn(B, 1, H*hd, device=dev).to(torch.bfloat16)
    oc = _AttnCls.apply(q, kv, H, 0.0); oc.backward(gc)
    q32 = q.detach().float().requires_grad_(True); kv32 = kv.detach().float().requires_grad_(True)
    k32, v32 = kv32.view(B, S, 2, H, hd).unbind(2)
    oc32 = F.scaled_dot_product_attention(q32.view(B,1,H,hd).transpose(1,2), k32.transpose(1,2), v32.transpose(1,2)).transpose(1,2).reshape(B,1,H*hd)
    oc32.backward(gc.float())
    print(f"cls    B={B}: fwd err {e(oc,oc32):.2e}  dq err {e(q.grad,q32.grad):.2e}  dkv err {e(kv.grad,kv32.grad):.2e}")
# dropout statistics: E[o] unchanged, keep fraction
B = 4096
qkv = torch.randn(B, S, 3*H*hd, device=dev).to(torch.bfloat16)
o0 = _AttnPacked.apply(qkv.clone().requires_grad_(True), H, 0.0).float()
acc = torch.zeros_like(o0)
for _ in range(20): acc += _AttnPacked.apply(qkv.clone().requires_grad_(True), H, 0.1).float()
print("dropout mean preserved: rel err of 20-sample average", ((acc/20 - o0).norm()/o0.norm()).item())
# gradient consistency with dropout: finite-difference-free check: backward uses the same mask -> d(sum(o*g))/dv equals P~^T g; test linearity in v
x = qkv.clone().requires_grad_(True); torch.manual_seed(5); o1 = _AttnPacked.apply(x, H, 0.3); (o1.float()*g[:B].float() if False else o1.float()).sum().backward()
print("dropout grad finite:", torch.isfinite(x.grad).all().item())
def t(fn, n=30):
    for _ in range(5): fn()
    torch.cuda.synchronize(); s=time.time()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.time()-s)/n*1e6
B = 2048
qkv = torch.randn(B, S, 3*H*hd, device=dev).to(torch.bfloat16).requires_grad_(True); g = torch.randn(B, S, H*hd, device=dev).to(torch.bfloat16)
def mine():
    o = _AttnPacked.apply(qkv, H, 0.1); o.backward(g); qkv.grad = None
def sdpa():
    q, k, v = qkv.view(B, S, 3, H, hd).unbind(2)
    o = F.scaled_dot_product_attention(q.transpose(1,2), k.transpose(1,2), v.transpose(1,2), dropout_p=0.1).transpose(1,2).reshape(B, S, H*hd); o.backward(g); qkv.grad = None
print(f"fwd+bwd at B=2048: custom {t(mine):.0f} us, sdpa {t(sdpa):.0f} us")
