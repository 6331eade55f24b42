import sys, os, time
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, os.path.join(ROOT, "2048-ppo-agent_amd")); sys.path.insert(0, ROOT)
import torch, torch.nn.functional as F
dev = torch.device("cuda:0"); torch.manual_seed(0)
T = 2048 * 17
def t(fn, n=30):
    for _ in range(5): fn()
    torch.cuda.synchronize(); s=time.time()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.time()-s)/n*1e6
for (N, K) in ((768, 256), (256, 256), (1024, 256), (256, 1024)):
    dy = torch.randn(T, N, device=dev, dtype=torch.bfloat16); x = torch.randn(T, K, device=dev, dtype=torch.bfloat16)
    base = t(lambda: dy.t() @ x)
    res = [f"N={N} K={K}: dy^T@x {base:.0f}us"]
    ref = (dy.t().float() @ x.float())
    for S in (4, 8, 16, 32):
        def f():
            p = torch.bmm(dy.view(S, T // S, N).transpose(1, 2), x.view(S, T // S, K))
            return p.float().sum(0)
        us = t(f); err = ((f() - ref).abs().max() / ref.abs().max()).item()
        res.append(f"S={S}: {us:.0f}us (err {err:.1e})")
    try:
        def g():
            return torch.bmm(dy.view(8, T // 8, N).transpose(1, 2), x.view(8, T // 8, K), out_dtype=torch.float32).sum(0)
        res.append(f"S=8 f32out: {t(g):.0f}us")
    except Exception as e:
        res.append("f32out n/a " + str(e)[:40])
    e0 = (((dy.t() @ x).float() - ref).abs().max() / ref.abs().max()).item()
    print("  ".join(res), f" base err {e0:.1e}")
    # forward / dX GEMMs for reference
    w = torch.randn(N, K, device=dev, dtype=torch.bfloat16)
    print(f"    fwd x@w^T {t(lambda: x @ w.t()):.0f}us   dX dy@w {t(lambda: dy @ w):.0f}us   bias-grad sum {t(lambda: dy.sum(0)):.0f}us  float-sum {t(lambda: dy.float().sum(0)):.0f}us  ones-mv {t(lambda: torch.mv(dy.t(), torch.ones(T, device=dev, dtype=torch.bfloat16))):.0f}us")
