import os, sys
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, os.path.join(ROOT, "2048-ppo-agent_amd")); sys.path.insert(0, ROOT)
import torch
from torch.amp import autocast
from src.g2048 import native as nv
dev = torch.device("cuda:0"); torch.manual_seed(0)
M = 2048
logits = (torch.randn(M, 4, device=dev) * 0.05).to(torch.bfloat16).requires_grad_(True)
values = (torch.randn(M, 1, device=dev) * 0.1).to(torch.bfloat16).requires_grad_(True)
actions = torch.randint(0, 4, (M,), device=dev)
adv, ret = torch.randn(M, device=dev), torch.randn(M, device=dev)
with torch.no_grad():
    old_lp = torch.distributions.Categorical(logits=logits.float()).log_prob(actions) + 1e-3 * torch.randn(M, device=dev)
with autocast(device_type="cuda", dtype=torch.bfloat16):
    d = torch.distributions.Categorical(logits=logits, validate_args=False)
    nlp, ent = d.log_prob(actions), d.entropy()
    ratio = torch.exp(nlp - old_lp)
    pl = -torch.min(ratio * adv, torch.clamp(ratio, 0.8, 1.2) * adv)
    vl = torch.nn.functional.mse_loss(values.flatten(), ret, reduction="none")
    loss = (pl + 0.5 * vl + 0.01 * (-ent)).mean()
print("dtypes", nlp.dtype, ent.dtype, vl.dtype, loss.dtype)
(loss * 65536.0).backward()
got_lp, sums, dl, dv = nv.ppo_loss(logits.detach(), values.detach().reshape(-1), actions.to(torch.uint8), None, old_lp, adv, ret, 0.2, 0.5, 0.01)
dl = dl * 65536.0; dv = dv * 65536.0
rel = lambda a, b: ((a.float() - b.float()).norm() / b.float().norm()).item()
print("loss", loss.item(), sums[3].item(), "rel dlogits", rel(dl, logits.grad), "rel dvalues", rel(dv.view_as(values.grad), values.grad))
diff = (dl.float() - logits.grad.float()).abs()
i = diff.view(-1).argmax().item() // 4
print("worst row", i, dl[i], logits.grad[i], "adv", adv[i].item(), "ratio", ratio[i].item())
print("frac elements differing", (diff > 0).float().mean().item())
