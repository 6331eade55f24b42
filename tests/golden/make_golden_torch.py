"""Golden vectors from the reference's PyTorch-side modules (run in the build container only).

Imports the reference's pure-torch/numpy modules from /root/reference (read-only; nothing is copied) and
stores inputs + outputs as data in tests/golden/torch_reference.npz:
  * PPOAgent forward / evaluate_actions for a small fixed state-dict      (src/ppo/ppo_agent.py)
  * the same + the PPO loss for the DEFAULT model shape (d 256, 8 heads, 4 layers, ff 1024, "cls") with the weights of
    tests/golden/weights_recipe.py (outputs only: the 15.8 MB state-dict is rebuilt from the recipe by the tests)
  * PPODataset GAE (raw and z-scored)                                     (src/ppo/data_loader.py:103-130,61-67)
  * PPOTrainer._compute_ppo_loss components                               (src/ppo/ppo_trainer.py:251-314)
  * RolloutBuffer.store_batch -> get_buffer_data                          (src/ppo/rollout_buffer.py:128-206)
  * weight-decay group membership of configure_bert_optimizers            (src/optim/configure_optimizers.py:120-213)
  * two Lamb steps                                                        (src/optim/lamb.py:106-209)
The reference's src/ppo/__init__.py also imports modules that need jax/torch2jax/tensorboard (absent here);
those names are registered as empty placeholder modules only so that the package imports -- none of the
functions exercised below touches them.
"""
import os
import sys
import types

import numpy as np
import torch

sys.dont_write_bytecode = True
REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))


def _placeholder(name, **attrs):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    sys.modules[name] = m
    return m


class _NoWriter:
    def __init__(self, *a, **k):
        pass


_placeholder("torch.utils.tensorboard", SummaryWriter=_NoWriter)
jax = _placeholder("jax", Array=object)
jax.numpy = _placeholder("jax.numpy")
_placeholder("torch2jax", t2j=lambda x: x)
_placeholder("pgx", State=object)
sys.path.insert(0, REF)

from src.optim import Lamb, configure_bert_optimizers  # noqa: E402
from src.ppo.data_loader import PPODataset  # noqa: E402
from src.ppo.ppo_agent import PPOAgent  # noqa: E402
from src.ppo.ppo_trainer import PPOTrainer  # noqa: E402
from src.ppo.rollout_buffer import RolloutBuffer  # noqa: E402

out = {}
rng = np.random.default_rng(0)
torch.manual_seed(0)

# ---- agent forward ------------------------------------------------------------------------------
cfg = dict(observation_dim=31, action_dim=4, hidden_dim=48, d_model=32, nhead=4, num_layers=2, dim_feedforward=64,
           dropout=0.1)
for red in ("cls", "mean"):
    agent = PPOAgent(reduction=red, **cfg).eval()
    boards = rng.integers(0, 12, size=(6, 16)).astype(np.uint8)
    obs = torch.nn.functional.one_hot(torch.from_numpy(boards).long(), 31).float()
    masks = torch.tensor(rng.integers(0, 2, size=(6, 4)).astype(bool))
    masks[:, 0] = True
    actions = torch.tensor(rng.integers(0, 4, size=6))
    with torch.no_grad():
        logits, values = agent(obs, None)
        mlogits, _ = agent(obs, masks)
        lp, v2, ent = agent.evaluate_actions(obs, actions, masks)
    for k, t in agent.state_dict().items():
        out[f"agent_{red}/sd/{k}"] = t.numpy()
    out[f"agent_{red}/boards"] = boards
    out[f"agent_{red}/masks"] = masks.numpy()
    out[f"agent_{red}/actions"] = actions.numpy()
    out[f"agent_{red}/logits"] = logits.numpy()
    out[f"agent_{red}/values"] = values.numpy()
    out[f"agent_{red}/masked_logits"] = mlogits.numpy()
    out[f"agent_{red}/eval_logp"] = lp.numpy()
    out[f"agent_{red}/eval_entropy"] = ent.numpy()

# ---- PPO loss (agent in eval mode so dropout is off) ----------------------------------------------
tr = PPOTrainer.__new__(PPOTrainer)
tr.agent, tr.clip_epsilon, tr.value_loss_coef, tr.entropy_coef, tr.use_action_mask = agent, 0.2, 0.5, 0.01, True
old_lp = lp + torch.tensor(rng.normal(0, 0.3, size=6), dtype=torch.float32)
adv = torch.tensor(rng.normal(0, 1, size=6), dtype=torch.float32)
ret = torch.tensor(rng.normal(0, 1, size=6), dtype=torch.float32)
with torch.no_grad():
    loss, pl, vl, el, nlp = tr._compute_ppo_loss(obs, actions, masks, old_lp, adv, ret)
out["loss/old_logp"], out["loss/adv"], out["loss/ret"] = old_lp.numpy(), adv.numpy(), ret.numpy()
out["loss/total"], out["loss/policy"], out["loss/value"] = loss.numpy(), pl.numpy(), vl.numpy()
out["loss/entropy"], out["loss/new_logp"] = el.numpy(), nlp.numpy()

# ---- GAE ----------------------------------------------------------------------------------------
for i, (gamma, lam, N) in enumerate([(0.99, 0.95, 300), (0.9, 0.8, 57), (1.0, 1.0, 40)]):
    r = rng.normal(0, 3, size=N).astype(np.float32)
    v = rng.normal(0, 1, size=N).astype(np.float32)
    term = rng.random(N) < 0.06
    term[-1] = i != 2  # case 2 ends without a termination flag
    data = dict(observations=np.zeros((N, 16, 31), np.float32), actions=np.zeros((N, 4), np.float32),
                action_masks=np.ones((N, 4), bool), rewards=r, values=v, log_probs=np.zeros(N, np.float32),
                terminations=term)
    ds = PPODataset(data, gamma=gamma, lambda_gae=lam)
    raw_adv, raw_ret = ds._compute_gae_returns()
    out[f"gae{i}/params"] = np.array([gamma, lam])
    out[f"gae{i}/rewards"], out[f"gae{i}/values"], out[f"gae{i}/terms"] = r, v, term
    out[f"gae{i}/raw_adv"], out[f"gae{i}/raw_ret"] = raw_adv.numpy(), raw_ret.numpy()
    out[f"gae{i}/adv"], out[f"gae{i}/ret"] = ds.advantages.numpy(), ds.returns.numpy()

# ---- rollout buffer -----------------------------------------------------------------------------
B, T = 5, 9
obs_bt = rng.integers(0, 2, size=(B, T, 4, 4, 31)).astype(bool)
act_bt = rng.random((B, T, 4)).astype(np.float32)
msk_bt = rng.integers(0, 2, size=(B, T, 4)).astype(bool)
rew_bt, val_bt, lp_bt = (rng.normal(size=(B, T)).astype(np.float32) for _ in range(3))
term_bt = np.zeros((B, T), bool)
term_bt[0, 3] = term_bt[0, 6] = term_bt[1, 8] = term_bt[3, 0] = term_bt[4, 5] = True  # env 2 never terminates
buf = RolloutBuffer(31, 16, 4)
buf.store_batch(obs_bt, act_bt, msk_bt, rew_bt, val_bt, lp_bt, term_bt)
got = buf.get_buffer_data()
for k, a in dict(obs=obs_bt, act=act_bt, msk=msk_bt, rew=rew_bt, val=val_bt, lp=lp_bt, term=term_bt).items():
    out[f"buffer/in_{k}"] = a
for k, a in got.items():
    out[f"buffer/out_{k}"] = a
out["buffer/size"] = np.array(buf.buffer_size)

# ---- optimizer groups (default model + default trainer.optim config) -------------------------------
full = PPOAgent(hidden_dim=512, d_model=256, nhead=8, num_layers=4, dim_feedforward=1024, reduction="cls")
opt = configure_bert_optimizers(full, "adamw", 4e-4, (0.9, 0.999), 1e-6, 0.01, 500000, 0.025,
                                ["constant", "constant"], ["norm", "embedding"])["optimizer"]
ids = {id(p): n for n, p in full.named_parameters()}
out["optim/decay_names"] = np.array([ids[id(p)] for p in opt.param_groups[0]["params"]])
out["optim/no_decay_names"] = np.array([ids[id(p)] for p in opt.param_groups[1]["params"]])

# ---- Lamb ---------------------------------------------------------------------------------------
w0 = rng.normal(size=(7, 5)).astype(np.float32)
b0 = rng.normal(size=(5,)).astype(np.float32)
g = [(rng.normal(size=(7, 5)).astype(np.float32) * 3, rng.normal(size=(5,)).astype(np.float32)) for _ in range(2)]
w, b = torch.nn.Parameter(torch.tensor(w0)), torch.nn.Parameter(torch.tensor(b0))
lamb = Lamb([{"params": [w], "weight_decay": 0.01}, {"params": [b], "weight_decay": 0.0}], lr=1e-2)
for gw, gb in g:
    w.grad, b.grad = torch.tensor(gw), torch.tensor(gb)
    lamb.step()
out["lamb/w0"], out["lamb/b0"] = w0, b0
out["lamb/gw"], out["lamb/gb"] = np.stack([x[0] for x in g]), np.stack([x[1] for x in g])
out["lamb/w2"], out["lamb/b2"] = w.detach().numpy(), b.detach().numpy()

# ---- default-shape agent (configs/model/transformer_combined.yaml), weights from weights_recipe -----------------------
sys.path.insert(0, HERE)
from weights_recipe import fill_state_dict, sample_boards  # noqa: E402

big = PPOAgent(observation_dim=31, action_dim=4, hidden_dim=512, d_model=256, nhead=8, num_layers=4,
               dim_feedforward=1024, dropout=0.1, reduction="cls").eval()
sd = big.state_dict()
sd.update({k: torch.from_numpy(v) for k, v in fill_state_dict({k: tuple(v.shape) for k, v in sd.items()}).items()})
big.load_state_dict(sd)
M = 48
rng2 = np.random.default_rng(2)  # own stream: the older vectors below stay byte-identical
bboards = sample_boards(M)
bobs = torch.nn.functional.one_hot(torch.from_numpy(bboards).long(), 31).float()
bmask_bits = rng2.integers(1, 16, size=M).astype(np.uint8)
bmasks = torch.tensor((bmask_bits[:, None] >> np.arange(4)) & 1, dtype=torch.bool)
bactions = torch.tensor([int(rng2.choice(np.flatnonzero(m))) for m in bmasks.numpy()])
with torch.no_grad():
    blogits, bvalues = big(bobs, None)
    blp, _, bent = big.evaluate_actions(bobs, bactions, bmasks)
    bfeat = big.transformer(big.input_embedding(bobs), reduction="cls")
trb = PPOTrainer.__new__(PPOTrainer)
trb.agent, trb.clip_epsilon, trb.value_loss_coef, trb.entropy_coef, trb.use_action_mask = big, 0.2, 0.5, 0.01, True
bold = blp + torch.tensor(rng2.normal(0, 0.3, size=M), dtype=torch.float32)
bold[::6] = blp[::6]
badv = torch.tensor(rng2.normal(0, 1, size=M), dtype=torch.float32)
badv[::5] = 0.0
bret = torch.tensor(rng2.normal(0, 1, size=M), dtype=torch.float32)
with torch.no_grad():
    bloss, bpl, bvl, bel, bnlp = trb._compute_ppo_loss(bobs, bactions, bmasks, bold, badv, bret)
out["default/boards"], out["default/mask_bits"], out["default/actions"] = bboards, bmask_bits, bactions.numpy()
out["default/features"], out["default/logits"], out["default/values"] = bfeat.numpy(), blogits.numpy(), bvalues.numpy()
out["default/eval_logp"], out["default/eval_entropy"] = blp.numpy(), bent.numpy()
out["default/old_logp"], out["default/adv"], out["default/ret"] = bold.numpy(), badv.numpy(), bret.numpy()
out["default/loss_total"], out["default/loss_policy"], out["default/loss_value"] = bloss.numpy(), bpl.numpy(), bvl.numpy()
out["default/loss_entropy"], out["default/new_logp"] = bel.numpy(), bnlp.numpy()

np.savez_compressed(os.path.join(HERE, "torch_reference.npz"), **out)
print("wrote", len(out), "arrays")
