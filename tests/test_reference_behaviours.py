"""Behaviours the reference's own unit tests pin for the pure-torch pieces (tests/ppo/test_ppo_agent.py,
test_transformer_encoder.py, test_masking_grad.py), re-expressed against the drop-in classes.  CPU only."""
import numpy as np
import pytest
import torch

from src.ppo.ppo_agent import MLPAgent, PPOAgent
from src.ppo.transformer_encoder import PositionalEncoding2D, TransformerEncoder, get_emb

CFG = dict(hidden_dim=64, d_model=64, nhead=4, num_layers=2, dim_feedforward=128, dropout=0.1)


@pytest.mark.parametrize("reduction", ["mean", "cls"])
@pytest.mark.parametrize("batch", [1, 3, 32])
def test_forward_shapes(reduction, batch):
    agent = PPOAgent(reduction=reduction, **CFG)
    obs = torch.randn(batch, 16, 31)
    logits, values = agent(obs)
    assert logits.shape == (batch, 4) and values.shape == (batch, 1)
    a, lp, v = agent.get_action(obs, torch.ones(batch, 4, dtype=torch.bool))
    assert a.shape == (batch,) and lp.shape == (batch,) and v.shape == (batch, 1)
    assert ((a >= 0) & (a < 4)).all() and (lp <= 0).all()
    lp2, v2, ent = agent.evaluate_actions(obs, a)
    assert lp2.shape == (batch,) and ent.shape == (batch,) and (ent >= 0).all()


def test_eval_mode_is_deterministic_and_train_mode_uses_dropout():
    agent = PPOAgent(**CFG)
    obs = torch.randn(4, 16, 31)
    agent.eval()
    a = agent(obs)[0]
    assert torch.equal(a, agent(obs)[0])
    agent.train()
    assert not torch.equal(agent(obs)[0], agent(obs)[0])


def test_state_dict_round_trip_and_parameter_count():
    a, b = PPOAgent(**CFG), PPOAgent(**CFG)
    b.load_state_dict(a.state_dict())
    a.eval(), b.eval()
    obs = torch.randn(2, 16, 31)
    assert torch.equal(a(obs)[0], b(obs)[0])
    assert sum(p.numel() for p in PPOAgent().parameters()) == sum(p.numel() for p in PPOAgent(reduction="cls").parameters())


def test_mask_semantics_and_finite_gradients():
    """tests/ppo/test_masking_grad.py: the -1e8 idiom gives finite grads and p(masked) < 1e-6."""
    agent = PPOAgent(dropout=0.0, **{k: v for k, v in CFG.items() if k != "dropout"})
    obs = torch.randn(8, 16, 31)
    mask = torch.tensor([[1, 0, 1, 0]] * 8, dtype=torch.bool)
    logits, values = agent(obs, mask)
    probs = torch.softmax(logits, -1)
    assert (probs[:, [1, 3]] < 1e-6).all() and torch.allclose(probs.sum(-1), torch.ones(8))
    lp, v, ent = agent.evaluate_actions(obs, torch.zeros(8, dtype=torch.long), mask)
    (-(lp.mean()) + v.pow(2).mean() - 0.01 * ent.mean()).backward()
    for n, p in agent.named_parameters():
        assert p.grad is not None and torch.isfinite(p.grad).all(), n
    raw, _ = agent(obs, None)
    assert torch.allclose(raw[:, [0, 2]], logits[:, [0, 2]])  # legal logits are untouched by masking


def test_cls_only_last_layer_equals_full_layer():
    """The CLS-only shortcut of the last layer returns exactly the CLS row of the full computation."""
    enc = TransformerEncoder(64, 4, 3, 128, dropout=0.0).eval()
    x = torch.randn(5, 16, 64)
    full = enc.positional_encoding.forward_flat(x)
    full = torch.cat([enc.cls_token.expand(5, -1, -1), full], 1)
    layers = enc.encoder.layers
    h = layers[0].norm1(full)
    for i, layer in enumerate(layers):
        full, h = enc._layer(layer, full, h, layers[i + 1].norm1 if i + 1 < len(layers) else None, cls_only=False)
    np.testing.assert_allclose(enc(x, reduction="cls").detach().numpy(), full[:, 0].detach().numpy(), atol=1e-5)
    np.testing.assert_allclose(enc(x, reduction="mean").detach().numpy(), full[:, 1:].mean(1).detach().numpy(), atol=1e-5)
    ref = enc.encoder(torch.cat([enc.cls_token.expand(5, -1, -1), enc.positional_encoding.forward_flat(x)], 1))
    np.testing.assert_allclose(full.detach().numpy(), ref.detach().numpy(), atol=1e-5)  # == nn.TransformerEncoder


def test_positional_encoding():
    pe = PositionalEncoding2D(4, 4, 32, dropout=0.0)
    assert pe.pe.shape == (1, 4, 4, 32) and pe.inv_freq.shape == (8,)
    x = torch.zeros(2, 4, 4, 32)
    assert torch.allclose(pe(x)[0], pe.pe[0])
    assert torch.allclose(pe.forward_flat(torch.zeros(2, 16, 32))[0], pe.pe.reshape(16, 32))
    inds = torch.tensor([[0, 5, 15]])
    assert torch.allclose(pe.forward_with_inds(torch.zeros(1, 3, 32), inds)[0], pe.pe.reshape(16, 32)[[0, 5, 15]])
    # row code in the first half of the channels, column code in the second half
    assert torch.allclose(pe.pe[0, 2, 0, :16], pe.pe[0, 2, 3, :16]) and torch.allclose(pe.pe[0, 0, 1, 16:], pe.pe[0, 3, 1, 16:])
    s = get_emb(torch.tensor([[0.5]]))
    assert torch.allclose(s, torch.tensor([[np.sin(0.5), np.cos(0.5)]], dtype=torch.float32))


def test_mlp_agent_interface():
    m = MLPAgent(hidden_dim=32, trunk_dim=48)
    obs = torch.randn(6, 16, 31)
    a, lp, v = m.get_action(obs, torch.ones(6, 4, dtype=torch.bool))
    lp2, v2, ent = m.evaluate_actions(obs, a, torch.ones(6, 4, dtype=torch.bool))
    assert torch.allclose(lp, lp2, atol=1e-6) and v.shape == (6, 1)
    boards = torch.randint(0, 12, (6, 16), dtype=torch.uint8)
    m.train()  # training path uses the one-hot GEMM, inference the gather: same numbers
    tr = m(boards)[0]
    m.eval()
    with torch.no_grad():
        np.testing.assert_allclose(tr.detach().numpy(), m(boards)[0].numpy(), atol=1e-5)
