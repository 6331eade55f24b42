"""Host-side (pure torch / numpy) pieces of the drop-in API against vectors produced by the reference itself
(tests/golden/torch_reference.npz, made by tests/golden/make_golden_torch.py).  CPU only."""
import os

import numpy as np
import pytest
import torch

from src.env_definitions import ACTION_DIM, BOARD_DIM, BOARD_FLAT_DIM, OBS_DIM
from src.optim import Lamb, configure_bert_optimizers
from src.optim.configure_optimizers import split_decay_groups
from src.ppo.ppo_agent import MLPAgent, PPOAgent
from src.ppo.ppo_trainer import PPOTrainer, _History
from src.ppo.rollout_buffer import RolloutBuffer
from src.ppo.transformer_encoder import PositionalEncoding2D, TransformerEncoder, get_emb
from src.stats import RunningStatsVec

REF = np.load(os.path.join(os.path.dirname(__file__), "golden", "torch_reference.npz"))
SMALL = dict(observation_dim=31, action_dim=4, hidden_dim=48, d_model=32, nhead=4, num_layers=2, dim_feedforward=64,
             dropout=0.1)


def _agent(red):
    agent = PPOAgent(reduction=red, **SMALL).eval()
    sd = {k[len(f"agent_{red}/sd/"):]: torch.from_numpy(REF[k]) for k in REF.files if k.startswith(f"agent_{red}/sd/")}
    assert set(sd) == set(agent.state_dict())  # same parameter AND buffer names as the reference
    agent.load_state_dict(sd)
    return agent


def test_constants():
    assert (OBS_DIM, BOARD_DIM, BOARD_FLAT_DIM, ACTION_DIM) == (31, (4, 4), 16, 4)


@pytest.mark.parametrize("red", ["cls", "mean"])
def test_agent_matches_reference_forward(red):
    agent = _agent(red)
    boards = torch.from_numpy(REF[f"agent_{red}/boards"])
    obs = torch.nn.functional.one_hot(boards.long(), 31).float()
    masks = torch.from_numpy(REF[f"agent_{red}/masks"])
    actions = torch.from_numpy(REF[f"agent_{red}/actions"])
    with torch.no_grad():
        for x in (obs, boards):  # reference one-hot layout and the packed-board fast path
            logits, values = agent(x, None)
            np.testing.assert_allclose(logits.numpy(), REF[f"agent_{red}/logits"], atol=1e-5, rtol=1e-5)
            np.testing.assert_allclose(values.numpy(), REF[f"agent_{red}/values"], atol=1e-5, rtol=1e-5)
            ml, _ = agent(x, masks)
            np.testing.assert_allclose(ml.numpy(), REF[f"agent_{red}/masked_logits"], atol=1e-5, rtol=1e-5)
            lp, v, ent = agent.evaluate_actions(x, actions, masks)
            np.testing.assert_allclose(lp.numpy(), REF[f"agent_{red}/eval_logp"], atol=1e-5, rtol=1e-5)
            np.testing.assert_allclose(ent.numpy(), REF[f"agent_{red}/eval_entropy"], atol=1e-5, rtol=1e-5)


def test_pe_buffer_equals_reference_buffer():
    pe = PositionalEncoding2D(4, 4, 32).pe
    np.testing.assert_allclose(pe.numpy(), REF["agent_cls/sd/transformer.positional_encoding.pe"], atol=1e-7)
    e = get_emb(torch.tensor([[0.0, 1.0]]))
    np.testing.assert_allclose(e.numpy(), [[0.0, 1.0, np.sin(1.0), np.cos(1.0)]], atol=1e-7)


def test_default_agent_size_and_state_dict_families():
    agent = PPOAgent(hidden_dim=512, d_model=256, nhead=8, num_layers=4, dim_feedforward=1024, reduction="cls")
    assert sum(p.numel() for p in agent.parameters()) == 3958272
    keys = set(agent.state_dict())
    for k in ("input_embedding.weight", "transformer.cls_token", "transformer.positional_encoding.inv_freq",
              "transformer.positional_encoding.pe", "transformer.encoder.layers.3.self_attn.in_proj_weight",
              "transformer.encoder.layers.0.norm2.bias", "actor.4.weight", "critic.0.bias"):
        assert k in keys
    assert agent.state_dict()["transformer.positional_encoding.pe"].shape == (1, 4, 4, 256)


def test_agent_behaviour():
    agent = _agent("cls")
    obs = torch.nn.functional.one_hot(torch.randint(0, 12, (7, 16)), 31).float()
    mask = torch.zeros(7, 4, dtype=torch.bool)
    mask[:, 2] = True
    a, lp, v = agent.get_action(obs, mask)
    assert (a == 2).all() and a.shape == (7,) and v.shape == (7, 1)  # single legal action is forced
    lp2, v2, ent = agent.evaluate_actions(obs, a, mask)
    np.testing.assert_allclose(lp.detach().numpy(), lp2.detach().numpy(), atol=1e-6)
    with pytest.raises(RuntimeError):
        agent(torch.zeros(2, 16, 30))  # wrong observation dim
    with pytest.raises(ValueError):
        TransformerEncoder(32, 4, 1, 64)(torch.zeros(1, 16, 32), reduction="max")
    m = MLPAgent(hidden_dim=32, trunk_dim=24).eval()
    boards = torch.randint(0, 12, (5, 16), dtype=torch.uint8)
    l1, v1 = m(boards)
    l2, v2 = m(torch.nn.functional.one_hot(boards.long(), 31).float())
    np.testing.assert_allclose(l1.detach().numpy(), l2.detach().numpy(), atol=1e-5)


def test_ppo_loss_matches_reference():
    agent = _agent("mean")
    tr = PPOTrainer.__new__(PPOTrainer)
    tr.agent, tr.clip_epsilon, tr.value_loss_coef, tr.entropy_coef, tr.use_action_mask = agent, 0.2, 0.5, 0.01, True
    boards = torch.from_numpy(REF["agent_mean/boards"])
    t = lambda k: torch.from_numpy(REF[k])
    with torch.no_grad():
        loss, pl, vl, el, nlp = tr._compute_ppo_loss(boards, t("agent_mean/actions"), t("agent_mean/masks"),
                                                    t("loss/old_logp"), t("loss/adv"), t("loss/ret"))
    for got, key in ((loss, "total"), (pl, "policy"), (vl, "value"), (el, "entropy"), (nlp, "new_logp")):
        np.testing.assert_allclose(got.numpy(), REF[f"loss/{key}"], atol=1e-5, rtol=1e-5)


def test_rollout_buffer_numpy_interface_matches_reference():
    buf = RolloutBuffer(31, 16, 4)
    buf.store_batch(REF["buffer/in_obs"], REF["buffer/in_act"], REF["buffer/in_msk"], REF["buffer/in_rew"],
                    REF["buffer/in_val"], REF["buffer/in_lp"], REF["buffer/in_term"])
    assert buf.buffer_size == int(REF["buffer/size"]) == 20
    got = buf.get_buffer_data()
    for k in ("observations", "actions", "action_masks", "rewards", "values", "log_probs", "terminations"):
        assert got[k].dtype == REF[f"buffer/out_{k}"].dtype
        assert got[k].shape == REF[f"buffer/out_{k}"].shape
        assert (got[k] == REF[f"buffer/out_{k}"]).all()
    buf.store_batch(REF["buffer/in_obs"], REF["buffer/in_act"], REF["buffer/in_msk"], REF["buffer/in_rew"],
                    REF["buffer/in_val"], REF["buffer/in_lp"], REF["buffer/in_term"])
    assert buf.buffer_size == 40 and len(buf.get_buffer_data()["rewards"]) == 40
    buf.reset()
    assert buf.buffer_size == 0 and buf.get_buffer_data()["rewards"].shape == (0,)


def test_rollout_buffer_validation_errors():
    buf = RolloutBuffer(observation_dim=4, observation_length=5, action_dim=2)
    with pytest.raises(ValueError) as e:
        buf._validate_and_reshape_observations(np.zeros((3, 4, 15), np.float32))
    assert "Cannot reshape observations" in str(e.value)
    assert "15 elements per timestep but expected 20 elements" in str(e.value)
    with pytest.raises(ValueError) as e:
        buf._validate_and_reshape_observations(np.zeros(10, np.float32))
    assert "Observations must have at least 2 dimensions" in str(e.value) and "but got shape (10,)" in str(e.value)
    assert buf._validate_and_reshape_observations(np.zeros((3, 4, 20))).shape == (3, 4, 5, 4)
    buf2 = RolloutBuffer(3, (4, 4), 2)
    assert buf2._validate_and_reshape_observations(np.zeros((2, 3, 48))).shape == (2, 3, 4, 4, 3)


def test_weight_decay_groups_match_reference():
    agent = PPOAgent(hidden_dim=512, d_model=256, nhead=8, num_layers=4, dim_feedforward=1024, reduction="cls")
    decay, no_decay = split_decay_groups(agent, ["norm", "embedding"])
    assert decay == REF["optim/decay_names"].tolist() and no_decay == REF["optim/no_decay_names"].tolist()
    assert "transformer.cls_token" in decay and "input_embedding.weight" in no_decay
    d = configure_bert_optimizers(agent, "adamw", 4e-4, (0.9, 0.999), 1e-6, 0.01, 1000, 0.025,
                                  ["constant", "constant"], ["norm", "embedding"])
    g = d["optimizer"].param_groups
    assert [(len(x["params"]), sum(p.numel() for p in x["params"]), x["weight_decay"]) for x in g] == \
        [(23, 3934976, 0.01), (37, 23296, 0.0)]
    assert d["lr_scheduler"]["interval"] == "step"
    with pytest.raises(TypeError):
        configure_bert_optimizers(agent, "sgd", 1e-3, (0.9, 0.999), 1e-6, 0.0, 10, 0.1, ["linear", "cosine"])
    for names in (["linear", "linear"], ["linear", "cosine"], ["constant", "constant"]):
        d = configure_bert_optimizers(torch.nn.Linear(3, 3), "lamb", 1e-3, (0.9, 0.999), 1e-6, 0.01, 100, 0.1, names)
        d["optimizer"].step()
        for _ in range(20):
            d["lr_scheduler"]["scheduler"].step()


def test_lamb_matches_reference_steps():
    w = torch.nn.Parameter(torch.from_numpy(REF["lamb/w0"].copy()))
    b = torch.nn.Parameter(torch.from_numpy(REF["lamb/b0"].copy()))
    opt = Lamb([{"params": [w], "weight_decay": 0.01}, {"params": [b], "weight_decay": 0.0}], lr=1e-2)
    for gw, gb in zip(REF["lamb/gw"], REF["lamb/gb"]):
        w.grad, b.grad = torch.from_numpy(gw.copy()), torch.from_numpy(gb.copy())
        opt.step()
    np.testing.assert_allclose(w.detach().numpy(), REF["lamb/w2"], atol=1e-6, rtol=1e-5)
    np.testing.assert_allclose(b.detach().numpy(), REF["lamb/b2"], atol=1e-6, rtol=1e-5)


def test_running_stats_vec():
    rng = np.random.default_rng(0)
    xs = [rng.normal(size=(3, n)) for n in (5, 1, 17)]
    s = RunningStatsVec()
    assert s.mean == 0.0 and s.std == 0.0
    for x in xs:
        s.push(x)
    full = np.concatenate(xs, axis=1)
    np.testing.assert_allclose(s.mean[:, 0], full.mean(1))
    np.testing.assert_allclose(s.variance[:, 0], full.var(1))
    np.testing.assert_allclose(s.std[:, 0], full.std(1))
    assert s.num_samples[:, 0].tolist() == [23, 23, 23]
    with pytest.raises(ValueError):
        s.push(np.zeros(3))
    s.clear()
    assert s.mean == 0.0


def test_history_slices_like_a_list():
    h = _History(maxlen=5)
    h.extend(range(8))
    assert h[-3:] == [5, 6, 7] and h[0] == 3 and len(h) == 5


def default_shape_agent(dropout=0.1):
    """This repository's PPOAgent at the reference's default shape, with the weights of tests/golden/weights_recipe.py
    (the recipe make_golden_torch.py loaded into the REFERENCE's PPOAgent to produce the ``default/*`` vectors)."""
    import sys

    sys.path.insert(0, os.path.join(os.path.dirname(__file__), "golden"))
    from weights_recipe import fill_state_dict

    agent = PPOAgent(observation_dim=31, action_dim=4, hidden_dim=512, d_model=256, nhead=8, num_layers=4,
                     dim_feedforward=1024, dropout=dropout, reduction="cls").eval()
    sd = agent.state_dict()
    sd.update({k: torch.from_numpy(v) for k, v in fill_state_dict({k: tuple(v.shape) for k, v in sd.items()}).items()})
    agent.load_state_dict(sd)
    return agent


def test_default_shape_agent_and_loss_match_reference():
    """d_model 256 / 8 heads / 4 layers / ff 1024 / "cls": forward, evaluate_actions and the PPO loss of the torch
    restatement against vectors the reference's own modules produced for the same (recipe) weights."""
    agent = default_shape_agent()
    t = lambda k: torch.from_numpy(REF[f"default/{k}"])
    boards, bits, actions = t("boards"), t("mask_bits"), t("actions")
    masks = (bits.unsqueeze(-1) & torch.tensor([1, 2, 4, 8], dtype=torch.uint8)) != 0
    with torch.no_grad():
        np.testing.assert_allclose(agent.features(boards).numpy(), REF["default/features"], atol=2e-5, rtol=1e-5)
        logits, values = agent(boards, None)
        np.testing.assert_allclose(logits.numpy(), REF["default/logits"], atol=1e-5, rtol=1e-5)
        np.testing.assert_allclose(values.numpy(), REF["default/values"], atol=1e-5, rtol=1e-5)
        lp, _, ent = agent.evaluate_actions(boards, actions, masks)
        np.testing.assert_allclose(lp.numpy(), REF["default/eval_logp"], atol=1e-5, rtol=1e-5)
        np.testing.assert_allclose(ent.numpy(), REF["default/eval_entropy"], atol=1e-5, rtol=1e-5)
        tr = PPOTrainer.__new__(PPOTrainer)
        tr.agent, tr.clip_epsilon, tr.value_loss_coef, tr.entropy_coef, tr.use_action_mask = agent, 0.2, 0.5, 0.01, True
        loss, pl, vl, el, nlp = tr._compute_ppo_loss(boards, actions, masks, t("old_logp"), t("adv"), t("ret"))
    for got, key in ((loss, "loss_total"), (pl, "loss_policy"), (vl, "loss_value"), (el, "loss_entropy"), (nlp, "new_logp")):
        np.testing.assert_allclose(got.numpy(), REF[f"default/{key}"], atol=1e-5, rtol=1e-5)


def test_mlp_agent_host_paths_and_graph_gating():
    """Without a HIP device the MLP policy runs the generic forward (the one-hot GEMM written as a gather-sum), the bf16
    rollout path and the captured forward stay off, and the engine keeps its live-board compaction."""
    from src.ppo import MLPAgent, TorchActionFunction

    torch.manual_seed(3)
    agent = MLPAgent()
    boards = torch.randint(0, 12, (9, 16), dtype=torch.uint8)
    assert not agent._rollout_bf16_ok(boards)
    with torch.no_grad():
        logits, values = agent(boards)
        oh = torch.nn.functional.one_hot(boards.long(), 31).float().flatten(1)
        h = torch.relu(agent.trunk_hidden(torch.relu(agent.trunk_in(oh))))
        assert torch.allclose(logits, agent.actor(h), atol=1e-5) and torch.allclose(values, agent.critic(h), atol=1e-5)
        masked, _ = agent(boards, torch.tensor([[1, 0, 1, 1]] * 9))
    assert (masked[:, 1] < -1e7).all() and torch.allclose(masked[:, [0, 2, 3]], logits[:, [0, 2, 3]])
    fn = TorchActionFunction(agent, graph_cache={})
    assert fn.compact is True and fn._graph_cache is None  # CPU: no capture, compaction stays
    lg, vl = fn.policy_fn(boards, None)
    assert torch.allclose(lg, logits, atol=1e-5) and vl.shape == (9,)
    agent.prepare_rollout()  # no-op off the device
    assert agent._trunk_shadow is None and agent._head_shadow is None


class _ShadowOwner(torch.nn.Module):  # (module level: pickle needs an importable class)
    def __init__(self):
        super().__init__()
        from src.ppo.hip_ops import Bf16Shadow

        self.lin = torch.nn.Linear(8, 8)
        self.sh = Bf16Shadow([self.lin.weight, self.lin.bias], transposed=(0,))


def test_bf16_shadow_copies_start_cold():
    """copy.deepcopy / pickle of a module that owns a Bf16Shadow: the copy shadows the COPIED parameters, is registered for
    invalidation, and carries neither the original's buffers nor its maintainer (an optimiser must not travel with a pickled
    agent)."""
    import copy
    import pickle

    from src.ppo.hip_ops import Bf16Shadow

    m = _ShadowOwner()
    views = m.sh()
    assert views[0].dtype == torch.bfloat16 and m.sh.tviews[0].shape == (8, 8)
    m.sh.maintainer = object()  # what FlatAdamWStep.adopt_shadows sets
    for c in (copy.deepcopy(m), pickle.loads(pickle.dumps(m))):
        assert c.sh is not m.sh and c.sh.maintainer is None and c.sh.views is None and c.sh.key is None
        assert c.sh.params[0] is c.lin.weight and c.sh.transposed == (0,) and c.sh in Bf16Shadow._live
        with torch.no_grad():
            c.lin.weight.add_(1.0)
        assert torch.equal(c.sh()[0], c.lin.weight.detach().to(torch.bfloat16)) and not c.sh.maintained()


def test_capture_guard_keeps_the_collector_off_and_restores_it():
    """capture.no_gc_during_capture: collect first, cyclic collector off inside, previous state restored (also on error);
    CollectionsWhileCapturing sees collections (none of them 'while capturing' on a CPU box)."""
    import gc

    from src.ppo.capture import CollectionsWhileCapturing, no_gc_during_capture

    class Node:
        def __init__(self):
            self.me = self

    assert gc.isenabled()
    with CollectionsWhileCapturing() as seen:
        with no_gc_during_capture():
            assert not gc.isenabled()
            before = seen.total  # (the guard's own gc.collect() has been counted)
            assert before >= 1
            junk = [Node() for _ in range(20000)]  # far beyond the generation-0 threshold: would trigger a collection
            del junk
            assert seen.total == before
        assert gc.isenabled()
        with pytest.raises(RuntimeError):
            with no_gc_during_capture():
                raise RuntimeError("x")
        assert gc.isenabled()
        gc.disable()
        try:
            with no_gc_during_capture():
                pass
            assert not gc.isenabled()  # it was off before: stays off
        finally:
            gc.enable()
    assert seen.during_capture == 0


def test_private_torch_interfaces_exist_with_the_expected_shapes(monkeypatch):
    """src/ppo/torch_compat.py: every private PyTorch interface the device update path uses is present in this torch and
    behaves as the callers assume; a missing one produces ONE error that names the torch version and the interface."""
    from src.ppo import torch_compat

    assert torch_compat.problems() == []
    torch_compat.check()
    monkeypatch.setattr(torch_compat, "_checked", False)
    monkeypatch.delattr(torch, "_addmm_activation")
    with pytest.raises(torch_compat.TorchInterfaceError) as e:
        torch_compat.check()
    assert torch.__version__ in str(e.value) and "_addmm_activation" in str(e.value)


def test_rollout_precision_default():
    """mixed_precision bfloat16 implies the bf16 rollout (what the unmodified reference CLI gets); G2048_ROLLOUT_FP32=1 and an
    explicit argument restore the reference's fp32 rollout; the round-2 switch G2048_ROLLOUT_AMP still decides when set."""
    from src.ppo.ppo_trainer import resolve_rollout_amp as r

    assert r(None, "bfloat16", {}) is True
    assert r(None, "bfloat16", {"G2048_ROLLOUT_FP32": "1"}) is False
    assert r(None, None, {}) is False and r(None, "float16", {}) is False
    assert r(False, "bfloat16", {}) is False and r(True, None, {"G2048_ROLLOUT_FP32": "1"}) is True
    assert r(None, "bfloat16", {"G2048_ROLLOUT_AMP": "0"}) is False and r(None, None, {"G2048_ROLLOUT_AMP": "1"}) is True


def test_fixed_horizon_per_episode_reward_statistic():
    """FixedTrajectory.finished_episode_max_rewards: one value per FINISHED episode (aligned with finished_episode_lengths), the
    largest single-step reward of that episode, carried across rollouts for episodes that span several of them -- against a
    per-lane Python walk over two consecutive windows."""
    from src.g2048.engine import FixedTrajectory

    g = torch.Generator().manual_seed(4)
    T, B = 9, 6

    def window():
        rewards = torch.randint(0, 50, (T, B), generator=g).float()
        done = torch.rand(T, B, generator=g) < 0.3
        meta = (done.to(torch.uint8) << 6)
        z = torch.zeros
        return FixedTrajectory(boards=z(T, B, 16, dtype=torch.uint8), meta=meta, rewards=rewards, log_probs=z(T, B), values=z(T, B),
                               final_boards=z(B, 16, dtype=torch.uint8), final_masks=z(B, dtype=torch.uint8),
                               ep_len=z(B, dtype=torch.int32), ep_len_before=z(B, dtype=torch.int32), T=T, B=B), rewards, done

    run = [float("-inf")] * B  # the walk's running maxima
    carry = None
    for _ in range(2):
        traj, rewards, done = window()
        want = []
        for t in range(T):  # row-major over (t, lane): the order of boolean indexing with the [T, B] mask
            for b in range(B):
                run[b] = max(run[b], float(rewards[t, b]))
                if bool(done[t, b]):
                    want.append(run[b])
                    run[b] = float("-inf")
        vals, carry = traj.finished_episode_max_rewards(carry)
        assert vals.tolist() == want and vals.numel() == traj.finished_episode_lengths().numel()
        assert carry.tolist() == run


# ---------------------------------------------------------------------------------------------------------------------
# config trees (run/config_tree.py): the reference's Hydra layout through PyYAML
# ---------------------------------------------------------------------------------------------------------------------
PKG = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "2048-ppo-agent_amd")


def _config_tree():
    import importlib.util

    spec = importlib.util.spec_from_file_location("config_tree", os.path.join(PKG, "run", "config_tree.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_config_tree_composes_defaults_overrides_and_interpolations(tmp_path):
    """A synthetic tree with every form the loader promises: defaults order with _self_ first (groups override the primary file),
    a `# @package _global_` group merged at the root, group re-selection, +group, dotted overrides, ~delete, ${...} interpolation,
    exponent floats without a dot (PyYAML reads `4e-4` as a string)."""
    ct = _config_tree()
    w = lambda rel, text: (os.makedirs(os.path.dirname(tmp_path / rel), exist_ok=True), open(tmp_path / rel, "w").write(text))
    w("main.yaml", "defaults:\n  - _self_\n  - model: small\n  - trainer: default\n  - paths: default\n\ntask_name: t\nseed: null\n"
                   "trainer:\n  gamma: 0.5\n")
    w("model/small.yaml", "d_model: 64\nnhead: 4\n")
    w("model/big.yaml", "d_model: 256\nnhead: 8\n")
    w("trainer/default.yaml", "gamma: 0.99\noptim:\n  max_lr: 4e-4\n  eps: 1e-6\n  betas:\n    - 0.9\n    - 0.999\nresume_from_checkpoint: null\n")
    w("paths/default.yaml", "work_dir: ${hydra:runtime.cwd}\nmodels_dir: ${paths.work_dir}/../.models\n")
    w("experiment/resume.yaml", "# @package _global_\n\ntask_name: resumed\ntrainer:\n  optim:\n    max_lr: 1e-5\n"
                                "  resume_from_checkpoint: ${paths.models_dir}/best.pt\n")
    c = ct.compose(str(tmp_path / "main.yaml"))
    assert c["model"] == {"d_model": 64, "nhead": 4} and c["seed"] is None and c["task_name"] == "t"
    assert c["trainer"]["gamma"] == 0.99, "_self_ comes first: the trainer group overrides the primary file's value"
    assert c["trainer"]["optim"] == {"max_lr": 4e-4, "eps": 1e-6, "betas": [0.9, 0.999]}
    assert c["paths"]["work_dir"] == os.getcwd() and c["paths"]["models_dir"] == os.getcwd() + "/../.models"
    c = ct.compose(str(tmp_path / "main.yaml"), ["model=big", "+experiment=resume", "trainer.gamma=0.9", "trainer.optim.eps=1e-8",
                                                 "extra.deep.key=[1, 2]", "~task_name"])
    assert c["model"]["d_model"] == 256 and c["trainer"]["gamma"] == 0.9 and c["trainer"]["optim"]["max_lr"] == 1e-5
    assert c["trainer"]["optim"]["eps"] == 1e-8 and c["trainer"]["optim"]["betas"] == [0.9, 0.999]
    assert c["trainer"]["resume_from_checkpoint"] == os.getcwd() + "/../.models/best.pt"
    assert c["extra"]["deep"]["key"] == [1, 2] and "task_name" not in c and "experiment" not in c
    with pytest.raises(FileNotFoundError):
        ct.compose(str(tmp_path / "main.yaml"), ["model=missing"])
    # a file without a defaults list (this repository's flattened config) passes through, overrides applied
    flat = ct.compose(os.path.join(PKG, "configs", "train_ppo_agent.yaml"), ["trainer.rollout_batch_size=64", "model.kind=mlp"])
    assert flat["trainer"]["rollout_batch_size"] == 64 and flat["model"]["kind"] == "mlp" and flat["trainer"]["optim"]["max_lr"] == 4e-4


def test_reference_config_tree_loads_unmodified():
    """The reference's own configs/ directory (configs/train_ppo_agent.yaml:5-17) through the loader: the composed values are the
    ones this repository's flattened file states, and they build the default agent and the trainer's argument set."""
    ref = "/root/reference/configs/train_ppo_agent.yaml"
    if not os.path.exists(ref):
        pytest.skip("the reference tree is only present in the build container")
    ct = _config_tree()
    c = ct.compose(ref)
    flat = ct.compose(os.path.join(PKG, "configs", "train_ppo_agent.yaml"))
    assert c["seed"] is None and c["data"]["seed"] == 42 and c["task_name"] == "train_transformer_combined"
    assert c["model"] == {k: v for k, v in flat["model"].items() if k != "kind"}
    assert c["trainer"] == flat["trainer"]
    assert isinstance(c["trainer"]["optim"]["max_lr"], float) and isinstance(c["trainer"]["optim"]["eps"], float)
    assert "hydra" in c and "paths" in c
    r = ct.compose(ref, ["+experiment=resume_train_ppo_agent", "trainer.rollout_batch_size=65536"])
    assert r["trainer"]["entropy_coef"] == 0.0001 and r["trainer"]["optim"]["max_lr"] == 1e-5 and r["trainer"]["optim"]["eps"] == 1e-6
    assert r["trainer"]["resume_from_checkpoint"].endswith("/.models/current_best.pt") and "${" not in r["trainer"]["resume_from_checkpoint"]
    assert r["trainer"]["rollout_batch_size"] == 65536 and r["task_name"] == "resume_train_transformer_combined"
    from src.ppo import PPOAgent

    m = dict(c["model"])
    m.pop("observation_length")
    assert sum(p.numel() for p in PPOAgent(**m).parameters()) == 3958272
