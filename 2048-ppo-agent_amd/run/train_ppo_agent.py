#!/usr/bin/env python3
"""Train a PPO agent on the MI355X engine -- the reference's run/train_ppo_agent.py:19-138 without Hydra.

    python run/train_ppo_agent.py [--config configs/train_ppo_agent.yaml] [key=value ...]
    python run/train_ppo_agent.py --config /path/to/reference/configs/train_ppo_agent.yaml +experiment=resume_train_ppo_agent
    torchrun --nproc-per-node 8 run/train_ppo_agent.py trainer.rollout_batch_size=524288 trainer.rollout_batches=1

``--config`` is either this repository's flattened file or the primary file of a Hydra-style config TREE -- the reference's own
``configs/`` directory works unmodified: its ``defaults`` list (configs/train_ppo_agent.yaml:5-11) is composed by
``run/config_tree.py`` (PyYAML only), with Hydra's command-line forms ``key.sub=value``, ``group=option``, ``+group=option``.
Overrides use dotted keys (``trainer.total_timesteps=2000000 model.kind=mlp``).  Multi-GPU: one rank per GPU,
``rollout_batch_size`` is the global number of envs.  (The reference's own script also runs unmodified against this
package when Hydra is installed: see INTEGRATION.md.)
"""
import argparse
import json
import logging
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

from src.ppo import MLPAgent, PPOAgent, PPOTrainer, RolloutBuffer  # noqa: E402
from src.runs import BatchRunner, evaluate_agent  # noqa: E402

logging.basicConfig(level=logging.INFO)
logger = logging.getLogger("train_ppo_agent")


def load_config(path, overrides=()):
    """Flattened file or config tree + command-line overrides -> dict (``config_tree.compose``)."""
    if HERE not in sys.path:
        sys.path.insert(0, HERE)
    from config_tree import compose

    return compose(path, overrides)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", default=os.path.join(os.path.dirname(HERE), "configs", "train_ppo_agent.yaml"))
    ap.add_argument("--eval-episodes", type=int, default=0, help="greedy masked evaluation after training")
    ap.add_argument("overrides", nargs="*")
    args = ap.parse_args()
    cfg = load_config(args.config, args.overrides)

    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    device = torch.device("cuda", local_rank)
    torch.cuda.set_device(device)
    if world > 1:
        dist.init_process_group("nccl", device_id=device)
    # the reference: cfg.get("seed", cfg.data.seed) (run/train_ppo_agent.py:40-42) -- its tree has a top-level `seed: null`, which
    # OmegaConf returns as None: no torch seed and BatchRunner seed 0; `data.seed` only counts when the top-level key is absent
    seed = cfg["seed"] if "seed" in cfg else (cfg.get("data") or {}).get("seed")
    if seed is not None:
        torch.manual_seed(seed)
    m = dict(cfg["model"])
    kind = m.pop("kind", "transformer")
    m.pop("observation_length", None)
    if kind == "mlp":
        agent = MLPAgent(observation_dim=m["observation_dim"], action_dim=m["action_dim"], hidden_dim=m["hidden_dim"])
    else:
        agent = PPOAgent(**m)
    logger.info("Created %s agent with %d parameters", kind, sum(p.numel() for p in agent.parameters()))
    t = dict(cfg["trainer"])
    runner = BatchRunner(init_seed=seed if seed is not None else 0, act_fn=None, rng_mode=cfg.get("rng_mode"),
                         device=device)
    buf = RolloutBuffer(cfg["model"]["observation_dim"], cfg["model"]["observation_length"], cfg["model"]["action_dim"])
    trainer = PPOTrainer(
        agent=agent, batch_runner=runner, rollout_buffer=buf, optimizer_param_dict=t["optim"], max_steps=t["max_steps"],
        gamma=t["gamma"], lambda_gae=t["lambda_gae"], clip_epsilon=t["clip_epsilon"],
        value_loss_coef=t["value_loss_coef"], entropy_coef=t["entropy_coef"], max_grad_norm=t["max_grad_norm"],
        target_kl=t["target_kl"], use_action_mask=t["use_action_mask"], device=device,
        mixed_precision=t["mixed_precision"], max_samples_per_epoch=t["max_samples_per_epoch"],
        shuffle_on_reset=t["shuffle_on_reset"], rollout_amp=t.get("rollout_amp"))  # None: bf16 rollout when mixed_precision is bfloat16 (G2048_ROLLOUT_FP32=1: fp32)
    if t.get("resume_from_checkpoint"):
        if not os.path.exists(t["resume_from_checkpoint"]):
            raise FileNotFoundError(f"Checkpoint file not found: {t['resume_from_checkpoint']}")
        trainer.load_checkpoint(t["resume_from_checkpoint"])
    trainer.train(total_timesteps=t["total_timesteps"], rollout_batch_size=t["rollout_batch_size"],
                  rollout_batches=t["rollout_batches"], update_epochs=t["update_epochs"],
                  train_batch_size=t["train_batch_size"], save_freq=t["save_freq"],
                  resume_extend_steps=t["resume_extend_steps"])
    if trainer.episode_rewards:
        tail = trainer.episode_rewards[-100:]
        logger.info("Final mean episode reward (last 100 episodes): %.2f", sum(tail) / len(tail))
    if args.eval_episodes and trainer.rank == 0:
        print(json.dumps(evaluate_agent(agent, device, args.eval_episodes)))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
