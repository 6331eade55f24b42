"""Two ranks sharing the one GPU of the test box (gloo backend on CUDA tensors): the trainer's sharded collect +
all-reduced update end to end.  (RCCL itself needs one GPU per rank; the driver's scaling run covers that.)"""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, results):
    import sys

    root = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
    sys.path.insert(0, os.path.join(root, "2048-ppo-agent_amd"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from src.ppo import PPOAgent, PPOTrainer, RolloutBuffer
        from src.runs import BatchRunner

        dev = torch.device("cuda:0")
        torch.manual_seed(1000 + rank)  # different init per rank: must be overwritten by rank 0's weights
        agent = PPOAgent(hidden_dim=32, d_model=32, nhead=4, num_layers=1, dim_feedforward=64, dropout=0.0)
        optim = dict(opt_name="adamw", max_lr=4e-4, betas=(0.9, 0.999), eps=1e-6, weight_decay=0.01,
                     warmup_steps_ratio=0.025, scheduler_names=["constant", "constant"],
                     blacklist_weight_modules=["norm", "embedding"])
        tr = PPOTrainer(agent, BatchRunner(init_seed=0, device=dev), RolloutBuffer(31, 16, 4), optim, max_steps=100,
                        use_action_mask=True, device=dev, mixed_precision="bfloat16", target_kl=0.25,
                        max_samples_per_epoch=2000, shuffle_on_reset=True, log_dir=f"/tmp/g2048_dist_{rank}")
        for it in range(2):
            tr.collect_rollouts(64, 1)  # 64 envs globally -> 32 per rank
            assert tr.batch_runner.env0 == 32 * rank and tr.batch_runner.total_envs == 64
            local = torch.tensor([tr.rollout_buffer.buffer_size], dtype=torch.int64, device=dev)
            dist.all_reduce(local)
            assert int(local.item()) == tr.last_rollout_stats["timesteps"]
            keys = [torch.zeros(2, dtype=torch.int64) for _ in range(world)]
            dist.all_gather(keys, torch.from_numpy(tr.batch_runner.key.astype("int64")))
            assert torch.equal(keys[0], keys[1]), "host key chains diverged between ranks"
            m = tr.update_policy(batch_size=128, n_epochs=2)
            assert m["n_updates"] > 0 and all(v == v for v in m.values())
            flat = torch.cat([p.detach().flatten() for p in agent.parameters()]).cpu()
            both = [torch.zeros_like(flat) for _ in range(world)]
            dist.all_gather(both, flat)
            assert torch.equal(both[0], both[1]), "parameters diverged between ranks"
            assert torch.isfinite(flat).all()
        results[rank] = (tr.total_timesteps, tr.total_update_steps)
    finally:
        dist.destroy_process_group()


def test_two_ranks_on_one_gpu(dev):
    world, port = 2, _free_port()
    with mp.Manager() as mgr:
        results = mgr.dict()
        mp.spawn(_worker, args=(world, port, results), nprocs=world, join=True)
        r = dict(results)
        assert r[0] == r[1] and r[0][0] > 0
