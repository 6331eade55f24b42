#!/usr/bin/env python3
"""bench.py -- env-steps/sec of the 2048 rollout + PPO engine on MI355X.

    python bench.py --gpus N --steps K --warmup W
    (N > 1: launched by torch.distributed.run, one rank per GPU, backend nccl = RCCL)

Workload (BASELINE.json configs[2] at N=1, configs[3] at N=8): 65 536 parallel boards PER GPU, the reference's
default Transformer policy (3.96 M parameters) in bf16, random-init weights, boards generated on the device by
self-play from seed 0.  One "step" = one complete PPO iteration in the reference's own mode: lock-step rollout of
65 536 x N complete episodes with the policy in the loop (fused sample + board step + trajectory write), GAE scan +
HIP compaction, then the clipped-surrogate update (5 epochs x up to 300 000 samples, minibatch 2048, AdamW, grad-clip,
one flat gradient all-reduce per minibatch when N > 1) -- the reference's configs/trainer/default.yaml.
`value` = live env-steps gathered by all ranks in the K timed steps / wall time (max over ranks).

Also on the JSON line (measured AFTER the timed region; none of it is part of `value`):
  hip_graph        whether the update really replayed its captured hipGraph (false = the slower eager fallback ran)
  frozen_policy    the same iteration with the learning rate at 0: does not depend on how far the policy has trained
  fixed_horizon    the throughput mode (per-lane auto-reset, fixed horizon, bootstrapped GAE): full PPO iterations
  value_frozen_policy  = frozen_policy.value, on the top level because `value` itself grows with --steps (the policy learns
                   during the run and episodes get longer)
  drop_in_default  collect phase as an unmodified run/train_ppo_agent.py gets it: PPOTrainer(rollout_amp=None) with the
                   reference's mixed_precision: bfloat16 -> bf16 rollout through the fused encoder
  drop_in_fp32_rollout  collect phase of the reference's own fp32 rollout forward (G2048_ROLLOUT_FP32=1), bounded sample
  roofline         the board-step kernel (g2048_step, 50 algorithmic bytes per env-step) at a saturating launch of
                   2^24 boards, HIP events on its launch stream; `valu` = the same launch against the vector-ALU issue peak
  in_loop          the fused policy-step kernel as it ran inside the timed region (HIP events around every launch)
  policy_step      the same kernel at a saturating launch (2^22 boards)
  policy_encoder   the fused bf16 MFMA Transformer-encoder kernels used for rollout inference, against the dense MFMA peak
  env_only         fused random-policy rollout of the same 65 536 boards (no network): env-steps/sec
  mlp4096          BASELINE.json configs[1] (4 096 boards, MLP policy, full PPO loop), N = 1 only
  cpu_baseline     the C oracle (OpenMP) on bounded samples: B in {128, 4096, 65536} x threads in {1, all} (rank 0, N=1)
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, "2048-ppo-agent_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec (MI355X_MICROARCH.md); measured copy ceiling is ~6290 GB/s
STEP_BYTES = 50  # board 16 r + 16 w, action 4, key 8, reward 4, mask 1, done 1
POLICY_STEP_BYTES = 93  # fused policy step: state 18 r + 18 w (+ ep_len 4 r/w), logits 16, value 4, trajectory 29
POLICY_STEP_VALU_PER_WAVE = 1150  # k_policy_step<partitionable, AUTO=0> (ISA count, NOTES.md 3)
STEP_VALU_PER_WAVE = 554  # vector instructions per wave of k_step<partitionable> (ISA count = SQ_INSTS_VALU / SQ_WAVES)
VALU_PEAK_GINSTR = 1024 * 2.4 / 2  # 1024 SIMDs x 2.4 GHz / 2 cycles per wave64 instruction on a SIMD-32 = 1228.8 G/s
# What k_step's OWN instruction mix can issue: its 554 vector instructions by opcode (hipcc -save-temps; NOTES.md 3) priced with the
# per-opcode issue cost measured on this chip at 8 waves per SIMD (profiles/round1_valu_rate_microbench.txt: add / xor / and ~2.7-2.8
# cycles per wave64 instruction, every three-operand, shift, permute or bit-field opcode 4.3-4.6):
#   threefry x4: 80 alignbit x 4.52 + 78 xor x 2.72 + 95 add x 2.77 + 27 add3 x 4.59 + 20 xad x 4.56          = 1 052 cycles
#   board logic: 46 bitop3 + 45 cndmask + 25 perm + 35 shifts + 10 bfe + 9 or3 at 4.3-4.6, 12 and at 2.7, ~72 others at ~3.5 = 1 044 cycles
# = 2 096 cycles per wave = 3.78 cycles per instruction -> 1024 SIMDs x 2.4 GHz / 3.78 = 650 G wave-instructions/s.
STEP_MIX_CYCLES_PER_INSTR = 2096.0 / 554.0
VALU_MIX_PEAK_GINSTR = 1024 * 2.4 / STEP_MIX_CYCLES_PER_INSTR

TRAINER_CFG = dict(gamma=0.99, lambda_gae=0.95, clip_epsilon=0.2, value_loss_coef=0.5, entropy_coef=0.01,
                   max_grad_norm=0.5, target_kl=0.25, use_action_mask=True, mixed_precision="bfloat16",
                   max_samples_per_epoch=300000, shuffle_on_reset=True)
OPTIM_CFG = dict(opt_name="adamw", max_lr=4e-4, betas=(0.9, 0.999), eps=1e-6, weight_decay=0.01,
                 warmup_steps_ratio=0.025, scheduler_names=["constant", "constant"],
                 blacklist_weight_modules=["norm", "embedding"])
MODEL_CFG = dict(observation_dim=31, action_dim=4, hidden_dim=512, d_model=256, nhead=8, num_layers=4,
                 dim_feedforward=1024, dropout=0.1, reduction="cls")


def _events():
    return torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)


def _mid_game_state(dev, B):
    """B boards 24 DRUL moves into their games, with keys and legal actions: inputs of the step-kernel microbenchmarks."""
    from src.g2048 import native as nv

    boards = torch.empty((B, 16), dtype=torch.uint8, device=dev)
    masks = torch.empty(B, dtype=torch.uint8, device=dev)
    done = torch.empty(B, dtype=torch.uint8, device=dev)
    ep = torch.empty(B, dtype=torch.int32, device=dev)
    rew = torch.empty(B, dtype=torch.float32, device=dev)
    nv.reset_fused((1, 2), boards, masks, done, ep, B, 0, nv.RNG_PARTITIONABLE)
    keys = nv.split((5, 6), B, nv.RNG_PARTITIONABLE, dev)
    actions = torch.empty(B, dtype=torch.int32, device=dev)
    for _ in range(24):
        nv.act_drul(masks, actions)
        nv.step(boards, masks, done, actions, keys, rew, nv.RNG_PARTITIONABLE)
    nv.act_drul(masks, actions)
    return boards, masks, done, ep, rew, keys, actions


def step_kernel_roofline(dev, boards_per_launch: int, launches: int = 20):
    """g2048_step at a saturating size, HIP events on the stream it is launched on."""
    from src.g2048 import native as nv

    B = boards_per_launch
    boards, masks, done, ep, rew, keys, actions = _mid_game_state(dev, B)
    live = float((done == 0).float().mean().item())
    stream = torch.cuda.current_stream()
    start, end = _events()
    torch.cuda.synchronize()
    start.record(stream)
    for _ in range(launches):
        nv.step(boards, masks, done, actions, keys, rew, nv.RNG_PARTITIONABLE)
    end.record(stream)
    torch.cuda.synchronize()
    us = start.elapsed_time(end) * 1e3 / launches
    gbs = STEP_BYTES * B / (us * 1e-6) / 1e9
    # HBM bytes per launch: NOT measured in this run (PMC counters need rocprofv3 around the process).  The number is the
    # committed PMC pass (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, FETCH_SIZE x2 gfx950 correction) of the same kernel,
    # scaled to this launch size; `traffic_source` says so.  null if that profile is not in the tree
    traffic, traffic_source = None, None
    try:
        pmc = json.load(open(os.path.join(ROOT, "profiles", "step_kernel_pmc_latest.json")))
        traffic = round(pmc["hbm_bytes_per_env_step"] * B)
        traffic_source = f"profiles/step_kernel_pmc_latest.json ({pmc.get('measured_in', 'round 1')}, re-measured whenever g2048.hip changes; not measured in this run)"
    except Exception:
        pass
    ginstr = STEP_VALU_PER_WAVE * (B / 64) / (us * 1e-6) / 1e9
    # the same kernel with the reference's RNG schedule stubbed out (diagnostic build, tools/step_rng_floor.py): what the board
    # logic alone reaches; committed measurement, not repeated in this run
    rng_floor = {}
    try:
        fl = json.load(open(os.path.join(ROOT, "profiles", "round3_step_rng_floor.json")))
        rng_floor = {"rng_floor_frac": fl["rng_floor_frac"],
                     "rng_floor_source": f"profiles/round3_step_rng_floor.json: k_step with -DG2048_RNG_STUB (no threefry, same bytes) "
                                         f"{fl['rng_stub']['launch_us']} us vs {fl['real']['launch_us']} us for the real kernel at "
                                         f"{fl['boards_per_launch']} boards; the 4 threefry2x32 blocks per step that the bit-exact "
                                         f"jax.random stream prescribes are the difference"}
    except Exception:
        pass
    return {"bound": "hbm", "achieved": round(gbs, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": round(gbs / HBM_PEAK_GBS, 4), "traffic": traffic, "traffic_source": traffic_source,
            # what actually binds this kernel: vector-ALU issue (PMC: SQ_WAIT_INST_ANY 72 % of wave cycles, HBM traffic = the
            # algorithmic bytes); `binding_frac` = issued wave-instructions per second against what the kernel's own opcode mix
            # can issue on 1024 SIMDs at 2.4 GHz (VALU_MIX_PEAK_GINSTR above; the chip holds ~2.1 GHz under this load)
            "binding_bound": "valu", "binding_achieved": round(ginstr, 1), "binding_peak": round(VALU_MIX_PEAK_GINSTR, 1),
            "binding_unit": "G wave-instr/s", "binding_frac": round(ginstr / VALU_MIX_PEAK_GINSTR, 4),
            "binding_peak_source": "554 instructions per wave priced per opcode with profiles/round1_valu_rate_microbench.txt "
                                   f"({STEP_MIX_CYCLES_PER_INSTR:.2f} cycles per instruction for this mix; bench.py header)",
            "kernel": "k_step (g2048_step)",
            "boards_per_launch": B, "launch_us": round(us, 2), "algorithmic_bytes_per_env_step": STEP_BYTES,
            "live_fraction": round(live, 3), **rng_floor,
            "valu": {"instr_per_wave": STEP_VALU_PER_WAVE, "achieved_Ginstr_s": round(ginstr, 1),
                     "peak_Ginstr_s": round(VALU_PEAK_GINSTR, 1), "frac": round(ginstr / VALU_PEAK_GINSTR, 4),
                     "what": "wave64 vector instructions issued per second vs 1024 SIMDs x 2.4 GHz / 2 cycles; the kernel's mix "
                             "(3-operand and shift/permute opcodes) measures 4.0-4.5 cycles per instruction "
                             "(profiles/round1_valu_rate_microbench.txt), i.e. ~0.5 of this peak is its issue ceiling"}}


def policy_step_saturated(dev, B: int = 1 << 22, launches: int = 10):
    """g2048_policy_step (the kernel the training loop runs) at a saturating launch."""
    from src.g2048 import native as nv

    boards, masks, done, ep, rew, keys, actions = _mid_game_state(dev, B)
    logits = torch.randn((B, 4), device=dev)
    values = torch.randn(B, device=dev)
    tr = dict(b=torch.empty((1, B, 16), dtype=torch.uint8, device=dev), m=torch.empty((1, B), dtype=torch.uint8, device=dev),
              r=torch.empty((1, B), device=dev), l=torch.empty((1, B), device=dev), v=torch.empty((1, B), device=dev))
    live = torch.zeros(1, dtype=torch.int32, device=dev)
    stream = torch.cuda.current_stream()
    start, end = _events()

    def launch(counter):
        nv.policy_step((3, 4), (5, 6), logits, values, True, True, 0, boards, masks, done, ep, tr["b"], tr["m"], tr["r"], tr["l"],
                       tr["v"], B, 0, True, nv.RNG_PARTITIONABLE, counter)

    def timed(counter):
        launch(counter)
        torch.cuda.synchronize()
        start.record(stream)
        for _ in range(launches):
            launch(counter)
        end.record(stream)
        torch.cuda.synchronize()
        return start.elapsed_time(end) * 1e3 / launches

    us, us_counting = timed(None), timed(live)
    gbs = POLICY_STEP_BYTES * B / (us * 1e-6) / 1e9
    ginstr = POLICY_STEP_VALU_PER_WAVE * (B / 64) / (us * 1e-6) / 1e9
    return {"kernel": "k_policy_step (g2048_policy_step)", "boards_per_launch": B, "launch_us": round(us, 2),
            "launch_us_counting_live": round(us_counting, 2),
            "algorithmic_bytes_per_env_step": POLICY_STEP_BYTES, "achieved_GBps": round(gbs, 1),
            "frac_of_hbm_peak": round(gbs / HBM_PEAK_GBS, 4),
            "binding_bound": "valu", "binding_achieved": round(ginstr, 1), "binding_peak": round(VALU_MIX_PEAK_GINSTR, 1),
            "binding_unit": "G wave-instr/s", "binding_frac": round(ginstr / VALU_MIX_PEAK_GINSTR, 4),
            "what": "launch_us: as 7 of 8 lock-steps of the rollout loop run it (no live counter); launch_us_counting_live: the "
                    "launch the host polls, one atomic per workgroup with live lanes (rounds 1-3: one per wave to one address, "
                    "which serialised at ~12 ns each and WAS the 757 us this object reported at 2^22 boards); ~1 150 vector "
                    "instructions per wave (8 threefry blocks, 8 f32 logs, exp / log of the log-softmax), priced with k_step's mix"}


def policy_encoder_roofline(agent, dev, boards: int, launches: int = 5):
    """The fused bf16 MFMA encoder kernels (g2048_policy_encoder) at the workload's batch, HIP events on their stream."""
    from src.ppo import fused_policy

    if not fused_policy.supports(agent):
        return None
    fp = fused_policy.FusedPolicy(agent)
    x = torch.randint(0, 12, (boards, 16), dtype=torch.uint8, device=dev)
    fp.features(x)
    stream = torch.cuda.current_stream()
    start, end = _events()
    torch.cuda.synchronize()
    start.record(stream)
    for _ in range(launches):
        fp.features(x)
    end.record(stream)
    torch.cuda.synchronize()
    ms = start.elapsed_time(end) / launches
    layers = fp.n_layers
    # algorithmic = what the "cls" reduction needs: every token through layers 0..L-2; in the last layer K/V of every
    # token but Q, attention, out-proj and feed-forward of the CLS token only (the kernel pair computes exactly that)
    full = 17 * 2 * (256 * 768 + 256 * 256 + 2 * 256 * 1024) + 8 * 2 * 2 * 17 * 17 * 32
    last = 17 * 2 * 256 * 512 + 2 * (256 * 256 + 256 * 256 + 2 * 256 * 1024) + 8 * 2 * 2 * 17 * 32
    flop_per_board = (layers - 1) * full + last
    tf = flop_per_board * boards / (ms * 1e-3) / 1e12
    return {"bound": "mfma", "achieved": round(tf, 1), "peak": 2500.0, "unit": "TFLOP/s", "frac": round(tf / 2500.0, 4),
            "kernel": "k_encoder_main<HEAD> + k_encoder_tail (g2048_policy_encoder, CLS-only last layer)",
            "boards_per_launch": boards, "launch_ms": round(ms, 3), "algorithmic_flop_per_board": flop_per_board,
            "flop_per_board_all_tokens_all_layers": layers * full, "dtype": "bf16 in / f32 accumulate"}


class _TimedPolicyStep:
    """Wraps native.policy_step with HIP events on the launch stream to time the kernel inside the timed region."""

    def __init__(self, nv):
        self.nv, self.orig, self.events, self.on = nv, nv.policy_step, [], False

    def __call__(self, *a, **k):
        if not self.on:
            return self.orig(*a, **k)
        s, e = _events()
        st = torch.cuda.current_stream()
        s.record(st)
        self.orig(*a, **k)
        e.record(st)
        self.events.append((s, e))

    def mean_us(self):
        return float(np.mean([s.elapsed_time(e) for s, e in self.events])) * 1e3 if self.events else None


def _cpu_model():
    model, phys = "unknown", set()
    try:
        pid = cid = None
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name") and model == "unknown":
                model = line.split(":", 1)[1].strip()
            elif line.startswith("physical id"):
                pid = line.split(":", 1)[1].strip()
            elif line.startswith("core id"):
                cid = line.split(":", 1)[1].strip()
            elif not line.strip():
                if pid is not None and cid is not None:
                    phys.add((pid, cid))
                pid = cid = None
    except OSError:
        pass
    return model, len(phys) or None


def cpu_baseline(boards: int, total_seconds: float = 18.0):
    """C oracle (OpenMP over boards, random policy, whole episodes, env + RNG + policy draw, no network) on bounded
    samples: B in {128, 4096, boards} x threads in {1, all}.  The headline cell (`value`) is boards x all threads."""
    from oracle import c_oracle as orc
    from oracle import g2048_oracle as npo

    all_threads = orc.num_threads()
    orc.rollout(npo.key(99), 1024, 0, 1024, 1, 1)  # warm (build, page-in)
    cells = [(b, t) for b in (128, 4096, boards) for t in (1, all_threads)]
    budget = {c: total_seconds * (0.4 if c == (boards, all_threads) else 0.6 / (len(cells) - 1)) for c in cells}
    matrix, sample = [], None
    for b, t in cells:
        orc.set_num_threads(t)
        # a single thread steps a bounded slice of the batch per repetition (envs [0, n) of the b-board batch: the per-env
        # work does not depend on the batch size), all threads step the whole batch
        n = b if t > 1 else min(b, 1024)
        t0 = time.time()
        steps = reps = 0
        while time.time() - t0 < budget[(b, t)] and reps < 4096:
            steps += orc.rollout(npo.key(reps), b, 0, n, 1, 1)["total_steps"]
            reps += 1
        dt = time.time() - t0
        matrix.append({"boards": b, "threads": t, "value": round(steps / dt, 1), "episodes": reps * n, "seconds": round(dt, 2),
                       **({"slice": f"envs 0..{n - 1} of the {b}-board batch"} if n != b else {})})
        if (b, t) == (boards, all_threads):
            sample = (reps, steps, dt)
    orc.set_num_threads(all_threads)
    model, phys = _cpu_model()
    head = matrix[-1]
    return {"value": head["value"], "unit": "env-steps/sec", "cores": all_threads, "kind": "port",
            "sample": f"{sample[0]} x {boards} random-policy episodes ({sample[1]} env-steps, {sample[2]:.1f} s) on {all_threads} OpenMP "
                      f"threads; env + RNG + policy draw only (no policy network); C restatement of the Pgx 2048 env "
                      f"(oracle/g2048_oracle.c), bit-exact vs the reference's golden frames",
            "unpinned_semantics": "no reference artifact pins three env rules the oracle (and the kernels) restate from Pgx: the merge-reward "
                                  "magnitude, -1 on an illegal action, the spawn on a full board going to cell 0 (NOTES.md 4); rewards "
                                  "feed GAE and the loss",
            "cpu_model": model, "logical_cpus": os.cpu_count(), "physical_cores": phys, "matrix": matrix}


def main():
    if os.environ.get("G2048_BENCH_STACKS"):  # debugging aid: dump every thread's stack to stderr every N seconds
        import faulthandler

        faulthandler.dump_traceback_later(int(os.environ["G2048_BENCH_STACKS"]), repeat=True)
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", default="transformer65536", choices=["transformer65536", "mlp4096"],
                    help="transformer65536 = BASELINE.json configs[2] (the headline); mlp4096 = configs[1]: 4096 boards, "
                         "MLP policy (flattened one-hot -> 512 -> 512 trunk + the reference's heads), full PPO loop")
    ap.add_argument("--boards", type=int, default=None, help="parallel boards per GPU (default: per workload)")
    ap.add_argument("--train-batch", type=int, default=2048)
    ap.add_argument("--epochs", type=int, default=5)
    ap.add_argument("--horizon", type=int, default=128, help="rows per lane of the fixed_horizon extra")
    ap.add_argument("--roofline-boards", type=int, default=1 << 24)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X; there is no CPU path")
    # G2048_BENCH_DRYRUN=1: rehearse the multi-rank control flow on ONE GPU (all ranks on cuda:0, gloo backend);
    # the numbers of such a run mean nothing and the JSON line says so
    dryrun = os.environ.get("G2048_BENCH_DRYRUN") == "1"
    dev = torch.device("cuda", 0 if dryrun else local_rank)
    torch.cuda.set_device(dev)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if dryrun:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    if args.boards is None:
        args.boards = 65536 if args.workload == "transformer65536" else 4096
    from src.actions import act_randomly
    from src.g2048 import native as nv
    from src.ppo import MLPAgent, PPOAgent, PPOTrainer, RolloutBuffer, TorchActionFunction
    from src.runs import BatchRunner

    timed = _TimedPolicyStep(nv)
    nv.policy_step = timed
    global_boards = args.boards * world

    def make_trainer(workload, seed=0, optim=OPTIM_CFG, **kw):
        torch.manual_seed(seed)
        agent = PPOAgent(**MODEL_CFG) if workload == "transformer65536" else MLPAgent(hidden_dim=512, trunk_dim=512)
        runner = BatchRunner(init_seed=0, rng_mode="partitionable", device=dev)
        cfg = dict(TRAINER_CFG, **kw)
        return PPOTrainer(agent, runner, RolloutBuffer(31, 16, 4), optim, max_steps=500000, device=dev, rollout_amp=True,
                          log_dir=os.path.join("/tmp", f"g2048_bench_logs_{rank}"), **cfg)

    def sync():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def run_iterations(trainer, boards_global, n_warm, n_timed, phase=None):
        """n_timed full PPO iterations -> (env_steps, seconds (max over ranks), per-iteration metrics of the last update)."""
        m = {}

        def one(record):
            t_a = time.perf_counter()
            trainer.collect_rollouts(boards_global, 1)
            n = trainer.last_rollout_stats["timesteps"]
            if record and phase is not None:
                torch.cuda.synchronize()
                t_b = time.perf_counter()
            m.update(trainer.update_policy(batch_size=args.train_batch, n_epochs=args.epochs))
            if record and phase is not None:
                torch.cuda.synchronize()
                phase["collect_s"] += t_b - t_a
                phase["update_s"] += time.perf_counter() - t_b
            return n

        for _ in range(n_warm):
            one(False)
        sync()
        if phase is not None:
            timed.on = True
        t0 = time.perf_counter()
        steps = 0
        for _ in range(n_timed):
            steps += one(True)
        sync()
        elapsed = time.perf_counter() - t0
        timed.on = False
        if world > 1:
            t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            elapsed = float(t.item())
        return steps, elapsed, m

    # ------------------------------------------------------------------ the timed region (headline)
    trainer = make_trainer(args.workload)
    agent = trainer.agent
    phase = {"collect_s": 0.0, "update_s": 0.0}
    env_steps, elapsed, last_metrics = run_iterations(trainer, global_boards, args.warmup, args.steps, phase)

    if not all(bool(torch.isfinite(p).all()) for p in agent.parameters()):
        raise SystemExit("non-finite policy parameters after the timed region: the measurement is void")
    hip_graph = bool(last_metrics.get("hip_graph", False))
    out = {
        "metric": "env-steps/sec at N parallel boards (full PPO loop: rollout with policy in the loop + GAE + update)",
        "value": round(env_steps / elapsed, 1), "unit": "env-steps/sec", "n_gpus": world, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": round(1e3 * elapsed / max(args.steps, 1), 2),
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u8 boards / bf16 policy",
        "data": "synthetic (self-play from seed 0, random-init policy weights)",
        "config": {"workload": (f"{args.boards} parallel boards per GPU, Transformer policy bf16 (3.96 M params), "
                                f"full PPO iteration (BASELINE.json configs[2]; configs[3] when n_gpus=8)"
                                if args.workload == "transformer65536" else
                                f"{args.boards} parallel boards per GPU, MLP policy bf16 "
                                f"({sum(p.numel() for p in agent.parameters())} params), full PPO iteration "
                                f"(BASELINE.json configs[1])"),
                   "boards_per_gpu": args.boards, "global_boards": global_boards, "train_batch": args.train_batch,
                   "update_epochs": args.epochs, "max_samples_per_epoch": TRAINER_CFG["max_samples_per_epoch"],
                   "max_samples_per_epoch_is": "global (each rank draws its 1/N share), as the reference's single subset",
                   "max_samples_per_epoch_per_rank": trainer.per_rank_samples_per_epoch(),
                   "rollout_mode": "episodes (reference lock-step semantics)",
                   "rng_mode": "partitionable", "parallelism": f"env-shard x{world} + 1 grad all-reduce/minibatch"},
        "hip_graph": hip_graph, "hip_graphs_captured": int(last_metrics.get("hip_graphs_captured", 0)),
        "env_steps_per_ppo_iteration": int(env_steps / max(args.steps, 1)),
        "phase_seconds_per_step": {k: round(v / max(args.steps, 1), 3) for k, v in phase.items()},
        **({"DRYRUN": "all ranks shared one GPU over gloo: control-flow rehearsal, not a measurement"} if dryrun else {}),
        "update_minibatches_per_step": trainer.total_update_steps // max(args.steps + args.warmup, 1),
    }
    if world > 1:  # every rank must have run the same number of optimiser steps (= collectives)
        t = torch.tensor([trainer.total_update_steps], dtype=torch.int64, device=dev)
        allt = [torch.zeros_like(t) for _ in range(world)]
        dist.all_gather(allt, t)
        out["update_steps_per_rank"] = [int(x.item()) for x in allt]
        if len(set(out["update_steps_per_rank"])) != 1:
            raise SystemExit(f"ranks ran different numbers of optimiser steps: {out['update_steps_per_rank']}")
        if "allreduce" in last_metrics:
            out["allreduce"] = last_metrics["allreduce"]
    if trainer.hip_graph_fallback:
        out["hip_graph_fallback"] = trainer.hip_graph_fallback
    if trainer.rollout_graph_fallback:
        out["rollout_graph_fallback"] = trainer.rollout_graph_fallback
    if args.workload == "transformer65536":
        # the update as a whole: forward + backward of the policy over one minibatch = 3 x the forward FLOPs the "cls"
        # reduction needs (same accounting as policy_encoder), against the time a minibatch takes end to end
        full = 17 * 2 * (256 * 768 + 256 * 256 + 2 * 256 * 1024) + 8 * 2 * 2 * 17 * 17 * 32
        last = 17 * 2 * 256 * 512 + 2 * (256 * 256 + 256 * 256 + 2 * 256 * 1024) + 8 * 2 * 2 * 17 * 32
        layers = MODEL_CFG["num_layers"]
        mb = out["update_minibatches_per_step"]
        if mb:
            ms_mb = out["phase_seconds_per_step"]["update_s"] / mb * 1e3
            tf = 3 * ((layers - 1) * full + last) * args.train_batch / (ms_mb * 1e-3) / 1e12
            out["update"] = {"ms_per_minibatch": round(ms_mb, 3), "minibatch": args.train_batch,
                             "algorithmic_TFLOPs": round(tf, 1), "frac_of_bf16_peak": round(tf / 2500.0, 4),
                             "what": f"forward + loss + backward ({'hipGraph replay' if hip_graph else 'EAGER fallback'}) + clip + "
                                     "AdamW per minibatch"}
    us = timed.mean_us()
    if us:
        live_per_launch = env_steps / world / max(len(timed.events), 1)
        gbs = POLICY_STEP_BYTES * live_per_launch / (us * 1e-6) / 1e9
        out["in_loop"] = {"kernel": "k_policy_step (g2048_policy_step)", "launches": len(timed.events),
                          "launch_us": round(us, 2), "boards_per_launch": args.boards,
                          "mean_live_boards_per_launch": round(live_per_launch, 1),
                          "algorithmic_bytes_per_env_step": POLICY_STEP_BYTES, "achieved_GBps": round(gbs, 1),
                          "frac_of_hbm_peak": round(gbs / HBM_PEAK_GBS, 5)}
    del trainer
    torch.cuda.empty_cache()

    # ------------------------------------------------------------------ extras, after the timed region
    if not args.no_extras and args.workload == "transformer65536":
        # (all ranks: these trainers run the same collectives as the headline)
        fr = make_trainer(args.workload, optim=dict(OPTIM_CFG, max_lr=0.0))
        n, s, m = run_iterations(fr, global_boards, 1, 1)
        out["frozen_policy"] = {"value": round(n / s, 1), "unit": "env-steps/sec", "env_steps_per_ppo_iteration": int(n),
                                "hip_graph": bool(m.get("hip_graph", False)),
                                "what": "the headline iteration with max_lr = 0 (random-init policy stays put: same work per "
                                        "iteration whatever --steps is); 1 warm-up + 1 timed iteration"}
        out["value_frozen_policy"] = out["frozen_policy"]["value"]
        # (round 4, same box, same tree, only the summation order of the rollout heads changed: 9.85 M vs 8.43 M env-steps per
        # iteration and value 4.27 vs 3.94 M while value_frozen_policy went UP 0.7 % - NOTES.md, round-4 appendix)
        out["value_note"] = ("value = env-steps per iteration / seconds per iteration of a LEARNING policy: the numerator follows the "
                             "episode length of the 5-to-25-iterations-old policy, i.e. the learning trajectory, which any change of "
                             "the rollout's rounding re-rolls (+-8 % measured); value_frozen_policy and update.ms_per_minibatch "
                             "measure the code")
        del fr
        torch.cuda.empty_cache()
        fx = make_trainer(args.workload, rollout_mode="fixed_horizon", rollout_horizon=args.horizon)
        n, s, m = run_iterations(fx, global_boards, 1, 2)
        out["fixed_horizon"] = {"value": round(n / s, 1), "unit": "env-steps/sec", "horizon": args.horizon,
                                "env_steps_per_ppo_iteration": int(n / 2), "boards_per_launch": args.boards,
                                "mean_live_boards_per_launch": args.boards, "hip_graph": bool(m.get("hip_graph", False)),
                                "what": "throughput mode (per-lane auto-reset, every lane live at every lock-step, GAE "
                                        "bootstrapped at the horizon), same update; 1 warm-up + 2 timed iterations"}
        del fx
        torch.cuda.empty_cache()
    if rank == 0:
        out["roofline"] = step_kernel_roofline(dev, args.roofline_boards)
        if not args.no_extras:
            out["policy_step"] = policy_step_saturated(dev)
        if args.workload == "transformer65536":
            enc = policy_encoder_roofline(agent, dev, args.boards)
            if enc:
                out["policy_encoder"] = enc
        if not args.no_extras:
            r = BatchRunner(init_seed=0, act_fn=act_randomly, rng_mode="partitionable", device=dev)
            r.collect(args.boards)  # warm
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            n = 0
            for _ in range(5):
                n += r.collect(args.boards).num_steps()
            torch.cuda.synchronize()
            out["env_only"] = {"value": round(n / (time.perf_counter() - t1), 1), "unit": "env-steps/sec",
                               "what": f"fused random-policy rollout of {args.boards} boards, complete episodes, "
                                       "trajectory written to HBM, no policy network"}
            if args.workload == "transformer65536":
                # what an unmodified reference CLI run gets: rollout_amp=None + mixed_precision bfloat16 -> the fused bf16 encoder
                dd = PPOTrainer(PPOAgent(**MODEL_CFG), BatchRunner(init_seed=0, rng_mode="partitionable", device=dev),
                                RolloutBuffer(31, 16, 4), OPTIM_CFG, max_steps=500000, device=dev,
                                log_dir=os.path.join("/tmp", f"g2048_bench_logs_{rank}"), **TRAINER_CFG)
                dd.collect_rollouts(args.boards, 1)  # warm
                torch.cuda.synchronize()
                t1 = time.perf_counter()
                dd.collect_rollouts(args.boards, 1)
                torch.cuda.synchronize()
                out["drop_in_default"] = {
                    "value": round(dd.last_rollout_stats["timesteps"] / (time.perf_counter() - t1), 1),
                    "unit": "env-steps/sec (collect phase only)", "boards": args.boards, "rollout_amp": dd.rollout_amp,
                    "what": "PPOTrainer(rollout_amp=None) as the reference's run/train_ppo_agent.py constructs it: mixed_precision "
                            "bfloat16 implies the bf16 rollout (fused encoder); G2048_ROLLOUT_FP32=1 restores the fp32 rollout"}
                del dd
                torch.cuda.empty_cache()
                # the reference's own rollout precision: the fp32 PyTorch forward
                fb = min(8192, args.boards)
                r32 = BatchRunner(init_seed=0, rng_mode="partitionable", device=dev,
                                  act_fn=TorchActionFunction(agent, use_mask=True, device=dev))
                torch.cuda.synchronize()
                t1 = time.perf_counter()
                with torch.no_grad():
                    n32 = r32.collect(fb).num_steps()
                torch.cuda.synchronize()
                out["drop_in_fp32_rollout"] = {
                    "value": round(n32 / (time.perf_counter() - t1), 1), "unit": "env-steps/sec (collect phase only)",
                    "boards": fb, "what": "the reference's fp32 rollout forward in PyTorch (no fused encoder): what "
                                          "run/train_ppo_agent.py gets with G2048_ROLLOUT_FP32=1; bounded sample"}
            if world == 1 and args.workload == "transformer65536":
                ml = make_trainer("mlp4096")
                n, s, m = run_iterations(ml, 4096, 1, 3)
                out["mlp4096"] = {"value": round(n / s, 1), "unit": "env-steps/sec", "boards": 4096,
                                  "hip_graph": bool(m.get("hip_graph", False)),
                                  "rollout_graph": ml.rollout_graph_fallback is None and len(ml._rollout_graphs) > 0,
                                  **({"rollout_graph_fallback": ml.rollout_graph_fallback} if ml.rollout_graph_fallback else {}),
                                  "what": "BASELINE.json configs[1]: 4 096 boards, MLP policy (flattened one-hot -> 512 -> 512 "
                                          "trunk + the reference's heads), full PPO iterations, reference trainer config"}
                del ml
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(args.boards)
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
