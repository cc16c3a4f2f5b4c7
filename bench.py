#!/usr/bin/env python3
"""Env-step throughput of the HIP monopod stepper (BASELINE.json metric).

  python bench.py --gpus N --steps K --warmup W
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

One "step" = one env-step launch over this rank's batch: 10 physics iterations with the action
held + observation/reward/done + auto-reset, random actions drawn on device (no H2D on the timed
path).  Default workload = BASELINE configuration C4 (the per-GPU share of C5): 65 536 envs of the
free-boom balancing task (task_mode 'free_hip', BalancingV1) with ground contact and per-env domain
randomisation (link mass, joint damping/friction, ground friction, gravity), randomised resets.
Envs shard embarrassingly over ranks (contiguous ranges, no data-path collective): weak scaling.

The workload is rolled into its stationary regime before anything is timed (--preroll steps, outside
the timed region and independent of --warmup): freshly reset robots hang 1-5 cm above the ground and the
first few hundred env-steps (free fall, first impacts, collapse) are lighter than the steady state the
rollout then stays in, so a short window right after the reset would measure a phase, not the path.

Rank 0 prints ONE JSON line: the contract keys plus
  roofline      HBM view of the step kernel (algorithmic bytes / HIP-event kernel time of this run)
  roofline_valu fp64 vector-ALU view of the same kernel (the resource that actually binds): the flops the
                kernel executed in THIS run's timed window -- its work counted by replaying the window from a
                checkpoint with the counting variant of the kernel (same arithmetic, bit for bit), priced by
                profiles/flop_model.json (per-unit flops fitted to rocprofv3 PMC counts) -- over the kernel time
  cpu_baseline  the CPU oracle (scalar fp64 port) timed on this box's host cores, N=1 only
"""
import argparse
import contextlib
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8 TB/s spec
FP64_VECTOR_PEAK_TF = 78.6     # MI355X fp64 vector peak (half the 157.3 TF fp32 vector rate)
FP32_VECTOR_PEAK_TF = 157.3

WORKLOADS = {
    # name: (task_mode, reward, contact, domain randomisation, description)
    "C2": ("fixed_hip", "BalancingV1", False, False, "4096 envs fixed_hip, ground contact off, random actions"),
    "C3": ("free_hip", "BalancingV1", True, False, "65536 envs free_hip balancing with ground contact"),
    "C4": ("free_hip", "BalancingV1", True, True,
           "65536 envs/GPU free_hip balancing, ground contact, per-env domain randomisation (C5 = 8 x C4)"),
    # not a BASELINE configuration: the reference's default env id (Monopod-balance-v1) at the same batch size
    "V1": ("fixed_hip_simple", "BalancingV1", True, True,
           "65536 envs/GPU Monopod-balance-v1 (fixed_hip_simple, 4 dof), ground contact, domain randomisation"),
}


def build_config(args, rank, world, split=0, splits=1):
    """Config of rank `rank` of `world` (contiguous shard `rank * envs_per_gpu ...`), or of shard `split` of `splits` of it."""
    import gym_os2r_amd as g
    from gym_os2r_amd import abi, rewards
    from gym_os2r_amd.tasks.monopod import MonopodTask
    mode, reward, contact, dr, _ = WORKLOADS[args.workload]
    task = MonopodTask(1000, task_mode=mode, reward_class=getattr(rewards, reward), reset_positions=["stand"])
    task.create_spaces()
    model = g.get_model(g.config.SettingsConfig().get_config(f"task_modes/{mode}/model"))
    if getattr(args, "runtime_model", False):
        # any change to the robot constants leaves the four compiled-in variants: run-time-model kernels
        model = dict(model)
        model["mass"] = [m * (1.0 + 1e-9) for m in model["mass"]]
    spec = task.kernel_spec(model, reset_mode=abi.RESET_RANDOM if dr else abi.RESET_FIXED,
                            randomize_params=dr, max_episode_steps=100_000)
    n = args.envs_per_gpu
    base = rank * n
    if getattr(args, "total_envs", None):
        # rehearsal of uneven shards: the job's environments are cut into `world` contiguous ranges that differ by at most one
        from gym_os2r_amd.distributed import shard_range
        base, n = shard_range(args.total_envs, rank, world)
    if splits > 1:
        from gym_os2r_amd.distributed import shard_range
        off, n = shard_range(args.envs_per_gpu, split, splits)
        base += off
    cfg = abi.config_struct(model, spec, num_envs=n, env_offset=base, seed=args.seed,
                            dtype=abi.F64 if args.dtype == "f64" else abi.F32, contact=contact,
                            pgs_iters=args.pgs_iters, pgs_normal_iters=args.pgs_normal_iters, pgs_tol=args.pgs_tol, pgs_exact=getattr(args, 'pgs_exact', None))
    return cfg, model, spec


def algorithmic_bytes_per_env_step(cfg, esz):
    """HBM bytes one env-step must move (DESIGN.md 'Data layout'): state + per-env parameters in,
    state + observation + terminal observation + reward + done out (what a gym-level env.step asks of the launch);
    actions are generated on device."""
    nq, D = cfg.model.nq, cfg.task.obs_dim
    dr = cfg.task.reset_mode == 1
    reads = (2 * nq + 2) * esz + 4 + 4 + 1            # q, qd, last action; step / episode counters, pose
    if dr:
        reads += (4 * nq + 1) * esz                   # mass scale, damping, friction, mu, gravity
    writes = (2 * nq + 4) * esz + 2 * D * esz + esz + 1 + 4  # q, qd, action history x2; obs and terminal obs; reward; done; steps
    if esz == 8 and cfg.pgs_exact > 0 and cfg.pgs_normal_iters > 0:
        # the contact solver's state (round 4): impulses of the contacts of the bodies that can touch + joint impulses + flags, in and out
        bodies = len({int(cfg.model.cand_body[k]) for k in range(int(cfg.model.ncand))}) if cfg.contact else 0
        reads += (3 * bodies + nq) * esz + 4
        writes += 4 * nq * esz + 4
    return reads + writes


def survey_bytes_per_env_step(cfg, esz):
    """SURVEY.md 8(d)'s own count of the algorithmic bytes: reads q, qd, action, previous action, per-env parameters; writes q, qd,
    obs, reward, done (1 B), current action -- 125 / 149 / 233 B for C2 / C3 / C4 in fp32, twice that (minus the flag byte) in fp64.
    What the launch moves beyond it (terminal observation, counters, the contact solver's state in and out: `algorithmic_bytes_per_env_step`)
    is the builder's addition to the environment's state and outputs, not part of the survey's yardstick."""
    nq, D = cfg.model.nq, cfg.task.obs_dim
    P = 4 * nq + 1 if cfg.task.reset_mode == 1 else 0
    return (2 * nq + 2 + 2 + P) * esz + (2 * nq + D + 1 + 2) * esz + 1


def gym_level(args, cfg_bench, ck, value):
    """The surface users call (replaces gym_os2r/runtimes/gazebo_runtime.py:65-97): env-steps/s through
    `HipRuntime.step(actions)` behind the randomizer wrapper -- Python, ctypes marshalling, fresh output tensors per step, the
    action check, info -- on the bench workload, continued from the timed window's checkpoint (the same stationary regime), with a
    fresh U(-1, 1) device action tensor per step (views of one pre-generated block: what a policy hands over, without a policy's
    kernels), bracketed by synchronize like the headline.  Plus the host's own cost per call, measured with the GPU idle (64
    environments, the queue allowed to run deep so that nothing waits)."""
    import functools
    import torch
    from gym_os2r_amd import rewards
    from gym_os2r_amd.common import make_env_from_id
    from gym_os2r_amd.randomizers.monopod import MonopodEnvRandomizer
    from gym_os2r_amd.randomizers.monopod_no_rand import MonopodEnvNoRandomizer
    mode, reward, contact, dr, _ = WORKLOADS[args.workload]
    env_id = {"free_hip": "Monopod-hop-v1", "fixed_hip": "Monopod-stand-v1", "fixed_hip_simple": "Monopod-balance-v1"}[mode]
    wrapper = MonopodEnvRandomizer if dr else MonopodEnvNoRandomizer

    def make(n):
        env = wrapper(env=functools.partial(make_env_from_id, env_id=env_id, num_envs=n, seed=args.seed, contact=contact,
                                            reward_class=getattr(rewards, reward), reset_positions=["stand"], max_episode_steps=100_000,
                                            dtype=args.dtype, pgs_iters=args.pgs_iters, pgs_normal_iters=args.pgs_normal_iters,
                                            pgs_tol=args.pgs_tol, pgs_exact=getattr(args, "pgs_exact", None)))
        env.reset()
        return env
    n = args.envs_per_gpu
    env = make(n)
    sim = env.unwrapped.sim
    if ck is not None:                                # the stationary regime the headline was timed in
        if not dr:                                    # (restoring parameters would switch a nominal handle to per-env ones)
            ck = {k_: v_ for k_, v_ in ck.items() if k_ != "params"}
            ck["params"] = {}
        sim.restore(ck)
    else:
        sim.bench_steps(max(args.preroll, 1))
    K, warm = 300, 30        # its own window whatever --steps is: over 20 steps the loop's start-up (first launches, allocator) is 10 % of it
    acts = torch.rand(K + warm, n, 2, dtype=sim.dtype, device=sim.device) * 2 - 1
    for k in range(warm):
        env.step(acts[k])
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(warm, warm + K):
        obs, rew, done, info = env.step(acts[k])
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    env.close()
    small = make(64)
    small.unwrapped.max_inflight = 1 << 30            # nothing waits: what is timed is the host
    a64 = torch.rand(400, 64, 2, dtype=sim.dtype, device=sim.device) * 2 - 1
    for k in range(50):
        small.step(a64[k])
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    for k in range(50, 400):
        small.step(a64[k])
    host = (time.perf_counter() - t1) / 350
    torch.cuda.synchronize()
    small.close()
    v = n * K / dt
    return {"value": v, "unit": "env-steps/s", "frac_of_value": v / value, "us_per_step": dt / K * 1e6, "steps": K,
            "host_us_per_call": host * 1e6,
            "surface": "MonopodEnvRandomizer(HipRuntime).step(actions[N, 2] device tensor) -> obs, reward, done, info: fresh output tensors per "
                       "step, done mask and action check from the step launch itself (ABI 5): one kernel launch per call",
            "actions": "a fresh U(-1,1) device tensor per step (pre-generated: no policy kernels)",
            "host_us_per_call_note": "64 environments, GPU idle, run-ahead unbounded: Python + ctypes + torch.empty + the HIP enqueue"}


def rollout_line(args, sim, K=10):
    """The separate open-loop line (never the headline): K env-steps per launch (os2r_rollout), continued from where the handle is."""
    import torch
    n_l = max(1, min(args.steps, 1000) // K)
    obs_k = torch.empty(K, sim.N, sim.D, dtype=sim.dtype, device=sim.device)
    term_k = torch.empty_like(obs_k)
    rew_k = torch.empty(K, sim.N, dtype=sim.dtype, device=sim.device)
    done_k = torch.empty(K, sim.N, dtype=torch.uint8, device=sim.device)
    for _ in range(2):
        sim.rollout_into(K, None, obs_k, rew_k, done_k, term_k)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n_l):
        sim.rollout_into(K, None, obs_k, rew_k, done_k, term_k)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    return {"K": K, "value": sim.N * K * n_l / dt, "unit": "env-steps/s", "us_per_env_step_of_the_batch": dt / (K * n_l) * 1e6, "launches": n_l,
            "note": "open-loop rollouts, K env-steps per launch (os2r_rollout: no device-wide barrier between the env-steps), every "
                    "output of every step written; a separate line, never the headline"}


def host_cores():
    """Cores this process may actually use: the affinity mask, cut down to the cgroup's CPU quota (a GPU box hands a
    container a share of the host -- 16 cores per GPU on this pool -- while os.cpu_count() reports the whole machine;
    64 threads on such a share is what made the round-2 baseline scale 9x on '64 cores')."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        with open("/sys/fs/cgroup/cpu.max") as f:
            quota, period = f.read().split()[:2]
        if quota != "max":
            n = min(n, max(1, int(float(quota) / float(period) + 0.5)))
    except (OSError, ValueError):
        pass
    return max(1, n)


def cpu_baseline(args, cfg):
    """The CPU oracle (same algorithm, scalar fp64 C port) on the host cores: reported baseline."""
    import copy
    from oracle import oracle_py
    oracle_py.build()
    cores = min(host_cores(), 64)
    cfg2 = copy.copy(cfg)
    cfg2.num_envs = args.cpu_envs
    orc = oracle_py.OracleSim(cfg2, threads=cores)
    orc.step(None)                                    # warm
    t0 = time.perf_counter()
    steps = 0
    while True:
        orc.step(None)
        steps += 1
        if time.perf_counter() - t0 >= args.cpu_seconds:
            break
    dt = time.perf_counter() - t0
    orc.close()
    # the same port on one core (SURVEY 8d asks for both), a few seconds
    cfg1 = copy.copy(cfg)
    cfg1.num_envs = 128
    one = oracle_py.OracleSim(cfg1, threads=1)
    one.step(None)
    t1, steps1 = time.perf_counter(), 0
    while time.perf_counter() - t1 < min(3.0, args.cpu_seconds):
        one.step(None)
        steps1 += 1
    dt1 = time.perf_counter() - t1
    one.close()
    return {"value": steps * args.cpu_envs / dt, "unit": "env-steps/s", "cores": cores, "kind": "port",
            "reference_runtime": "unavailable (gym-ignition / ScenarIO / Ignition Gazebo / DART are not installed here; SURVEY 8d item 2)",
            "sample": f"{steps} env-steps x {args.cpu_envs} envs of the same workload, OpenMP over envs ({cores} threads = the cores this container may use; os.cpu_count() = {os.cpu_count()}), {dt:.1f} s",
            "single_core": {"value": steps1 * 128 / dt1, "sample": f"{steps1} env-steps x 128 envs, {dt1:.1f} s"}}


def load_json(name):
    p = os.path.join(ROOT, "profiles", name)
    if os.path.exists(p):
        with open(p) as f:
            return json.load(f)
    return None


def counted_flops(sims, cks, steps, dr, workload, skip=0):
    """Replay the timed window from its checkpoint(s) with the counting kernel variant -- every shard of the batch in turn --:
    the work the timed launches did (the rollout is deterministic), priced with profiles/flop_model.json.  -> dict or None.
    `flops_per_launch` is per env-step of the whole batch (all its shards)."""
    from gym_os2r_amd.sim import Os2rError
    model = (load_json("flop_model.json") or {}).get(f"{workload}_f64")
    c, waves = {}, 0
    for sim, ck in zip(sims, cks):
        try:
            sim.set_state(ck["q"], ck["qd"])
            if "solver_lambda" in ck:
                sim.set_solver_state(ck["solver_lambda"], ck["solver_flags"])
            sim.set_action_history(0, ck["hist0"]); sim.set_action_history(1, ck["hist1"])
            if dr:                                    # (restoring parameters would switch a nominal handle to per-env ones)
                for f, v in ck["params"].items():
                    sim.set_params(f, v)
            sim.set_episode_info(ck["steps"], ck["episode"], ck["pose"])
            sim.step_count = ck["step_count"]
            if skip:
                sim.bench_steps(skip)                 # the untimed steps that lay between the checkpoint and the window
            sim.count_work(True)
            sim.bench_steps(steps)
            ci = sim.work_counters()
            sim.count_work(False)
        except Os2rError:
            sim.count_work(False)
            return None                               # no counting variant for this configuration
        for k_, v_ in ci.items():
            c[k_] = c.get(k_, 0) + v_
        waves += (sim.N + 63) // 64
    wi = max(c["wave_iterations"], 1)
    out = {"activity": {"scanned_bodies_per_wave_iteration": c["scanned_bodies"] / wi,
                        "row_bodies_per_wave_iteration": c["row_bodies"] / wi,
                        "phase2_sweeps_per_wave_iteration": c["sweeps"] / wi,
                        "bodies_in_contact_per_env": c["lane_contacts"] / (wi * 64.0),
                        "live_envs_per_sweep": c["live_lane_sweeps"] / max(c["sweeps"], 1),
                        "full_sincos_per_wave_iteration": c["full_sincos"] / wi,
                        "exact_solves_per_wave_iteration": c["exact_solves"] / wi,
                        "envs_per_exact_solve": c["lane_exact_solves"] / max(c["exact_solves"], 1)},
           "counters": c}
    if model:
        k = model["flops_per_unit"]
        fixed = k["launch_wave"] * waves * steps + k["wave_iteration"] * c["wave_iterations"] + k["scanned_body"] * c["scanned_bodies"]
        rows_, sweeps_, solves_ = k["row_body"] * c["row_bodies"], k["body_sweep"] * c["body_sweeps"] + k["sweep"] * c["sweeps"], k.get("exact_solve", 0.0) * c["exact_solves"]
        out["flops_per_launch"] = (fixed + rows_ + sweeps_ + solves_) / steps
        # The same work with the masked-off lanes taken out: a wave-instruction of a body's rows is useful for the
        # lanes in contact with that body, one of a sweep for the lanes still live in it, one of an exact solve for the
        # lanes that take part; the per-launch and per-iteration parts and the candidate scans run for every lane.
        live_rows = c["lane_contacts"] / max(64.0 * c["row_bodies"], 1.0)
        live_sweeps = c["live_lane_sweeps"] / max(64.0 * c["sweeps"], 1.0)
        live_solves = c["lane_exact_solves"] / max(64.0 * c["exact_solves"], 1.0)
        out["useful_flops_per_launch"] = (fixed + rows_ * live_rows + sweeps_ * live_sweeps + solves_ * live_solves) / steps
        out["model"] = model.get("source")
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=1000)
    ap.add_argument("--warmup", type=int, default=100)
    ap.add_argument("--preroll", type=int, default=1000,
                    help="env-steps run before warm-up, outside the timed region, so that the window measures the "
                         "stationary regime of the rollout whatever --warmup and --steps are (0: start at the reset)")
    ap.add_argument("--workload", default="C4", choices=sorted(WORKLOADS))
    ap.add_argument("--envs-per-gpu", type=int, default=None)
    ap.add_argument("--total-envs", type=int, default=None,
                    help="rehearsals only: the whole job's environments, cut into one contiguous range per rank (ranges differ by at "
                         "most one environment: the gather's uneven-shard path); default: --envs-per-gpu on every rank (weak scaling)")
    ap.add_argument("--dtype", default="f64", choices=["f64", "f32"])
    ap.add_argument("--seed", type=int, default=42)
    ap.add_argument("--pgs-iters", type=int, default=None, help="cap on the phase-2 sweeps (default: abi.config_struct's: 14 with the exact finish -- 12 below five dof --, 20 without)")
    ap.add_argument("--pgs-exact", type=int, default=None, help="exact free-set solves per physics iteration at most (default 12 in f64; 0: sweeps only, the round-2 solver)")
    ap.add_argument("--pgs-normal-iters", type=int, default=None, help="normal sweeps that fix the friction box (default: abi.DEFAULT_PGS_NORMAL_ITERS = 2)")
    ap.add_argument("--pgs-tol", type=float, default=None,
                    help="stopping tolerance of the solver's sweeps [J] (default: 1e-24 for f64, 1e-13 for f32; 0: fixed counts)")
    ap.add_argument("--dump-gathered", default=None,
                    help="with --gather-obs: rank 0 saves the observations / rewards / done flags gathered in the last "
                         "timed step to this .npz (tests compare them with a single handle of all the environments)")
    ap.add_argument("--no-count", action="store_true", help="skip the counting replay of the timed window")
    ap.add_argument("--cpu-envs", type=int, default=2048)
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-gym-level", action="store_true", help="skip the gym-level (HipRuntime.step) and rollout lines")
    ap.add_argument("--runtime-model", action="store_true",
                    help="perturb the robot constants so that the generic (run-time model) kernels are used")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="process-group backend for the barrier / max-time reduction (nccl = RCCL; gloo lets several "
                         "ranks share one GPU when rehearsing the multi-rank path)")
    ap.add_argument("--share-gpu", action="store_true", help="rehearsal: every rank uses cuda:0")
    ap.add_argument("--splits", type=int, default=1,
                    help="cut this rank's batch into that many contiguous shards, one handle and one stream each, advancing "
                         "independently (a shard waits for its own slowest wave only; DESIGN.md 7).  Results per environment "
                         "are those of the single batch, bit for bit.  Pays over long rollouts, once the shards have drifted out of "
                         "phase (+5 % with 4 shards over 1000 steps, a loss over 20); default 1: one handle, one launch per env-step")
    ap.add_argument("--rollout", type=int, default=0, metavar="K",
                    help="a SEPARATE measurement, never the headline: advance in open-loop rollouts of K env-steps per launch "
                         "(os2r_rollout: every wave steps its own environments K times, no device-wide barrier per env-step) instead "
                         "of one launch per env-step; --steps is rounded down to a multiple of K")
    ap.add_argument("--gather-obs", action="store_true",
                    help="also gather obs / reward / done of every step to rank 0 (the optional RCCL collective of "
                         "SURVEY 8e; off the step path, so off by default): one launch + one gather per step")
    args = ap.parse_args()
    if args.envs_per_gpu is None:
        args.envs_per_gpu = 4096 if args.workload == "C2" else 65536

    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus > 1 and world == 1:
        # started without a launcher: become the launcher (child processes; nothing here touched the GPU)
        port = os.environ.get("MASTER_PORT", "29511")
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
               "--master-addr", "127.0.0.1", "--master-port", port, os.path.abspath(__file__)] + sys.argv[1:]
        sys.exit(subprocess.call(cmd))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))

    if args.gather_obs:
        args.splits = 1                               # the gather experiment times one launch + one gather per step
    if args.rollout > 0:
        args.splits, args.gather_obs = 1, False
        args.steps = max(args.rollout, args.steps // args.rollout * args.rollout)
    if args.splits > 1:
        # one hardware queue per shard stream (the HIP runtime maps streams onto GPU_MAX_HW_QUEUES queues, 4 by default,
        # and two shards that share a queue run one after the other: 311 instead of 157 us per step with four shards);
        # read when the runtime initialises, i.e. before anything below touches the GPU
        os.environ.setdefault("GPU_MAX_HW_QUEUES", str(max(8, 2 * args.splits)))
    import torch
    import torch.distributed as dist
    from gym_os2r_amd.sim import HipSim

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the stepper has no CPU fallback")
    if args.share_gpu:
        local_rank = 0
    if torch.cuda.device_count() < local_rank + 1:
        raise SystemExit(f"rank {rank} (LOCAL_RANK {local_rank}) has no GPU: {torch.cuda.device_count()} device(s) visible; "
                         "one process per GPU is expected (or --share-gpu for a rehearsal on one)")
    torch.cuda.set_device(local_rank)
    use_dist = world > 1 or os.environ.get("OS2R_BENCH_FORCE_DIST") == "1"   # the latter: rehearse RCCL init on one GPU
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group("gloo")

    if use_dist and dist.get_world_size() != world:
        raise SystemExit(f"WORLD_SIZE={world} but the process group has {dist.get_world_size()} ranks")
    S = max(1, args.splits)
    if args.total_envs and S > 1:
        raise SystemExit("--total-envs (uneven ranges per rank) is a rehearsal of the gather path: not with --splits")
    cfg, model, spec = build_config(args, rank, world)
    if args.total_envs:
        args.envs_per_gpu = int(cfg.num_envs)         # this rank's range (the ranges differ by at most one environment)
    esz = 8 if args.dtype == "f64" else 4
    if S == 1:
        sims, streams = [HipSim(cfg, device=f"cuda:{local_rank}")], [None]
    else:
        # shards of this rank's batch: a handle and a stream each, advancing independently
        sims = [HipSim(build_config(args, rank, world, i, S)[0], device=f"cuda:{local_rank}") for i in range(S)]
        streams = [torch.cuda.Stream(device=sims[0].device) for _ in range(S)]
    sim = sims[0]

    def on(i):
        return torch.cuda.stream(streams[i]) if streams[i] is not None else contextlib.nullcontext()

    def barrier():
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    for i in range(S):
        with on(i):
            if args.preroll > 0:
                sims[i].bench_enqueue(args.preroll)
            if args.warmup > 0:
                sims[i].bench_enqueue(args.warmup)
    torch.cuda.synchronize()
    count = rank == 0 and not args.no_count and not args.gather_obs and args.dtype == "f64" and args.rollout == 0
    for _ in range(3 if use_dist else 0):
        barrier()                                     # communicator set-up and first-use costs of the barrier itself stay outside
    cks = [s_.checkpoint() for s_ in sims] if count else None   # for the counting replay of the timed window
    # A GPU that idles just before the window -- through the checkpoint's small copies, or while the host sits in the
    # opening barrier (150-400 us under RCCL) -- has dropped its clock, and the first launches of a short timed window run
    # up to 6 % slow (measured: 145.6 against 137.9 us per launch over 20 steps behind an RCCL barrier).  So a few more
    # untimed steps are in flight from here until the synchronize that ends the opening bracket.
    busy_steps = 40
    for i in range(S):
        with on(i):
            # (the first of them through the timed entry point: the first use of the handle's timing events costs the host ~35 us,
            # which would otherwise sit inside a 20-step window -- tools/dbg/window_overhead.py)
            sims[i].bench_steps(1)
            sims[i].bench_enqueue(busy_steps - 1)
    barrier()
    t0 = time.perf_counter()
    if S > 1:
        # every shard's K launches go to its own stream (enqueue only: the host does not wait in between); the
        # shard's launches run back to back there, so its events give its average launch duration
        evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(S)]
        for i in range(S):
            evs[i][0].record(streams[i])
        HipSim.bench_enqueue_shards(sims, streams, args.steps)   # round robin in C: every stream starts within microseconds
        for i in range(S):
            evs[i][1].record(streams[i])
        torch.cuda.synchronize()
        kernel_ms = sum(e0.elapsed_time(e1) for e0, e1 in evs) / S   # mean over the shards of (K launches of that shard)
    elif args.rollout > 0:
        # open-loop rollouts of K env-steps per launch, device-drawn actions, every output of every step written
        K = args.rollout
        obs_k = torch.empty(K, sim.N, sim.D, dtype=sim.dtype, device=sim.device)
        term_k = torch.empty_like(obs_k)
        rew_k = torch.empty(K, sim.N, dtype=sim.dtype, device=sim.device)
        done_k = torch.empty(K, sim.N, dtype=torch.uint8, device=sim.device)
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ev0.record()
        for _ in range(args.steps // K):
            sim.rollout_into(K, None, obs_k, rew_k, done_k, term_k)
        ev1.record()
        torch.cuda.synchronize()
        kernel_ms = ev0.elapsed_time(ev1)
    elif not args.gather_obs:
        kernel_ms = sim.bench_steps(args.steps)       # K launches, HIP events on the launch stream
    else:
        # The optional gather of SURVEY 8e / f-4: observations, rewards and done flags of every step travel to rank 0.
        # Double-buffered and on a side stream: the gather of step k runs behind the launch of step k + 1 (5.9 MB per
        # rank and step at 65 536 envs: ~40 us on one xGMI link against a ~140 us step), the host waits for nothing
        # inside the loop, and a buffer is reused only after the gather that read it has finished (an event per buffer).
        from gym_os2r_amd.distributed import gather_to_rank0
        host = use_dist and args.backend == "gloo"    # gloo gathers host tensors: that rehearsal path stays synchronous
        n, D = sim.N, sim.D
        total_g = args.total_envs or args.envs_per_gpu * world
        bufs = [[torch.empty(n, D, dtype=sim.dtype, device=sim.device), torch.empty(n, dtype=sim.dtype, device=sim.device),
                 torch.empty(n, dtype=torch.uint8, device=sim.device)] for _ in range(2)]
        free = [None, None]                           # event: the gather that read buffer b has finished
        side = torch.cuda.Stream(device=sim.device)
        main = torch.cuda.current_stream(sim.device)
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ev0.record()
        gathered = None
        for k in range(args.steps):
            b = k & 1
            if free[b] is not None:
                main.wait_event(free[b])
            sim.step_into(None, bufs[b][0], bufs[b][1], bufs[b][2])
            stepped = torch.cuda.Event()
            stepped.record(main)
            if host:
                gathered = [gather_to_rank0(t.cpu(), total_g, key=nm) for t, nm in zip(bufs[b], ("obs", "reward", "done"))]
            else:
                side.wait_event(stepped)
                with torch.cuda.stream(side):
                    # (rank 0's [world * n, ...] outputs are allocated once, the collective writes into their slices)
                    gathered = [gather_to_rank0(t, total_g, key=nm) for t, nm in zip(bufs[b], ("obs", "reward", "done"))]
                    free[b] = torch.cuda.Event()
                    free[b].record(side)
        ev1.record()
        torch.cuda.synchronize()
        kernel_ms = ev0.elapsed_time(ev1)             # launches + whatever of the gathers was not hidden behind them
    # closing bracket: this rank's K steps are over when its device has drained; the clock is read there, then the
    # ranks meet at the barrier, and the MAX over ranks of the elapsed times below is when the slowest rank was done.
    # (Reading the clock behind the barrier would add the collective's own latency -- 150-400 us under RCCL, 5-10 % of
    # a 20-step window -- to a path that has no collective.)
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    barrier()
    if args.gather_obs and args.dump_gathered and rank == 0:
        import numpy as np
        np.savez(args.dump_gathered, obs=gathered[0].cpu().numpy(), reward=gathered[1].cpu().numpy(), done=gathered[2].cpu().numpy(),
                 steps_run=args.preroll + args.warmup + busy_steps + args.steps)
    if use_dist:
        t = torch.tensor([elapsed, kernel_ms], dtype=torch.float64, device="cuda" if args.backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed, kernel_ms = float(t[0]), float(t[1])

    total_envs = args.total_envs or args.envs_per_gpu * world
    value = total_envs * args.steps / elapsed
    if rank == 0:
        per_launch_s = kernel_ms * 1e-3 / args.steps     # (per env-step of the batch; a rollout launch makes --rollout of them)
        bytes_launch = algorithmic_bytes_per_env_step(cfg, esz) * args.envs_per_gpu
        achieved = bytes_launch / per_launch_s / 1e9
        traffic = load_json("traffic.json")
        out = {
            "metric": "env-steps/sec (aggregate) monopod balance task" + (f", open-loop rollouts of {args.rollout} env-steps per launch (not the headline)" if args.rollout > 0 else ""),
            "value": value, "unit": "env-steps/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed * 1e3 / args.steps, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
            "config": {"workload": f"{args.workload}: {WORKLOADS[args.workload][4]}",
                       "envs_per_gpu": args.envs_per_gpu, "total_envs": total_envs, "task_mode": WORKLOADS[args.workload][0],
                       "substeps": int(cfg.substeps), "dt": float(cfg.dt), "pgs_sweeps": [int(cfg.pgs_normal_iters), int(cfg.pgs_iters)],
                       "pgs_tol": float(cfg.pgs_tol), "pgs_exact": int(cfg.pgs_exact), "preroll_steps": args.preroll,
                       "contact": bool(cfg.contact), "domain_randomisation": WORKLOADS[args.workload][3],
                       "actions": "U(-1,1) Philox on device",
                       "sharding": f"envs x{world}, " + ("obs/reward/done gathered to rank 0 every step (side stream, double-buffered)" if args.gather_obs else "no step-path collective"),
                       "n_ranks_seen": dist.get_world_size() if use_dist else 1,
                       "splits": S, "rollout_steps_per_launch": args.rollout if args.rollout > 0 else 1},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS,
                         "traffic": None if traffic is None else traffic.get("hbm_bytes_per_launch"),
                         "traffic_source": None if traffic is None else
                         "profiles/traffic.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of " + str(traffic.get("source")) + "; not measured by this run)",
                         "algorithmic_bytes_per_launch": bytes_launch / S,
                         "algorithmic_bytes_per_env_step": algorithmic_bytes_per_env_step(cfg, esz),
                         # SURVEY 8(d)'s own count beside the builder's: the launch also moves the terminal
                         # observation, counters and -- since round 4 -- the contact solver's state in and out
                         "algorithmic_bytes_survey": survey_bytes_per_env_step(cfg, esz),
                         "achieved_survey": survey_bytes_per_env_step(cfg, esz) * args.envs_per_gpu / per_launch_s / 1e9,
                         "frac_survey": survey_bytes_per_env_step(cfg, esz) * args.envs_per_gpu / per_launch_s / 1e9 / HBM_PEAK_GBS,
                         "kernel_ms_per_launch": per_launch_s * 1e3,
                         "concurrent_launches": S,
                         "achieved_per_launch": bytes_launch / S / per_launch_s / 1e9,
                         "note": "ALU-bound path: see roofline_valu" + ("" if S == 1 else f"; {S} shards on {S} streams: a step of the batch is {S} concurrent launches of "
                                                                          "1/S of the bytes each; kernel_ms_per_launch is the mean launch duration of a shard (HIP events on its stream), "
                                                                          "achieved_per_launch = algorithmic_bytes_per_launch / kernel_ms_per_launch, and `achieved` = concurrent_launches x that: the rate of the launches in flight together")},
        }
        peak = FP64_VECTOR_PEAK_TF if args.dtype == "f64" else FP32_VECTOR_PEAK_TF
        cf = counted_flops(sims, cks, args.steps, WORKLOADS[args.workload][3], args.workload, skip=busy_steps) if count else None
        if cf and "flops_per_launch" in cf:
            tf = cf["flops_per_launch"] / per_launch_s / 1e12
            tfu = cf["useful_flops_per_launch"] / per_launch_s / 1e12
            out["roofline_valu"] = {"bound": "valu_" + args.dtype, "achieved": tf, "peak": peak, "unit": "TFLOP/s",
                                    "frac": tf / peak, "issued_lane_flops_per_env_step": cf["flops_per_launch"] / args.envs_per_gpu,
                                    "achieved_useful": tfu, "frac_useful": tfu / peak,
                                    "useful_flops_per_env_step": cf["useful_flops_per_launch"] / args.envs_per_gpu,
                                    "flops_source": "work of this run's timed window (counting replay from a checkpoint) x " + str(cf["model"])
                                                    + "; `achieved` / `frac` price every wave-instruction as 64 lane operations whatever its exec mask "
                                                      "(issue slots), `achieved_useful` / `frac_useful` take the masked-off lanes of rows, sweeps and exact solves out",
                                    "activity": cf["activity"]}
        elif cf:
            out["roofline_valu"] = {"bound": "valu_" + args.dtype, "achieved": None, "peak": peak, "unit": "TFLOP/s", "frac": None,
                                    "flops_source": "profiles/flop_model.json missing: work counted, not priced", "activity": cf["activity"]}
        spec = (load_json("flops.json") or {}).get(f"{args.workload}_{args.dtype}")
        if spec and "roofline_valu" in out:
            # the specification's straightforward operation count (tests/diag/count_flops.py), for reference only
            out["roofline_valu"]["specification_flops_per_env_step"] = spec["flops_per_env_step"]
        if world == 1 and S == 1 and args.rollout == 0 and not args.gather_obs and not args.no_gym_level and not getattr(args, "runtime_model", False):
            # computed after the timed region, like cpu_baseline: the surface users call, and the open-loop rollout line
            # (add-on measurements: a failure here is recorded, it does not take the headline with it)
            try:
                ck_g = cks[0] if cks else sim.checkpoint()
                out["rollout"] = rollout_line(args, sim)
            except Exception as e:                    # noqa: BLE001
                out["rollout"] = {"error": f"{type(e).__name__}: {e}"}
            try:
                out["gym_level"] = gym_level(args, cfg, ck_g, value)
            except Exception as e:                    # noqa: BLE001
                out["gym_level"] = {"error": f"{type(e).__name__}: {e}"}
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(args, cfg)
        print(json.dumps(out), flush=True)
    for s_ in sims:
        s_.close()
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
