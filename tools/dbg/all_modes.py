"""Diagnostic: steady-state env-steps/s of every task mode of the reference (65 536 envs, contact, domain randomisation,
random actions; 600 pre-roll steps) -- catches a kernel variant that is out of line with the others."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import gym_os2r_amd as g
from gym_os2r_amd import abi, rewards
from gym_os2r_amd.sim import HipSim
from gym_os2r_amd.tasks import monopod, monopod_no_norm

n = int(os.environ.get("OS2R_ENVS", "65536"))
SKIP = set(os.environ.get("OS2R_SKIP", "").split(","))
for mode, rew, norm in (("free_hip", "BalancingV1", True), ("fixed_hip", "BalancingV2", True), ("fixed_hip_simple", "BalancingV1", True),
                        ("fixed_hip_torque", "BalancingV3", True), ("fixed", "BalancingV2", True), ("simple", "StraightV1", True),
                        ("free_hip", "BalancingV1", False)):
    if mode in SKIP and norm:
        continue
    cls = monopod.MonopodTask if norm else monopod_no_norm.MonopodTask
    task = cls(1000, task_mode=mode, reward_class=getattr(rewards, rew), reset_positions=["stand"])
    task.create_spaces()
    model = g.get_model(g.config.SettingsConfig().get_config(f"task_modes/{mode}/model"))
    spec = task.kernel_spec(model, reset_mode=abi.RESET_RANDOM, randomize_params=True, max_episode_steps=100_000)
    for dtype in (abi.F64, abi.F32):
        cfg = abi.config_struct(model, spec, num_envs=n, seed=42, dtype=dtype, contact=True)
        sim = HipSim(cfg, device="cuda:0")
        sim.bench_steps(600)
        ms = sim.bench_steps(300) / 300
        print(f"{mode:18s} {'norm' if norm else 'no_norm':8s} {rew:12s} {'f64' if dtype == abi.F64 else 'f32'}: {ms * 1e3:7.1f} us  {n / ms / 1e3:7.1f} M env-steps/s", flush=True)
        sim.close()
