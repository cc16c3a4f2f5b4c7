"""Shared builders for the tests: task/model -> config struct."""
import warnings

import gym_os2r_amd as g
from gym_os2r_amd import abi, rewards
from gym_os2r_amd.tasks import monopod, monopod_no_norm

MODES = ["free_hip", "fixed_hip", "fixed_hip_torque", "fixed_hip_simple", "fixed", "simple"]


def make_task(mode, reward_name="BalancingV1", normalized=True, reset_positions=("stand",)):
    cls = monopod.MonopodTask if normalized else monopod_no_norm.MonopodTask
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        t = cls(agent_rate=1000, task_mode=mode, reward_class=getattr(rewards, reward_name),
                reset_positions=list(reset_positions))
    t.create_spaces()
    return t


def model_for(mode):
    cfg = g.config.SettingsConfig()
    return g.get_model(cfg.get_config(f"task_modes/{mode}/model"))


def make_config(mode="fixed_hip", reward_name="BalancingV1", normalized=True,
                reset_positions=("stand",), reset_mode=abi.RESET_FIXED, randomize_params=False,
                max_episode_steps=0, model_overrides=None, **cfg_kw):
    task = make_task(mode, reward_name, normalized, reset_positions)
    model = dict(model_for(mode))
    if model_overrides:
        model.update(model_overrides)
    spec = task.kernel_spec(model, reset_mode=reset_mode, randomize_params=randomize_params,
                            max_episode_steps=max_episode_steps)
    return abi.config_struct(model, spec, **cfg_kw), task, model


def perturbed_model(mode, rng):
    """The reference's chain with every inertial / frame number nudged: no compiled-in table matches."""
    m = {k: (list(v) if isinstance(v, list) else v) for k, v in model_for(mode).items()}
    nq = m["nq"]
    m["mass"] = [x * rng.uniform(0.9, 1.1) for x in m["mass"]]
    m["com"] = [[c + rng.uniform(-2e-3, 2e-3) for c in row] for row in m["com"]]
    m["rpos"] = [[c + rng.uniform(-1e-3, 1e-3) for c in row] for row in m["rpos"]]
    m["damping"] = [0.01 * rng.uniform(0.5, 1.5) for _ in range(nq)]
    m["friction"] = [0.004 * rng.uniform(0.5, 1.5) for _ in range(nq)]
    m["mu"] = [rng.uniform(0.3, 1.0) for _ in range(nq)]
    return m
