#!/usr/bin/env python3
"""Generate golden vectors under tests/golden/ by RUNNING the reference's own Python.

Build-container only: reads /root/reference at run time, never copies it.  What is run:

 (A) imported unmodified, by file path (they need only numpy / PyYAML):
       gym_os2r/rewards/rewards_utils.py   -> tolerance()
       gym_os2r/utils/reset.py             -> leg_joint_angles()
       gym_os2r/models/config/__init__.py  -> SettingsConfig (+ default/settings.yaml)
 (B) imported unmodified, with empty stand-in modules for third-party packages that are not
     installed in this image (gym, gym_ignition, scenario).  The stand-ins hold no reference
     logic: a Box with low/high/contains, a Task base with an agent_rate, type aliases and a
     silent logger.
       gym_os2r/tasks/monopod.py, gym_os2r/tasks/monopod_no_norm.py -> MonopodTask
       gym_os2r/rewards/__init__.py                                 -> reward classes
     The task is driven through its real methods (create_spaces, set_action,
     get_observation, get_reward, is_done) against a fake `model` that returns the joint
     positions / velocities we choose -- the only thing the physics backend supplies.

Outputs (committed):
  tests/golden/tolerance.npz       grid x 8 sigmoids x parameter sets
  tests/golden/reset_ik.json       leg_joint_angles per reset pose / task mode + random pitches
  tests/golden/task_epilogue.npz   obs / reward / done for 6 task modes x {norm, no_norm}
  tests/golden/task_layout.json    observation_index / mask / periodic joints / limits per mode
"""
import argparse
import importlib
import importlib.util
import json
import os
import sys
import types

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, "tests", "golden")

SIGMOIDS = ["gaussian", "hyperbolic", "long_tail", "reciprocal", "cosine", "linear",
            "quadratic", "tanh_squared"]


def load_by_path(name, path):
    spec = importlib.util.spec_from_file_location(name, path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def install_standins(ref):
    """Empty third-party stand-ins + a `gym_os2r` namespace that does not run the package
    __init__ (which would pull in Gazebo-only modules)."""
    class Box:
        def __init__(self, low, high, dtype=np.float64):
            self.low = np.asarray(low, dtype=dtype)
            self.high = np.asarray(high, dtype=dtype)
            self.dtype = np.dtype(dtype)
            self.shape = self.low.shape

        def contains(self, x):
            x = np.asarray(x)
            return bool(np.can_cast(x.dtype, self.dtype) and x.shape == self.shape
                        and np.all(x >= self.low) and np.all(x <= self.high))

    gym = types.ModuleType("gym")
    gym.spaces = types.ModuleType("gym.spaces")
    gym.spaces.Box = Box
    gi = types.ModuleType("gym_ignition")
    gi.base = types.ModuleType("gym_ignition.base")
    gi.base.task = types.ModuleType("gym_ignition.base.task")

    class Task:
        def __init__(self, agent_rate):
            self.agent_rate = agent_rate
    gi.base.task.Task = Task
    gi.utils = types.ModuleType("gym_ignition.utils")
    gi.utils.typing = types.ModuleType("gym_ignition.utils.typing")
    for n in ("Action", "Reward", "ActionSpace", "ObservationSpace"):
        setattr(gi.utils.typing, n, object)
    gi.utils.typing.Observation = lambda x: x
    gi.utils.logger = types.ModuleType("gym_ignition.utils.logger")
    gi.utils.logger.debug = lambda *a, **k: None
    gi.utils.logger.warn = lambda *a, **k: None
    sc = types.ModuleType("scenario")
    sc.core = types.ModuleType("scenario.core")
    sc.core.JointControlMode_force = 1
    for name, mod in {"gym": gym, "gym.spaces": gym.spaces, "gym_ignition": gi,
                      "gym_ignition.base": gi.base, "gym_ignition.base.task": gi.base.task,
                      "gym_ignition.utils": gi.utils, "gym_ignition.utils.typing": gi.utils.typing,
                      "gym_ignition.utils.logger": gi.utils.logger, "scenario": sc,
                      "scenario.core": sc.core}.items():
        sys.modules[name] = mod
    pkg = types.ModuleType("gym_os2r")
    pkg.__path__ = [os.path.join(ref, "gym_os2r")]
    sys.modules["gym_os2r"] = pkg
    models = types.ModuleType("gym_os2r.models")
    models.__path__ = [os.path.join(ref, "gym_os2r", "models")]
    sys.modules["gym_os2r.models"] = models


class FakeModel:
    """What ScenarIO's Model gives the task: named joint state and force targets."""

    def __init__(self):
        self.pos, self.vel, self.targets = {}, {}, {}

    def joint_positions(self, names):
        return [self.pos[n] for n in names]

    def joint_velocities(self, names):
        return [self.vel[n] for n in names]

    def set_joint_generalized_force_targets(self, data, names):
        for d, n in zip(data, names):
            self.targets[n] = float(d)
        return True

    def joint_generalized_force_targets(self, names):
        return [self.targets[n] for n in names]


def gen_tolerance(ru):
    xs = np.concatenate([np.linspace(-3, 3, 121), [0.0, 0.25, 0.3, 0.11 / 1.57, 0.44 / 1.57,
                                                    1e-9, -1e-9, 1.0, -1.0]])
    params = [  # (lower, upper, margin, value_at_margin)
        (0.0, 0.0, 1.0, 0.4), (0.0, 0.0, 1.0, 0.1), (0.0, 0.0, 0.1, 0.0), (0.0, 0.0, 1.0, 0.0),
        (0.25, 0.3, 0.15, 0.1), (0.11 / 1.57, 0.44 / 1.57, 0.01, 0.1), (0.11, 0.44, 0.0, 0.1),
        (-0.5, 0.5, 2.0, 0.25), (0.0, 0.0, 0.2, 0.1),
    ]
    out = {"x": xs, "params": np.array(params)}
    for si, s in enumerate(SIGMOIDS):
        vals = np.full((len(params), len(xs)), np.nan)
        for pi, (lo, up, mg, vam) in enumerate(params):
            if vam == 0.0 and s not in ("cosine", "linear", "quadratic"):
                continue  # the reference raises ValueError for these; left as NaN
            with np.errstate(all="ignore"):
                vals[pi] = ru.tolerance(xs, bounds=(lo, up), margin=mg, sigmoid=s, value_at_margin=vam)
        out["sigmoid_%d" % si] = vals
    # scalar call path (np.float64 in, python float out) used by the balancing rewards
    sc = np.array([ru.tolerance(np.float64(x), (0.11 / 1.57, 0.44 / 1.57), margin=0.01,
                                sigmoid="long_tail") for x in xs])
    out["scalar_long_tail"] = sc
    return out


def gen_reset_ik(reset_mod, cfg):
    out = {"poses": {}, "random": []}
    resets = cfg.get_config("/resets")
    for mode in ("free_hip", "fixed_hip", "fixed_hip_torque", "fixed_hip_simple", "fixed", "simple",
                 "old-free_hip"):
        definition = cfg.get_config("task_modes/" + mode + "/definition")
        out["poses"][mode] = {"definition": definition, "angles": {}}
        for pname, conf in resets.items():
            if conf["laying_down"]:
                continue
            d = dict(definition)
            d["planarizer_pitch_joint"] = conf["planarizer_pitch_joint"]
            ang = reset_mod.leg_joint_angles(d)
            out["poses"][mode]["angles"][pname] = [conf["planarizer_pitch_joint"], float(ang[0]), float(ang[1])]
    rng = np.random.default_rng(7)
    definition = cfg.get_config("task_modes/free_hip/definition")
    for pitch in rng.uniform(-0.02, 0.25, size=64):
        d = dict(definition)
        d["planarizer_pitch_joint"] = float(pitch)
        ang = reset_mod.leg_joint_angles(d)
        out["random"].append([float(pitch), float(ang[0]), float(ang[1])])
    return out


def epilogue_cases(rng, n_random):
    """(q[5], qd[5], a_prev[2], a[2]) in YAML joint order hip, knee, pitch, yaw, boom_connector."""
    cases = []
    for _ in range(n_random):
        q = np.array([rng.uniform(-6.0, 6.0), rng.uniform(-9.0, 9.0), rng.uniform(-1.5, 1.5),
                      rng.uniform(-9.0, 9.0), rng.uniform(-6.0, 6.0)])
        qd = rng.normal(0, 40.0, size=5)
        cases.append((q, qd, rng.uniform(-1, 1, 2), rng.uniform(-1, 1, 2)))
    # balancing-window and limit edge cases
    base_q = np.array([0.3, -0.6, 0.15, 0.1, 0.05])
    base_qd = np.array([1.0, -2.0, 0.5, 0.3, -0.4])
    pi = np.pi
    for pitch in (0.11, np.nextafter(0.11, 0), 0.110056, 0.44, 0.440224, 0.2, 0.10999, 0.4403,
                  1.5708, -1.5708, np.nextafter(1.5708, 0), 1.57079, 1.6, -1.6):
        q = base_q.copy(); q[2] = pitch
        cases.append((q, base_qd.copy(), np.array([0.2, -0.3]), np.array([0.25, -0.2])))
    for hip in (6.28319, -6.28319, np.nextafter(6.28319, 0), 6.2832, 6.28318, -6.28318):
        q = base_q.copy(); q[0] = hip; q[4] = -hip
        cases.append((q, base_qd.copy(), np.zeros(2), np.array([1.0, -1.0])))
    for ang in (pi, -pi, np.nextafter(pi, 0), np.nextafter(-pi, 0), 3 * pi, -3 * pi, 2 * pi, 0.0,
                pi + 1e-9, -pi - 1e-9, 100.0):
        q = base_q.copy(); q[1] = ang; q[3] = -ang
        cases.append((q, base_qd.copy(), np.array([1.0, 0.5]), np.array([-1.0, 1.0])))
    for v in (369.0, 369.7, 369.74717045495476, 369.75, 370.0, 375.0, 400.0, -369.7, -369.75,
              -400.0, 1e3, 200.0):
        qd = base_qd.copy(); qd[0] = v; qd[3] = -v
        cases.append((base_q.copy(), qd, np.array([-1.0, 0.3]), np.array([0.1, 0.1])))
    for yv in (5.0, 5.5, 6.0, 6.5, 3.0, 9.0, -5.5):   # tanh(0.05*v) around the hopping window
        qd = base_qd.copy(); qd[3] = yv
        cases.append((base_q.copy(), qd, np.array([0.1, 0.12]), np.array([0.15, 0.1])))
    return cases


def gen_task(ref, n_random):
    tasks_norm = importlib.import_module("gym_os2r.tasks.monopod")
    tasks_nonorm = importlib.import_module("gym_os2r.tasks.monopod_no_norm")
    rewards = importlib.import_module("gym_os2r.rewards")
    reward_names = ["BalancingV1", "BalancingV2", "BalancingV3", "StandingV1", "HoppingV1", "StraightV1"]
    modes = ["free_hip", "fixed_hip", "fixed_hip_torque", "fixed_hip_simple", "fixed", "simple"]
    yaml_order = ["hip_joint", "knee_joint", "planarizer_pitch_joint", "planarizer_yaw_joint",
                  "boom_connector_joint"]
    cases = epilogue_cases(np.random.default_rng(1234), n_random)
    arrays = {"q": np.array([c[0] for c in cases]), "qd": np.array([c[1] for c in cases]),
              "a_prev": np.array([c[2] for c in cases]), "a": np.array([c[3] for c in cases])}
    layout = {"joint_order": yaml_order, "reward_names": reward_names, "combos": {}}
    import warnings
    for mode in modes:
        for norm, mod in ((1, tasks_norm), (0, tasks_nonorm)):
            key = f"{mode}__{'norm' if norm else 'nonorm'}"
            first = None
            rew_cols = {}
            for rname in reward_names:
                rcls = getattr(rewards, rname)
                with warnings.catch_warnings():
                    warnings.simplefilter("ignore")
                    task = mod.MonopodTask(agent_rate=1000, task_mode=mode, reward_class=rcls,
                                           reset_positions=["stand"])
                try:
                    task.action_space, task.observation_space = task.create_spaces()
                except AssertionError:
                    continue  # reward does not support this task mode (tasks/monopod.py:192)
                if rname == "HoppingV1" and "planarizer_yaw_joint_vel" not in task.observation_index:
                    continue  # would raise KeyError at reward time in the reference
                task.model = FakeModel()
                obs_l, done_l, rew_l = [], [], []
                for q, qd, ap, a in cases:
                    for i, n in enumerate(yaml_order):
                        task.model.pos[n] = float(q[i]); task.model.vel[n] = float(qd[i])
                    task.set_action(np.asarray(ap, dtype=np.float64), store_action=True)
                    task.set_action(np.asarray(a, dtype=np.float64), store_action=True)
                    obs_l.append(np.array(task.get_observation(), dtype=np.float64))
                    rew_l.append(float(task.get_reward()))
                    done_l.append(bool(task.is_done()))
                rew_cols[rname] = np.array(rew_l)
                if first is None:
                    first = task
                    arrays[key + "__obs"] = np.array(obs_l)
                    arrays[key + "__done"] = np.array(done_l, dtype=np.uint8)
                else:
                    assert np.array_equal(arrays[key + "__obs"], np.array(obs_l), equal_nan=True)
            for rname, col in rew_cols.items():
                arrays[f"{key}__reward__{rname}"] = col
            mask_attr = "observation_mask" if norm else "observaton_mask"
            layout["combos"][key] = {
                "task_mode": mode, "normalized": norm,
                "observation_index": first.observation_index,
                "observation_mask": [int(i) for i in getattr(first, mask_attr)],
                "periodic_joints": [int(i) for i in first.periodic_joints],
                "joint_names": first.joint_names, "action_names": first.action_names,
                "max_torques": [float(x) for x in first.max_torques],
                "observing_measured_torque": bool(first.observing_measured_torque),
                "obs_space_low": [float(x) for x in first.observation_space.low],
                "obs_space_high": [float(x) for x in first.observation_space.high],
                "reset_space_low": [float(x) for x in first.reset_space.low],
                "reset_space_high": [float(x) for x in first.reset_space.high],
                "rewards": sorted(rew_cols),
            }
    return arrays, layout


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reference", default="/root/reference")
    ap.add_argument("--n-random", type=int, default=48)
    args = ap.parse_args()
    ref = args.reference
    os.makedirs(OUT, exist_ok=True)

    ru = load_by_path("ref_rewards_utils", os.path.join(ref, "gym_os2r", "rewards", "rewards_utils.py"))
    rs = load_by_path("ref_reset", os.path.join(ref, "gym_os2r", "utils", "reset.py"))
    cf = load_by_path("ref_config", os.path.join(ref, "gym_os2r", "models", "config", "__init__.py"))
    np.savez_compressed(os.path.join(OUT, "tolerance.npz"), **gen_tolerance(ru))
    with open(os.path.join(OUT, "reset_ik.json"), "w") as f:
        json.dump(gen_reset_ik(rs, cf.SettingsConfig()), f, indent=1)

    install_standins(ref)
    arrays, layout = gen_task(ref, args.n_random)
    np.savez_compressed(os.path.join(OUT, "task_epilogue.npz"), **arrays)
    with open(os.path.join(OUT, "task_layout.json"), "w") as f:
        json.dump(layout, f, indent=1)
    print("cases:", len(arrays["q"]), "combos:", len(layout["combos"]))
    for k, v in layout["combos"].items():
        print(" ", k, "D=%d" % len(v["observation_mask"]), v["rewards"])
    print("numpy", np.__version__, "-> wrote", OUT)


if __name__ == "__main__":
    main()
