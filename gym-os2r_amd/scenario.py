"""N = 1 facade with the ScenarIO object surface the reference's tasks and randomizers call.

The subset implemented is exactly what gym-os2r touches: ``GazeboSimulator(step_size,
rtf, steps_per_run).run(paused) / initialize / initialized / get_world / close / step_size``
(runtimes/gazebo_runtime.py:107-121), ``World.to_gazebo().set_gravity / insert_model /
remove_model / get_model / model_names`` (randomizers/monopod.py:60,115,325; models/monopod.py:27),
``Model.joint_positions / joint_velocities / set_joint_generalized_force_targets /
joint_generalized_force_targets / set_joint_control_mode / get_joint / to_gazebo().reset_joint_*``
(tasks/monopod.py:225-249,309-316; randomizers/monopod.py:125-128) and
``Joint.set_joint_max_generalized_force``.  Methods return ``bool`` like ScenarIO's do.

One environment lives on the GPU; ``run()`` advances it by ``steps_per_run`` physics iterations
through the same step kernel as the batched runtime (force targets are consumed by a run, which
is why the reference re-sends them before every iteration, runtimes/gazebo_runtime.py:70-73).
This is plumbing for task-shaped Python code and tests, not a fast path.
"""
from __future__ import annotations

import os
from typing import List, Optional, Sequence

import numpy as np

from . import abi, get_model, load_models

PhysicsEngine_dart = 1
JointControlMode_force = 1
JointControlMode_idle = 0


class Pose:
    def __init__(self, position=(0.0, 0.0, 0.0), orientation=(1.0, 0.0, 0.0, 0.0)):
        self.position, self.orientation = tuple(position), tuple(orientation)


class Joint:
    def __init__(self, model: "Model", name: str):
        self._model, self._name = model, name

    def name(self) -> str:
        return self._name

    def set_joint_max_generalized_force(self, values: Sequence[float]) -> bool:
        self._model._max_force[self._name] = float(values[0])
        return True

    def set_max_generalized_force(self, value: float) -> bool:
        return self.set_joint_max_generalized_force([value])


class Model:
    """One compiled chain + its single-environment simulator handle."""

    def __init__(self, world: "World", compiled: dict, name: str):
        self._world, self._compiled, self._name = world, dict(compiled), name
        self._targets = {n: 0.0 for n in compiled["dof_names"]}
        self._max_force = {}
        self._sim = None

    # -- identification ---------------------------------------------------------------------
    def name(self) -> str:
        return self._name

    def joint_names(self) -> List[str]:
        return list(self._compiled["dof_names"])

    def to_gazebo(self) -> "Model":
        return self

    def get_joint(self, name: str) -> Joint:
        if name not in self._targets:
            raise RuntimeError(f"joint {name!r} not found in model {self._name!r}")
        return Joint(self, name)

    def set_joint_control_mode(self, mode, joint_names: Optional[Sequence[str]] = None) -> bool:
        return all(n in self._targets for n in (joint_names or []))

    # -- simulator --------------------------------------------------------------------------
    def _spec(self) -> dict:
        names, nq = self._compiled["dof_names"], self._compiled["nq"]
        dof = {n: i for i, n in enumerate(names)}
        inf = float("inf")
        has_pitch = "planarizer_pitch_joint" in dof
        return {
            "obs_dim": 2 * nq, "obs_kind": [abi.OBS_POS_RAW] * nq + [abi.OBS_VEL_RAW] * nq,
            "obs_src": list(range(nq)) * 2, "obs_low": [-inf] * (2 * nq), "obs_high": [inf] * (2 * nq),
            "done_lo": [-inf] * (2 * nq), "done_hi": [inf] * (2 * nq),
            "reward_id": abi.REWARD_BALANCING_V1 if has_pitch else abi.REWARD_STRAIGHT_V1, "normalized": 0,
            "idx_pitch_pos": dof.get("planarizer_pitch_joint", -1), "idx_yaw_vel": -1,
            "idx_hip_pos": dof.get("hip_joint", -1), "idx_knee_pos": dof.get("knee_joint", -1),
            "max_episode_steps": 0, "reset_mode": abi.RESET_FIXED, "reset_pose_id": [0], "reset_laying": [0],
            "reset_pitch": [0.0], "reset_hip": [0.0], "reset_knee": [0.0], "reset_simple": 0,
            "leg_def": [200, 190, 80, 2100, 0, 25],
            "dof_yaw": dof.get("planarizer_yaw_joint", -1), "dof_pitch": dof.get("planarizer_pitch_joint", -1),
            "dof_bc": dof.get("boom_connector_joint", -1), "dof_hip": dof.get("hip_joint", -1),
            "dof_knee": dof.get("knee_joint", -1), "randomize_params": 0,
            "dr_mass_lo": 1, "dr_mass_hi": 1, "dr_friction_lo": 0, "dr_friction_hi": 0, "dr_damping_lo": 1,
            "dr_damping_hi": 1, "dr_mu_base": 1, "dr_mu_lo": 1, "dr_mu_hi": 1, "dr_gravity_mean": -9.8,
            "dr_gravity_std": 0.0}

    @property
    def sim(self):
        if self._sim is None:
            from .sim import HipSim
            gz = self._world._sim
            m = dict(self._compiled)
            m["gravity_z"] = self._world._gravity[2]
            cfg = abi.config_struct(m, self._spec(), num_envs=1, dtype=abi.F64, substeps=gz._steps_per_run,
                                    dt=gz._step_size, contact=True, auto_reset=False)
            self._sim = HipSim(cfg, device=gz._device)
        return self._sim

    def _order(self, joint_names):
        names = self._compiled["dof_names"]
        return [names.index(n) for n in (joint_names if joint_names else names)]

    def joint_positions(self, joint_names: Optional[Sequence[str]] = None) -> List[float]:
        q, _ = self.sim.get_state()
        q = q.cpu().numpy()[:, 0]
        return [float(q[i]) for i in self._order(joint_names)]

    def joint_velocities(self, joint_names: Optional[Sequence[str]] = None) -> List[float]:
        _, qd = self.sim.get_state()
        qd = qd.cpu().numpy()[:, 0]
        return [float(qd[i]) for i in self._order(joint_names)]

    def reset_joint_positions(self, values, joint_names=None) -> bool:
        q, _ = self.sim.get_state()
        q = q.cpu().numpy()
        for v, i in zip(values, self._order(joint_names)):
            q[i, 0] = float(v)
        self.sim.set_state(q, None)
        return True

    def reset_joint_velocities(self, values, joint_names=None) -> bool:
        _, qd = self.sim.get_state()
        qd = qd.cpu().numpy()
        for v, i in zip(values, self._order(joint_names)):
            qd[i, 0] = float(v)
        self.sim.set_state(None, qd)
        return True

    def set_joint_generalized_force_targets(self, forces, joint_names=None) -> bool:
        names = list(joint_names) if joint_names else self._compiled["dof_names"]
        if len(forces) != len(names) or any(n not in self._targets for n in names):
            return False
        for f, n in zip(forces, names):
            lim = self._max_force.get(n)
            f = float(f)
            if lim is not None:
                f = min(max(f, -lim), lim)
            self._targets[n] = f
        return True

    def joint_generalized_force_targets(self, joint_names=None) -> List[float]:
        names = list(joint_names) if joint_names else self._compiled["dof_names"]
        return [self._targets[n] for n in names]

    def _advance(self) -> bool:
        import torch
        act_names = [self._compiled["dof_names"][i] for i in self._compiled["act_dof"]]
        for n, f in self._targets.items():
            if n not in act_names and f != 0.0:
                raise RuntimeError(f"force target on unactuated joint {n!r}: only {act_names} are driven")
        mt = self._compiled["max_torque"]
        a = [self._targets[act_names[k]] / mt[k] for k in range(2)]
        if any(abs(x) > 1.0 for x in a):
            return False
        self.sim.step(torch.tensor([a], dtype=torch.float64), want_terminal=False)
        for n in self._targets:
            self._targets[n] = 0.0                    # commands are consumed by the physics update
        return True


class World:
    def __init__(self, sim: "GazeboSimulator"):
        self._sim = sim
        self._models = {}
        self._gravity = (0.0, 0.0, -9.8)

    def name(self) -> str:
        return "default"

    def to_gazebo(self) -> "World":
        return self

    def set_physics_engine(self, engine) -> bool:
        return engine == PhysicsEngine_dart

    def set_gravity(self, gravity) -> bool:
        if any(m._sim is not None for m in self._models.values()):
            return False                               # like Gazebo: physics are fixed once running
        self._gravity = tuple(float(g) for g in gravity)
        return True

    def gravity(self):
        return self._gravity

    def insert_model(self, model_file: str, pose: Optional[Pose] = None, name: Optional[str] = None) -> bool:
        compiled = None
        if model_file in load_models():
            compiled = get_model(model_file)
        elif os.path.exists(model_file) and model_file.endswith(".urdf"):
            from .model_compiler import compile_urdf
            compiled = compile_urdf(model_file)
        if compiled is None:
            return False
        name = name or compiled["name"]
        if name in self._models:
            return False
        self._models[name] = Model(self, compiled, name)
        return True

    def remove_model(self, name: str) -> bool:
        m = self._models.pop(name, None)
        if m is None:
            return False
        if m._sim is not None:
            m._sim.close()
        return True

    def get_model(self, name: str) -> Model:
        if name not in self._models:
            raise RuntimeError(f"model {name!r} not found in the world")
        return self._models[name]

    def model_names(self) -> List[str]:
        return list(self._models)


class GazeboSimulator:
    def __init__(self, step_size: float = 0.001, rtf: float = 1.0, steps_per_run: int = 1, device=None):
        self._step_size, self._rtf, self._steps_per_run = float(step_size), rtf, int(steps_per_run)
        self._device = device
        self._world = None

    def initialize(self) -> bool:
        if self._world is None:
            self._world = World(self)
        return True

    def initialized(self) -> bool:
        return self._world is not None

    def insert_world_from_sdf(self, *_args, **_kwargs) -> bool:
        return True

    def step_size(self) -> float:
        return self._step_size

    def steps_per_run(self) -> int:
        return self._steps_per_run

    def get_world(self, name: str = "") -> World:
        if self._world is None:
            raise RuntimeError("simulator not initialized")
        return self._world

    def world_names(self) -> List[str]:
        return ["default"] if self._world else []

    def gui(self) -> bool:
        return False                                   # headless

    def run(self, paused: bool = False) -> bool:
        if self._world is None:
            return False
        if paused:
            return True                                # processes insertions/removals only
        return all(m._advance() for m in self._world._models.values())

    def close(self) -> bool:
        if self._world is not None:
            for n in list(self._world._models):
                self._world.remove_model(n)
        return True
