"""Batched gym.Env-shaped runtime on the HIP stepper.

Drop-in for ``gym_os2r.runtimes.gazebo_runtime.GazeboRuntime`` (runtimes/gazebo_runtime.py:12-123)
for the env-step path: same constructor kwargs (``task_cls, agent_rate, physics_rate,
real_time_factor`` + the task kwargs ``task_mode, reward_class, reset_positions``), same
``step / reset / seed / close / render`` surface, ``.task``, ``.action_space``,
``.observation_space``.  What changes is the batch dimension: one runtime owns ``num_envs``
environments on one GPU and ``step`` takes ``actions[N, 2]`` and returns
``(obs[N, D], reward[N], done[N], infos)`` with the reference's SubprocVecEnv semantics
(common/vec_env/subproc_vec_env.py:15-21): a done environment is reset inside the same kernel
launch, ``obs`` then holds its first observation of the new episode and
``infos['terminal_observation']`` the last one of the old.

Everything stays on the device (torch tensors); there is no CPU fallback.
"""
from __future__ import annotations

from typing import Optional

import numpy as np

from .. import abi, get_model
from ..config import SettingsConfig


class BatchedInfo(dict):
    """Info of one batched step.  Dict of tensors; ``as_list()`` builds the per-env list of dicts
    a SubprocVecEnv consumer expects (use for small N only).  Derived entries (``truncated``,
    ``reset_orientation_id``) are produced on first access: a training loop that never looks at
    them does not pay their kernel launches every step.  Read them before the next ``step()``."""

    def __init__(self, *args, lazy=None, **kwargs):
        super().__init__(*args, **kwargs)
        self._lazy = dict(lazy or {})

    def __missing__(self, key):
        if key in self._lazy:
            self[key] = self._lazy.pop(key)()
            return self[key]
        raise KeyError(key)

    def __contains__(self, key):
        return dict.__contains__(self, key) or key in self._lazy

    def get(self, key, default=None):
        return self[key] if key in self else default

    def keys(self):
        return list(dict.keys(self)) + list(self._lazy)

    def _materialise(self):
        for key in list(self._lazy):
            self[key]
        return self

    # whole-dict views see every entry: they build the lazy ones first
    def items(self):
        return dict.items(self._materialise())

    def values(self):
        return dict.values(self._materialise())

    def __iter__(self):
        return dict.__iter__(self._materialise())

    def __len__(self):
        return dict.__len__(self) + len(self._lazy)

    def __repr__(self):
        return dict.__repr__(self._materialise())

    def as_list(self, pose_names):
        n = int(self["done_flags"].shape[0])
        flags = self["done_flags"].cpu().numpy()
        poses = self["reset_orientation_id"].cpu().numpy()
        term = self["terminal_observation"].cpu().numpy() if "terminal_observation" in self else None
        out = []
        for i in range(n):
            d = {"reset_orientation": pose_names[int(poses[i])]}
            if flags[i] & abi.TRUNCATED_BIT:
                d["TimeLimit.truncated"] = not bool(flags[i] & abi.DONE_BIT)
            if flags[i] and term is not None:
                d["terminal_observation"] = term[i]
            out.append(d)
        return out


class HipRuntime:
    metadata = {"render.modes": ["human"]}

    def __init__(self, task_cls: type, agent_rate: float, physics_rate: float,
                 real_time_factor: float = float(np.finfo(np.float32).max), *,
                 num_envs: int = 1, device=None, seed: int = 0, dtype: str = "f64",
                 contact: bool = True, max_episode_steps: int = 0, env_offset: int = 0,
                 pgs_iters: Optional[int] = None, pgs_normal_iters: Optional[int] = None, pgs_exact: Optional[int] = None,
                 pgs_tol: Optional[float] = None, auto_reset: bool = True, done_reasons: bool = False,
                 physics_engine=None, world: Optional[str] = None, **kwargs):
        steps = physics_rate / agent_rate
        if steps != int(steps):
            import warnings
            warnings.warn(f"Rounding the number of iterations to {int(steps)} from the nominal {steps}")
        self.num_of_steps_per_run = int(steps)           # runtimes/gazebo_runtime.py:46-55
        self._physics_rate, self._agent_rate = float(physics_rate), float(agent_rate)
        self._real_time_factor = real_time_factor        # accepted for signature parity; runs flat out
        self.task = task_cls(agent_rate=agent_rate, **kwargs)
        self.action_space, self.observation_space = self.task.create_spaces()
        self.task.action_space, self.task.observation_space = self.action_space, self.observation_space
        self.num_envs = int(num_envs)
        self._opts = dict(device=device, seed=int(seed), dtype=dtype, contact=bool(contact),
                          max_episode_steps=int(max_episode_steps), env_offset=int(env_offset),
                          # contact solver (abi.config_struct has the defaults: fp64 -- 2 + at most 14 sweeps (12 below five dof) with the exact
                          # finish, 12 solves at most; `pgs_exact=0, pgs_iters=20` is the sweeps-only solver of rounds 1-2;
                          # `pgs_tol` [J] is the stopping tolerance of the sweeps, 1e-24 in fp64 and 1e-13 in fp32)
                          pgs_iters=None if pgs_iters is None else int(pgs_iters), pgs_normal_iters=None if pgs_normal_iters is None else int(pgs_normal_iters),
                          pgs_exact=None if pgs_exact is None else int(pgs_exact),
                          pgs_tol=None if pgs_tol is None else float(pgs_tol), auto_reset=bool(auto_reset),
                          # info['done_reason']: which observation ended each episode (the reference's debug line,
                          # tasks/monopod.py:288-296, as a bitmask per environment; `done_reason_names` decodes it)
                          done_reasons=bool(done_reasons))
        # set by the randomizer wrappers before the first reset (randomizers/*.py)
        self._reset_mode = abi.RESET_FIXED
        self._randomize_params = False
        self._gravity_rollouts = 0
        self._sim = None
        self._bad_pending = []    # step counters of the launches with device actions whose verdict is still out (oldest first)
        self._bad_seen = 0        # running count of clamped actions already reported
        self._bad_flag = None     # pinned word for the rare direct read (reset / close)
        # step() lets the host run at most this many of its launches ahead of the device (2: the verdict on the actions of the
        # step two calls ago is in before the next launch goes out; larger: later verdicts, a deeper queue)
        self.max_inflight = 2
        cfg = getattr(self.task, "cfg", None) or SettingsConfig()
        self.model = dict(get_model(cfg.get_config(f"task_modes/{self.task.task_mode}/model")))
        self.pose_names = list(cfg.get_config("/resets").keys())
        self._buffers = None

    # -- configuration hooks ----------------------------------------------------------------
    def configure_reset(self, reset_mode: int, randomize_params: bool, gravity_rollouts: int = 0):
        if self._sim is not None:
            raise RuntimeError("reset mode must be chosen before the first reset()")
        self._reset_mode, self._randomize_params = int(reset_mode), bool(randomize_params)
        self._gravity_rollouts = int(gravity_rollouts)
        if reset_mode == abi.RESET_FIXED:
            self.model["gravity_z"] = -9.80665
        return self

    def seed(self, seed=None):
        if self._sim is not None:
            raise RuntimeError("seed() must be called before the first reset()")
        self._opts["seed"] = 0 if seed is None else int(seed)
        self.action_space.seed(seed)
        return [seed]

    # -- lifecycle --------------------------------------------------------------------------
    @property
    def sim(self):
        """The device simulator handle (created on first use; raises without GPU / extension)."""
        if self._sim is None:
            from ..sim import HipSim
            o = self._opts
            spec = self.task.kernel_spec(self.model, reset_mode=self._reset_mode,
                                         randomize_params=self._randomize_params,
                                         max_episode_steps=o["max_episode_steps"],
                                         gravity_rollouts=self._gravity_rollouts)
            cfg = abi.config_struct(self.model, spec, num_envs=self.num_envs,
                                    dtype=abi.F64 if o["dtype"] == "f64" else abi.F32,
                                    env_offset=o["env_offset"], seed=o["seed"],
                                    substeps=self.num_of_steps_per_run, dt=1.0 / self._physics_rate,
                                    contact=o["contact"], pgs_iters=o["pgs_iters"],
                                    pgs_normal_iters=o["pgs_normal_iters"], pgs_exact=o["pgs_exact"],
                                    pgs_tol=o["pgs_tol"], auto_reset=o["auto_reset"])
            self._sim = HipSim(cfg, device=o["device"])
            if o["done_reasons"]:
                self._sim.done_reasons(True)
            self._bad_seen, self._bad_pending = 0, []     # a new handle counts from zero
        return self._sim

    def reset(self, mask=None):
        """Reset every environment (or those in ``mask``) and return the observations [N, D]."""
        self._raise_if_bad_actions(drain=True)
        first = self._sim is None
        sim = self.sim
        if first and mask is None:
            # os2r_create already performed the initial reset of every environment
            import torch
            return sim.reset(torch.zeros(self.num_envs, dtype=torch.uint8, device=sim.device))
        return sim.reset(mask)

    def step(self, actions):
        """actions [N, 2] in [-1, 1] -> (obs [N, D], reward [N], done [N] bool, BatchedInfo)."""
        import torch
        sim = self.sim
        # DLPack at the boundary: a device array of another framework (anything with `__dlpack__`, or a DLPack
        # capsule) is taken over without a copy; what comes back are torch tensors, which export `__dlpack__`
        if not isinstance(actions, (torch.Tensor, np.ndarray, list, tuple)) and (hasattr(actions, "__dlpack__") or type(actions).__name__ == "PyCapsule"):
            actions = torch.from_dlpack(actions)
        # The action space is enforced (gazebo_runtime.py:67-68 warns; the task asserts).  Host actions
        # are checked on the host before the upload.  Device actions are checked by the kernel and the
        # verdict is read two calls later (or in reset()/close()): reading it earlier would make the
        # host wait for a kernel that is still running, every step.  Reading it costs a load from pinned
        # host memory that the next launch's first wave writes (include/os2r.h: os2r_get_violation_mirror):
        # no copy, no event and no other kernel between two env-step launches.
        self._raise_if_bad_actions()
        if not isinstance(actions, torch.Tensor) or not actions.is_cuda:
            a_host = np.asarray(actions, dtype=np.float64)
            if ((a_host < -1) | (a_host > 1)).any():
                raise AssertionError("%r invalid: actions must lie in the action space [-1, 1]" % (actions,))
        a = torch.as_tensor(actions, device=sim.device).to(sim.dtype)
        if a.dim() == 1 and self.num_envs == 1:
            a = a.reshape(1, 2)
        device_actions = isinstance(actions, torch.Tensor) and actions.is_cuda
        if device_actions:
            # the kernel clamps out-of-range actions and counts them; the launch AFTER this one mirrors the running count
            # (never cleared) to the host, and it is compared with the count already reported two calls later
            self._bad_pending.append(sim.step_count & 0xFFFFFFFF)
        obs, rew, flags, term, done = sim.step(a, want_terminal=True, want_mask=True)   # `done` = flags != 0, from the same launch
        if getattr(self.task, "host_reward", False):
            # custom reward class (no in-kernel formula): the reference's extension point is kept
            # through a host evaluation on the stepped (pre-reset) observation and the action
            # history, one environment at a time -- correct but slow, meant for small batches
            o_np = term.cpu().numpy()
            h0 = sim.get_action_history(0).cpu().numpy()
            h1 = sim.get_action_history(1).cpu().numpy()
            vals = [float(self.task.calculate_reward(o_np[i], [h0[:, i], h1[:, i]])) for i in range(self.num_envs)]
            rew = torch.as_tensor(vals, dtype=rew.dtype, device=rew.device)
        lazy = {"reset_orientation_id": lambda: sim.episode_info()[2],
                "truncated": lambda: (flags & abi.TRUNCATED_BIT).bool()}
        if sim.reasons is not None:
            # a copy of this step's reasons, made now on the launching stream (2 bytes per environment): the handle's buffer is
            # rewritten by the next step, and a getter evaluated later would report that step's
            why = sim.reasons.clone()
            lazy["done_reason"] = lambda: why                      # bit d: observation d left the reset space
        info = BatchedInfo(done_flags=flags, terminal_observation=term, lazy=lazy)
        return obs, rew, done, info

    def _raise_if_bad_actions(self, drain: bool = False):
        """Look at the verdicts that are due.  The verdict on the launch with step counter k is in the mirror once a later
        launch has started (mirror[1] > k).  A step waits until at most ONE of its earlier launches is unconfirmed -- the one
        of two calls ago has then been checked, and the host runs at most two launches ahead of the device, as with the event
        of rounds 1-4, without an event; reset() / close() (drain) wait for the stream and read the count directly."""
        if self._sim is None or not self._bad_pending:
            return
        import time
        sim = self._sim
        count = None
        if drain:
            import torch
            if self._bad_flag is None:
                self._bad_flag = torch.zeros(1, dtype=torch.int32).pin_memory()
            sim.action_violations_into(self._bad_flag, clear=False)
            torch.cuda.current_stream(sim.device).synchronize()
            count = int(self._bad_flag[0])
            self._bad_pending.clear()
        else:
            m = sim.violation_mirror()
            deadline = None
            while self._bad_pending:
                started = int(m[1])
                if ((started - self._bad_pending[0] - 1) & 0xFFFFFFFF) < 0x80000000:    # a launch behind the oldest one has started
                    count = int(m[0])
                    self._bad_pending.pop(0)
                    continue
                if len(self._bad_pending) < max(1, int(self.max_inflight)):
                    break
                if deadline is None:
                    deadline = time.perf_counter() + 5.0
                elif time.perf_counter() > deadline:     # (launches on a stream that is not running: do not hang on them)
                    return self._raise_if_bad_actions(drain=True)
                time.sleep(0)
        if count is not None and count != self._bad_seen:
            self._bad_seen = count      # one report per burst
            raise AssertionError("invalid: actions of an earlier step() left the action space [-1, 1] "
                                 "(they were clamped, as the backend clamps the torque)")

    def done_reason_names(self, mask: int):
        """The observation names behind a `done_reason` bitmask (what tasks/monopod.py:288-296 logs in the reference)."""
        names = sorted(self.task.observation_index, key=self.task.observation_index.get)
        return [n for d, n in enumerate(names) if (int(mask) >> d) & 1]

    def get_state_info(self, obs, actions):
        """(reward, done) recomputed on the host for one observation (tasks/monopod.py:348-366)."""
        return self.task.get_state_info(np.asarray(obs, dtype=np.float64), actions)

    def render(self, mode: str = "human", **kwargs):
        return None                                      # headless: there is no Gazebo GUI to open

    def close(self):
        try:
            self._raise_if_bad_actions(drain=True)
        finally:
            if self._sim is not None:
                self._sim.close()
                self._sim = None

    @property
    def unwrapped(self):
        return self
