"""SubprocVecEnv-shaped surface over the batched runtime.

Mirrors the methods of ``gym_os2r.common.vec_env.SubprocVecEnv`` (common/vec_env/subproc_vec_env.py:52-262; base class
vec_env.py:32-245) that a trainer calls.  The per-process Pipe protocol is replaced by the batch dimension of the kernel:

* ``step_async`` *sends*: it enqueues the env-step launch (stream-ordered, the host does not wait) -- what
  ``remote.send(('step', action))`` does in the reference (:114-117) -- and ``step_wait`` *receives*: it returns the device
  tensors of that launch (:119-123).  Between the two a trainer can do its own work (logging, the optimiser step of the
  previous batch, ...) while the physics runs.
* ``num_splits > 1``: the batch is cut into contiguous shards, one runtime (one C-ABI handle) and one stream each.  The
  shards advance independently -- a shard waits for its own slowest wave only, and the SIMDs that one shard's early
  waves leave idle take the next launch of another (+5 % with two, +7 % with four shards at 65 536 environments over
  long rollouts, once the shards have drifted out of phase) -- and can be driven one at a time (``step_async(a, split=i)`` / ``step_wait(split=i)``): a policy evaluates
  shard A while the physics of shard B runs (examples/batched_rollout.py --splits).  Random streams are keyed by the global
  environment index, so the shards reproduce the single-handle batch bit for bit.
"""
from __future__ import annotations

import numpy as np


class HipVecEnv:
    def __init__(self, env, *more_envs):
        """``env``: a (wrapped) batched runtime; ``more_envs``: further shards of the same batch, in order of their
        ``env_offset`` (``common.make_mp_envs(..., num_splits=k)`` builds them)."""
        self.envs = [env, *more_envs]
        self.env = env
        self.num_splits = len(self.envs)
        self.split_sizes = [e.num_envs for e in self.envs]
        bounds = np.concatenate([[0], np.cumsum(self.split_sizes)])
        self.split_slices = [slice(int(bounds[i]), int(bounds[i + 1])) for i in range(self.num_splits)]
        self.num_envs = int(bounds[-1])
        self.observation_space = env.observation_space
        self.action_space = env.action_space
        self.waiting = False
        self.closed = False
        self._pending = [None] * self.num_splits     # results of the launch in flight per shard: (obs, rew, done, info, event)
        self._streams = None
        self._per_env = {}

    # -- streams of the shards (a single shard runs on the caller's current stream) ----------------------------------
    def _stream_of(self, i):
        if self.num_splits == 1:
            return None
        if self._streams is None:
            import torch
            # (the device from the runtime's options: touching `.sim` here would create the handle -- and its first reset -- early)
            dev = self.envs[0].unwrapped._opts.get("device")
            dev = torch.device(dev if dev is not None else "cuda:0")
            self._streams = [torch.cuda.Stream(device=dev) for _ in self.envs]
        return self._streams[i]

    def _launch(self, i, actions):
        import torch
        env, st = self.envs[i], self._stream_of(i)
        if st is None:
            out = env.step(actions)
            self._pending[i] = (*out, None, None)
            return
        cur = torch.cuda.current_stream(st.device)
        st.wait_stream(cur)                           # the actions were produced on the caller's stream
        with torch.cuda.stream(st):
            out = env.step(actions)
            ev = torch.cuda.Event()
            ev.record(st)
        # The shard's kernel reads the caller's tensor (or a view of it) on the shard's stream: the caching allocator must
        # not hand its block to the caller's stream again before that launch is done (the caller may drop or reassign
        # `actions` while evaluating the policy of another shard), and a reference is held until the results are collected.
        if isinstance(actions, torch.Tensor) and actions.is_cuda:
            actions.record_stream(st)
        self._pending[i] = (*out, ev, actions)

    def _collect(self, i):
        import torch
        if self._pending[i] is None:
            raise RuntimeError("step_wait() without a step_async() in flight")
        obs, rew, done, info, ev = self._pending[i][:5]
        self._pending[i] = None
        if ev is not None:
            cur = torch.cuda.current_stream(obs.device)
            cur.wait_event(ev)                        # stream-ordered: the host does not block
            for t in (obs, rew, done):
                t.record_stream(cur)                  # allocated on the shard's stream, consumed on the caller's
            for v in dict.values(info):
                if isinstance(v, torch.Tensor):
                    v.record_stream(cur)
        return obs, rew, done, info

    # -- the VecEnv surface -----------------------------------------------------------------------------------------------
    def step_async(self, actions, split=None):
        """Enqueue the env-step of every shard (or of shard ``split`` with its own actions [n_split, 2])."""
        if split is not None:
            self._launch(int(split), actions)
        elif self.num_splits == 1:
            self._launch(0, actions)
        else:
            for i, sl in enumerate(self.split_slices):
                self._launch(i, actions[sl])
        self.waiting = True

    def step_wait(self, split=None):
        """The results of the launch(es) in flight: device tensors, ordered behind the launch on the caller's stream."""
        if split is not None:
            out = self._collect(int(split))
            self.waiting = any(p is not None for p in self._pending)
            return out
        parts = [self._collect(i) for i in range(self.num_splits)]
        self.waiting = False
        if self.num_splits == 1:
            return parts[0]
        import torch
        from ..runtimes.hip_runtime import BatchedInfo
        obs, rew, done = (torch.cat([p[k] for p in parts]) for k in range(3))
        infos = [p[3] for p in parts]
        eager = {k: torch.cat([dict.__getitem__(d, k) for d in infos]) for k in dict.keys(infos[0])
                 if all(dict.__contains__(d, k) and isinstance(dict.__getitem__(d, k), torch.Tensor) for d in infos)}
        lazy = {k: (lambda k=k: torch.cat([d[k] for d in infos])) for k in infos[0].keys() if k not in eager}
        return obs, rew, done, BatchedInfo(eager, lazy=lazy)

    def step(self, actions):
        self.step_async(actions)
        return self.step_wait()

    def reset(self):
        if any(p is not None for p in self._pending):
            raise RuntimeError("reset() with a step_async() in flight: collect it with step_wait() first")
        if self.num_splits == 1:
            return self.env.reset()
        import torch
        # every shard resets on its own stream, behind whatever it was last given, and the caller's stream waits for all
        cur = torch.cuda.current_stream(self._stream_of(0).device)
        out = []
        for i, e in enumerate(self.envs):
            st = self._stream_of(i)
            st.wait_stream(cur)
            with torch.cuda.stream(st):
                o = e.reset()
            cur.wait_stream(st)
            o.record_stream(cur)
            out.append(o)
        return torch.cat(out)

    def seed(self, seed=None):
        """SubprocVecEnv seeds worker r with seed + r (subproc_vec_env.py:160-164); here every random stream is keyed
        by (seed, global env index), so one seed gives every environment its own stream: -> [seed] * num_envs."""
        out = [e.seed(seed) for e in self.envs][0]
        s0 = out[0] if isinstance(out, (list, tuple)) and out else out
        return [s0 for _ in range(self.num_envs)]

    def close(self):
        if not self.closed:
            for e in self.envs:
                e.close()
            self.closed = True

    def get_state_info(self, state, actions):
        """Per-env (reward, done) for given observations and action histories, on the host."""
        state = np.asarray(state, dtype=np.float64)
        if state.ndim == 1:
            return self.env.get_state_info(state, actions)
        return [self.env.get_state_info(s, a) for s, a in zip(state, actions)]

    # The batched runtime(s) stand for all the environments: an attribute (task, spaces, reward class ...) is shared by
    # construction, where SubprocVecEnv keeps one copy per worker process (common/vec_env/subproc_vec_env.py:125-214).
    # `indices` therefore selects how many answers come back.  What cannot be applied to a strict subset of the
    # environments is refused instead of being recorded without effect.
    def get_attr(self, attr_name, indices=None):
        idx = self._indices(indices)
        shared = getattr(self.env, attr_name) if hasattr(self.env, attr_name) else getattr(self.env.unwrapped, attr_name)
        return [shared for _ in idx]

    def set_attr(self, attr_name, value, indices=None):
        """All environments: the attribute of the shared runtime(s) is set.  A strict subset: refused -- the kernels read
        task attributes per handle, not per environment; what IS per environment has its own interface
        (`HipSim.set_params`, `set_state`, `set_episode_info`)."""
        if indices is not None and sorted(set(self._indices(indices))) != list(range(self.num_envs)):
            raise NotImplementedError(f"set_attr({attr_name!r}) for a subset of the environments: attributes are shared by the batch; "
                                      "per-environment quantities go through unwrapped.sim.set_params / set_state")
        for e in self.envs:
            setattr(e.unwrapped, attr_name, value)

    def env_method(self, method_name, *args, indices=None, **kwargs):
        """One call serves a whole shard; a strict subset of the environments is refused (the call would act on all)."""
        idx = self._indices(indices)
        if sorted(set(idx)) != list(range(self.num_envs)):
            raise NotImplementedError(f"env_method({method_name!r}) for a subset of the environments: the method acts on the whole batch "
                                      "(masked resets: unwrapped.reset(mask))")
        results = [getattr(e, method_name)(*args, **kwargs) for e in self.envs]
        return [results[0] for _ in idx] if self.num_splits == 1 else [results[self._split_of(i)] for i in idx]

    def get_images(self):
        raise NotImplementedError("headless stepper: no rendering")

    def render(self, mode="human"):
        return None

    def _split_of(self, i):
        for k, sl in enumerate(self.split_slices):
            if sl.start <= i < sl.stop:
                return k
        raise IndexError(i)

    def _indices(self, indices):
        if indices is None:
            return list(range(self.num_envs))
        if isinstance(indices, (int, np.integer)):
            indices = [int(indices)]
        idx = [int(i) for i in indices]
        if any(i < 0 or i >= self.num_envs for i in idx):
            raise IndexError(f"env index out of range for {self.num_envs} environments")
        return idx

    @property
    def unwrapped(self):
        return self.env.unwrapped
