"""Thin Python handle over one ``Os2rSim`` of the C-ABI: torch tensors in, torch tensors out.

torch is used only for device memory and the current HIP stream; every call goes straight
to ``libos2r.so``.  All tensors stay on the GPU.
"""
from __future__ import annotations

import ctypes as C
from typing import Optional

import torch

from . import _lib, abi


class Os2rError(RuntimeError):
    pass


class _PybindLib:
    """Adapter giving the pybind11 module (`_os2r_py`) the call shapes of the ctypes library, so the
    rest of this file is binding-agnostic.  Selected with OS2R_BINDING=pybind11."""

    def __init__(self):
        import importlib
        self.m = importlib.import_module("gym_os2r_amd._os2r_py")
        if self.m.abi_version() != abi.ABI_VERSION:
            raise ImportError("_os2r_py ABI version mismatch")

    @staticmethod
    def _a(x):
        if x is None:
            return 0
        v = getattr(x, "value", x)
        return 0 if v is None else int(v)

    def os2r_create(self, cfg_ref, out_ref):
        rc, h = self.m.create(C.addressof(cfg_ref._obj))
        out_ref._obj.value = h
        return rc

    def os2r_destroy(self, h):
        return self.m.destroy(self._a(h))

    def os2r_reset(self, h, mask, obs, st):
        return self.m.reset(self._a(h), self._a(mask), self._a(obs), self._a(st))

    def os2r_step(self, h, act, obs, rew, done, term, st):
        return self.m.step(self._a(h), self._a(act), self._a(obs), self._a(rew), self._a(done), self._a(term), self._a(st))

    def os2r_get_state(self, h, q, qd, st):
        return self.m.get_state(self._a(h), self._a(q), self._a(qd), self._a(st))

    def os2r_set_state(self, h, q, qd, st):
        return self.m.set_state(self._a(h), self._a(q), self._a(qd), self._a(st))

    def os2r_get_solver_state(self, h, lam, flags, st):
        return self.m.get_solver_state(self._a(h), self._a(lam), self._a(flags), self._a(st))

    def os2r_set_solver_state(self, h, lam, flags, st):
        return self.m.set_solver_state(self._a(h), self._a(lam), self._a(flags), self._a(st))

    def os2r_rollout(self, h, n, act, obs, rew, done, term, reason, st):
        return self.m.rollout(self._a(h), int(n), self._a(act), self._a(obs), self._a(rew), self._a(done), self._a(term),
                              self._a(reason), self._a(st))

    def os2r_get_action_history(self, h, w, o, st):
        return self.m.get_action_history(self._a(h), int(w), self._a(o), self._a(st))

    def os2r_set_action_history(self, h, w, i, st):
        return self.m.set_action_history(self._a(h), int(w), self._a(i), self._a(st))

    def os2r_set_params(self, h, f, s, st):
        return self.m.set_params(self._a(h), int(f), self._a(s), self._a(st))

    def os2r_get_params(self, h, f, d, st):
        return self.m.get_params(self._a(h), int(f), self._a(d), self._a(st))

    def os2r_get_episode_info(self, h, s, e, p, st):
        return self.m.get_episode_info(self._a(h), self._a(s), self._a(e), self._a(p), self._a(st))

    def os2r_set_episode_info(self, h, s, e, p, st):
        return self.m.set_episode_info(self._a(h), self._a(s), self._a(e), self._a(p), self._a(st))

    def os2r_get_action_violations(self, h, d, clear, st):
        return self.m.get_action_violations(self._a(h), self._a(d), int(clear), self._a(st))

    def os2r_get_step_count(self, h, out_ref):
        rc, v = self.m.get_step_count(self._a(h))
        out_ref._obj.value = v
        return rc

    def os2r_set_step_count(self, h, v):
        return self.m.set_step_count(self._a(h), int(getattr(v, "value", v)))

    def os2r_bench_steps(self, h, n, st, ms_ref):
        if ms_ref is None:
            return self.m.bench_enqueue(self._a(h), int(n), self._a(st))
        rc, ms = self.m.bench_steps(self._a(h), int(n), self._a(st))
        ms_ref._obj.value = ms
        return rc

    def os2r_set_work_counters(self, h, buf):
        return self.m.set_work_counters(self._a(h), self._a(buf))

    def os2r_set_done_reasons(self, h, buf):
        return self.m.set_done_reasons(self._a(h), self._a(buf))

    def os2r_set_done_mask(self, h, buf):
        return self.m.set_done_mask(self._a(h), self._a(buf))

    def os2r_get_violation_mirror(self, h, out_ref):
        rc, v = self.m.get_violation_mirror(self._a(h))
        out_ref._obj.value = v
        return rc

    def os2r_last_error(self, h):
        return self.m.last_error(self._a(h)).encode()


def _ptr(t: Optional[torch.Tensor]):
    return None if t is None else C.c_void_p(t.data_ptr())


class HipSim:
    """N environments resident on one GPU."""

    def __init__(self, cfg: abi.Os2rConfig, device: Optional[torch.device] = None, binding: Optional[str] = None):
        import os
        binding = binding or os.environ.get("OS2R_BINDING", "ctypes")
        _lib.load()                                   # the C-ABI library must be there either way
        self._lib = _PybindLib() if binding == "pybind11" else _lib.load()
        self.binding = binding
        if not torch.cuda.is_available():
            raise Os2rError("no GPU visible: the stepper runs on MI355X only (no CPU fallback)")
        self.device = torch.device(device if device is not None else f"cuda:{cfg.device}")
        if self.device.index is not None:
            cfg.device = self.device.index
        self.cfg = cfg
        self.N, self.nq, self.D = int(cfg.num_envs), int(cfg.model.nq), int(cfg.task.obs_dim)
        self.dtype = torch.float64 if cfg.dtype == abi.F64 else torch.float32
        # a robot that is not compiled in gets its own kernels (gym_os2r_amd/jit.py; OS2R_JIT=0: generic ones)
        from . import jit
        self.specialised = jit.specialise(_lib.load(), cfg)
        self._counters = None                         # count_work(True) allocates the work counters
        self._mirror = None                           # violation_mirror(): numpy view of the handle's two host words
        self.reasons = None                           # done_reasons(True) allocates the done-reason output
        self._h = C.c_void_p()
        rc = self._lib.os2r_create(C.byref(cfg), C.byref(self._h))
        if rc != abi.OK:
            raise Os2rError(f"os2r_create failed ({rc}): {self._lib.os2r_last_error(None).decode()}")

    # -- plumbing ---------------------------------------------------------------------------
    def _check(self, rc, what):
        if rc != abi.OK:
            raise Os2rError(f"{what} failed ({rc}): {self._lib.os2r_last_error(self._h).decode()}")

    def _stream(self):
        return C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

    def _new(self, *shape, dtype=None):
        return torch.empty(*shape, dtype=dtype or self.dtype, device=self.device)

    def _in(self, t, shape, dtype=None):
        # anything that speaks DLPack (`__dlpack__`: another framework's device array, a capsule) is taken over without a
        # copy when it already lives on this device in the handle's dtype; torch tensors and host arrays as before
        if not isinstance(t, torch.Tensor) and (hasattr(t, "__dlpack__") or type(t).__name__ == "PyCapsule") and not hasattr(t, "__array_interface__"):
            t = torch.from_dlpack(t)
        t = torch.as_tensor(t, device=self.device).to(dtype or self.dtype).contiguous()
        if tuple(t.shape) != tuple(shape):
            raise ValueError(f"expected shape {tuple(shape)}, got {tuple(t.shape)}")
        return t

    def close(self):
        if getattr(self, "_h", None):
            self._lib.os2r_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- hot path ---------------------------------------------------------------------------
    def step(self, actions: Optional[torch.Tensor] = None, want_terminal: bool = True, want_mask: bool = False):
        """-> obs [N,D], reward [N], done [N] uint8 flags, terminal_obs [N,D] or None (and, want_mask: done_mask [N] bool --
        `flags != 0`, written by the same launch: include/os2r.h, os2r_set_done_mask)."""
        a = None if actions is None else self._in(actions, (self.N, 2))
        obs, rew = self._new(self.N, self.D), self._new(self.N)
        done = self._new(self.N, dtype=torch.uint8)
        term = self._new(self.N, self.D) if want_terminal else None
        if want_mask:
            mask = self._new(self.N, dtype=torch.bool)     # one byte per element; the kernel stores 0 / 1
            self._check(self._lib.os2r_set_done_mask(self._h, _ptr(mask)), "os2r_set_done_mask")
        self._check(self._lib.os2r_step(self._h, _ptr(a), _ptr(obs), _ptr(rew), _ptr(done), _ptr(term),
                                        self._stream()), "os2r_step")
        if want_mask:
            self._check(self._lib.os2r_set_done_mask(self._h, None), "os2r_set_done_mask")   # the handle keeps no pointer to a tensor it does not own
            return obs, rew, done, term, mask
        return obs, rew, done, term

    def violation_mirror(self):
        """numpy uint32 [2] view of the handle's two words of pinned host memory (include/os2r.h: os2r_get_violation_mirror):
        [0] the running count of clamped caller actions as the launches before the newest one that has STARTED left it,
        [1] the low 32 bits of that launch's step counter.  Reading it costs a load: no copy, no event, no wait."""
        if self._mirror is None:
            import numpy as np
            p = C.c_void_p()
            self._check(self._lib.os2r_get_violation_mirror(self._h, C.byref(p)), "os2r_get_violation_mirror")
            self._mirror = np.ctypeslib.as_array((C.c_uint32 * 2).from_address(p.value))
        return self._mirror

    def step_into(self, actions, obs, rew, done, term=None):
        """Allocation-free variant writing into caller tensors."""
        self._check(self._lib.os2r_step(self._h, _ptr(actions), _ptr(obs), _ptr(rew), _ptr(done), _ptr(term),
                                        self._stream()), "os2r_step")

    def rollout(self, nsteps: int, actions=None, want_terminal: bool = False, want_reasons: bool = False):
        """`nsteps` env-steps in one call (include/os2r.h: os2r_rollout; one launch where a fused kernel exists): open-loop
        actions [K,N,2] or None (on-device random actions).  -> obs [K,N,D], reward [K,N], done [K,N] uint8 flags,
        terminal_obs [K,N,D] or None, reasons [K,N] int16 or None -- what K calls of step() return, bit for bit."""
        K = int(nsteps)
        a = None if actions is None else self._in(actions, (K, self.N, 2))
        obs, rew = self._new(K, self.N, self.D), self._new(K, self.N)
        done = self._new(K, self.N, dtype=torch.uint8)
        term = self._new(K, self.N, self.D) if want_terminal else None
        why = self._new(K, self.N, dtype=torch.int16) if want_reasons else None
        self._check(self._lib.os2r_rollout(self._h, K, _ptr(a), _ptr(obs), _ptr(rew), _ptr(done), _ptr(term), _ptr(why),
                                           self._stream()), "os2r_rollout")
        return obs, rew, done, term, why

    def _out(self, t, shape, dtype, what):
        """A caller-owned output (or raw input) tensor handed to the library by address: it must be what the kernel assumes."""
        if t is None:
            return
        if not isinstance(t, torch.Tensor) or tuple(t.shape) != tuple(shape) or t.dtype != dtype or t.device != self.device or not t.is_contiguous():
            raise ValueError(f"{what}: expected a contiguous {dtype} tensor of shape {tuple(shape)} on {self.device}, got "
                             f"{getattr(t, 'dtype', type(t))} {tuple(getattr(t, 'shape', ()))} on {getattr(t, 'device', None)}")

    def rollout_into(self, nsteps: int, actions, obs, rew, done, term=None, reasons=None):
        """Allocation-free variant of rollout() writing into caller tensors ([K,N,...]); shapes, dtypes, device and
        contiguity are checked (the library takes addresses: a wrong K or dtype would write out of bounds)."""
        K = int(nsteps)
        if obs is None or rew is None or done is None:
            raise ValueError("rollout_into: obs, rew and done are required")
        self._out(actions, (K, self.N, 2), self.dtype, "actions")
        self._out(obs, (K, self.N, self.D), self.dtype, "obs")
        self._out(rew, (K, self.N), self.dtype, "rew")
        self._out(done, (K, self.N), torch.uint8, "done")
        self._out(term, (K, self.N, self.D), self.dtype, "term")
        self._out(reasons, (K, self.N), torch.int16, "reasons")
        self._check(self._lib.os2r_rollout(self._h, int(nsteps), _ptr(actions), _ptr(obs), _ptr(rew), _ptr(done), _ptr(term),
                                           _ptr(reasons), self._stream()), "os2r_rollout")

    def reset(self, mask: Optional[torch.Tensor] = None):
        m = None if mask is None else self._in(mask, (self.N,), torch.uint8)
        obs = self._new(self.N, self.D)
        self._check(self._lib.os2r_reset(self._h, _ptr(m), _ptr(obs), self._stream()), "os2r_reset")
        return obs

    def bench_steps(self, nsteps: int) -> float:
        """GPU milliseconds (HIP events on the current stream) for nsteps random-action steps."""
        ms = C.c_float()
        self._check(self._lib.os2r_bench_steps(self._h, int(nsteps), self._stream(), C.byref(ms)), "os2r_bench_steps")
        return float(ms.value)

    def done_reasons(self, on: bool = True):
        """Switch the done-reason output on (include/os2r.h: os2r_set_done_reasons) or off.  While on, `self.reasons`
        ([N] int16: bit d = observation slot d was outside the reset space at the end of the last step) is rewritten by
        every step; -> the tensor (the same one every step) or None."""
        new = torch.zeros(self.N, dtype=torch.int16, device=self.device) if on else None
        torch.cuda.current_stream(self.device).synchronize()
        self._check(self._lib.os2r_set_done_reasons(self._h, _ptr(new)), "os2r_set_done_reasons")
        self.reasons = new
        return new

    def bench_enqueue(self, nsteps: int):
        """Enqueue nsteps random-action steps on the current stream and return at once (no events, no wait): for
        callers that drive several handles on several streams -- shards of one batch advancing independently."""
        self._check(self._lib.os2r_bench_steps(self._h, int(nsteps), self._stream(), None), "os2r_bench_steps")

    @staticmethod
    def bench_enqueue_shards(sims, streams, nsteps: int):
        """Enqueue nsteps random-action steps of every shard `sims[i]` on `streams[i]` (torch streams), round robin and
        without waiting (include/os2r.h: os2r_bench_steps_multi): all the streams start together."""
        lib = sims[0]._lib
        if isinstance(lib, _PybindLib):
            rc = lib.m.bench_steps_multi([lib._a(s._h) for s in sims], [int(st.cuda_stream) for st in streams], int(nsteps))
        else:
            hs = (C.c_void_p * len(sims))(*[s._h for s in sims])
            sts = (C.c_void_p * len(sims))(*[C.c_void_p(st.cuda_stream) for st in streams])
            rc = lib.os2r_bench_steps_multi(hs, sts, len(sims), int(nsteps))
        sims[0]._check(rc, "os2r_bench_steps_multi")

    WORK_COUNTERS = ("wave_iterations", "scanned_bodies", "row_bodies", "body_sweeps", "sweeps", "lane_contacts",
                     "live_lane_sweeps", "full_sincos", "exact_solves", "lane_exact_solves")

    def count_work(self, on: bool = True):
        """Switch the counting variant of the step kernel on (include/os2r.h: os2r_set_work_counters) or off.
        While on, `work_counters()` returns what the launches since have added up."""
        new = torch.zeros(len(self.WORK_COUNTERS), dtype=torch.int64, device=self.device) if on else None
        # nothing may still be writing the old buffer, and the handle must stop pointing at it, before it is released
        torch.cuda.current_stream(self.device).synchronize()
        self._check(self._lib.os2r_set_work_counters(self._h, _ptr(new)), "os2r_set_work_counters")
        self._counters = new

    def work_counters(self, clear: bool = True) -> dict:
        if getattr(self, "_counters", None) is None:
            raise Os2rError("work counting is off: call count_work(True) first")
        torch.cuda.current_stream(self.device).synchronize()
        v = self._counters.cpu().tolist()
        if clear:
            self._counters.zero_()
        return dict(zip(self.WORK_COUNTERS, v))

    # -- state ------------------------------------------------------------------------------
    def get_state(self):
        q, qd = self._new(self.nq, self.N), self._new(self.nq, self.N)
        self._check(self._lib.os2r_get_state(self._h, _ptr(q), _ptr(qd), self._stream()), "os2r_get_state")
        return q, qd

    def set_state(self, q=None, qd=None):
        """Set q and / or qd ([nq, N]).  Either way the contact solver's state is cleared (include/os2r.h: a state set from
        outside starts like a reset, also when only one of the two is given): a caller that nudges qd every step gives up the
        warm start of the contact solve -- restore it with set_solver_state() afterwards if the old impulses still apply."""
        q = None if q is None else self._in(q, (self.nq, self.N))
        qd = None if qd is None else self._in(qd, (self.nq, self.N))
        self._check(self._lib.os2r_set_state(self._h, _ptr(q), _ptr(qd), self._stream()), "os2r_set_state")
        torch.cuda.current_stream(self.device).synchronize()  # inputs may be temporaries

    def get_solver_state(self):
        """(lambda [4*nq, N], flags [N] int32 holding the uint32 payload): the impulses that ended every environment's last
        physics iteration and which of them are remembered (include/os2r.h: os2r_get_solver_state)."""
        lam, flags = self._new(4 * self.nq, self.N), self._new(self.N, dtype=torch.int32)
        self._check(self._lib.os2r_get_solver_state(self._h, _ptr(lam), _ptr(flags), self._stream()), "os2r_get_solver_state")
        return lam, flags

    def set_solver_state(self, lam, flags):
        lam = self._in(lam, (4 * self.nq, self.N))
        if not isinstance(flags, torch.Tensor):
            import numpy as np
            flags = torch.from_numpy(np.ascontiguousarray(flags).astype(np.uint32).view(np.int32))
        flags = self._in(flags, (self.N,), torch.int32)
        self._check(self._lib.os2r_set_solver_state(self._h, _ptr(lam), _ptr(flags), self._stream()), "os2r_set_solver_state")
        torch.cuda.current_stream(self.device).synchronize()  # inputs may be temporaries

    def get_action_history(self, which: int):
        out = self._new(2, self.N)
        self._check(self._lib.os2r_get_action_history(self._h, int(which), _ptr(out), self._stream()),
                    "os2r_get_action_history")
        return out

    def set_action_history(self, which: int, value):
        v = self._in(value, (2, self.N))
        self._check(self._lib.os2r_set_action_history(self._h, int(which), _ptr(v), self._stream()),
                    "os2r_set_action_history")
        torch.cuda.current_stream(self.device).synchronize()

    def get_params(self, field: int):
        out = self._new(1 if field == abi.PARAM_GRAVITY else self.nq, self.N)
        self._check(self._lib.os2r_get_params(self._h, int(field), _ptr(out), self._stream()), "os2r_get_params")
        return out

    def set_params(self, field: int, value):
        v = self._in(value, (1 if field == abi.PARAM_GRAVITY else self.nq, self.N))
        self._check(self._lib.os2r_set_params(self._h, int(field), _ptr(v), self._stream()), "os2r_set_params")
        torch.cuda.current_stream(self.device).synchronize()

    def episode_info(self):
        steps = self._new(self.N, dtype=torch.int32)
        epi = self._new(self.N, dtype=torch.int32)   # uint32 payload
        pose = self._new(self.N, dtype=torch.uint8)
        self._check(self._lib.os2r_get_episode_info(self._h, _ptr(steps), _ptr(epi), _ptr(pose), self._stream()),
                    "os2r_get_episode_info")
        return steps, epi, pose

    def set_episode_info(self, steps=None, episode=None, pose=None):
        s = None if steps is None else self._in(steps, (self.N,), torch.int32)
        e = None if episode is None else self._in(episode, (self.N,), torch.int32)
        p = None if pose is None else self._in(pose, (self.N,), torch.uint8)
        self._check(self._lib.os2r_set_episode_info(self._h, _ptr(s), _ptr(e), _ptr(p), self._stream()), "os2r_set_episode_info")
        torch.cuda.current_stream(self.device).synchronize()  # inputs may be temporaries

    # -- checkpoint / resume ----------------------------------------------------------------
    def checkpoint(self) -> dict:
        """Everything that determines the future of this handle, as device tensors (+ the step counter)."""
        q, qd = self.get_state()
        steps, episode, pose = self.episode_info()
        lam, flags = self.get_solver_state()
        return {"q": q, "qd": qd, "solver_lambda": lam, "solver_flags": flags, "hist0": self.get_action_history(0), "hist1": self.get_action_history(1),
                "params": {f: self.get_params(f) for f in (abi.PARAM_MASS_SCALE, abi.PARAM_DAMPING, abi.PARAM_FRICTION,
                                                            abi.PARAM_MU, abi.PARAM_GRAVITY)},
                "steps": steps, "episode": episode, "pose": pose, "step_count": self.step_count}

    def restore(self, ck: dict):
        """Continue from a `checkpoint()` (of this or of another handle with the same configuration)."""
        self.set_state(ck["q"], ck["qd"])
        if "solver_lambda" in ck and "solver_flags" in ck:     # (after set_state, which clears it; a checkpoint written before
            self.set_solver_state(ck["solver_lambda"], ck["solver_flags"])   # ABI 4 has none: the solver starts cold, as after a reset)
        self.set_action_history(0, ck["hist0"]); self.set_action_history(1, ck["hist1"])
        for f, v in ck["params"].items():
            self.set_params(f, v)
        self.set_episode_info(ck["steps"], ck["episode"], ck["pose"])
        self.step_count = ck["step_count"]

    def action_violations_into(self, dst: torch.Tensor, clear: bool = True):
        """Copy the running count of out-of-range caller actions into ``dst`` (one int32/uint32 element,
        device or pinned host memory) on the current stream; nothing waits."""
        self._check(self._lib.os2r_get_action_violations(self._h, C.c_void_p(dst.data_ptr()), 1 if clear else 0,
                                                        self._stream()), "os2r_get_action_violations")

    @property
    def step_count(self) -> int:
        v = C.c_uint64()
        self._check(self._lib.os2r_get_step_count(self._h, C.byref(v)), "os2r_get_step_count")
        return int(v.value)

    @step_count.setter
    def step_count(self, value: int):
        self._check(self._lib.os2r_set_step_count(self._h, C.c_uint64(int(value))), "os2r_set_step_count")
