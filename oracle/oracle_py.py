"""ctypes access to the CPU oracle (oracle/libos2r_oracle.so).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg.  Nothing under gym-os2r_amd/ may import this module.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def build(force: bool = False, lab: bool = False) -> str:
    """Build the checker (libos2r_oracle.so) or, lab=True, the laboratory (liboracle_lab.so: the same source with
    -DORC_EXPERIMENTS, i.e. with the experimental switches of the solver studies; tests/diag and docs/studies only)."""
    so = os.path.join(_HERE, "liboracle_lab.so" if lab else "libos2r_oracle.so")
    deps = [os.path.join(_HERE, "os2r_oracle.c"), os.path.join(_HERE, "os2r_oracle.h"), os.path.join(_HERE, "..", "include", "os2r.h")]
    if force or not os.path.exists(so) or any(os.path.exists(d) and os.path.getmtime(so) < os.path.getmtime(d) for d in deps):
        subprocess.check_call(["make", "-C", _HERE, "-s"] + (["-B"] if force else []) + (["lab"] if lab else []))
    return so


def _bind(so):
    L = C.CDLL(so)
    L.orc_tolerance.restype = C.c_double
    L.orc_tolerance.argtypes = [C.c_double, C.c_double, C.c_double, C.c_double, C.c_int, C.c_double]
    L.orc_wrap.restype = C.c_double
    L.orc_wrap.argtypes = [C.c_double]
    L.orc_reward.restype = C.c_double
    L.orc_get_step_count.restype = C.c_uint64
    L.orc_set_step_count.argtypes = [C.c_void_p, C.c_uint64]
    return L


def lib():
    """The library the module's functions call: the checker, unless inside `with laboratory():`."""
    global _LIB
    if _LIB is None:
        _LIB = _bind(build())
    return _LIB


_LAB = None


class laboratory:
    """`with oracle_py.laboratory() as L:` -- inside, the module's functions and every OracleSim created use the
    laboratory build (its experimental switches: L.orc_set_experimental_*); an OracleSim keeps the library it was created
    with.  Studies and diagnostics only: the parity tests, smoke() and the CPU baseline use the checker, which has no switch."""

    def __enter__(self):
        global _LIB, _LAB
        if _LAB is None:
            _LAB = _bind(build(lab=True))
            _LAB.orc_debug_counter.restype = C.c_longlong
        self._saved, _LIB = _LIB, _LAB
        return _LAB

    def __exit__(self, *exc):
        global _LIB
        _LIB = self._saved
        return False


def use_laboratory():
    """Switch this process to the laboratory build for good (tests/diag scripts); -> the library."""
    return laboratory().__enter__()


def _p(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


def philox(ctr, key):
    c = (C.c_uint32 * 4)(*ctr)
    k = (C.c_uint32 * 2)(*key)
    o = (C.c_uint32 * 4)()
    lib().orc_philox4x32_10(c, k, o)
    return [int(x) for x in o]


def uniform2(seed, env, stream, ctr, blk):
    u = (C.c_double * 2)()
    lib().orc_uniform2(C.c_uint64(seed), C.c_uint32(env), C.c_uint32(stream), C.c_uint32(ctr),
                       C.c_uint32(blk), u)
    return float(u[0]), float(u[1])


def tolerance(x, lower, upper, margin, sigmoid, value_at_margin):
    return lib().orc_tolerance(x, lower, upper, margin, int(sigmoid), value_at_margin)


def leg_joint_angles(def6, pitch):
    d = (C.c_double * 6)(*[float(v) for v in def6])
    o = (C.c_double * 2)()
    lib().orc_leg_joint_angles(d, C.c_double(pitch), o)
    return float(o[0]), float(o[1])


def observe(task_struct, q, qd, hist1):
    q = np.ascontiguousarray(q, dtype=np.float64)
    qd = np.ascontiguousarray(qd, dtype=np.float64)
    h = np.ascontiguousarray(hist1, dtype=np.float64)
    obs = np.zeros(task_struct.obs_dim)
    lib().orc_observe(C.byref(task_struct), _p(q), _p(qd), _p(h), _p(obs))
    return obs


def done(task_struct, obs):
    obs = np.ascontiguousarray(obs, dtype=np.float64)
    return bool(lib().orc_done(C.byref(task_struct), _p(obs)))


def reward(task_struct, obs, a0, a1):
    obs = np.ascontiguousarray(obs, dtype=np.float64)
    a0 = np.ascontiguousarray(a0, dtype=np.float64)
    a1 = np.ascontiguousarray(a1, dtype=np.float64)
    return float(lib().orc_reward(C.byref(task_struct), _p(obs), _p(a0), _p(a1)))


def dynamics(model_struct, q, qd, tau_full, dt=1e-4, mass_scale=None, damping=None, gravity_z=None):
    """-> qdd[nq], minv[nq,nq], rw[nq,3,3], ow[nq,3] for one environment."""
    n = model_struct.nq
    q = np.ascontiguousarray(q, dtype=np.float64)
    qd = np.ascontiguousarray(qd, dtype=np.float64)
    tau = np.ascontiguousarray(tau_full, dtype=np.float64)
    ms = None if mass_scale is None else np.ascontiguousarray(mass_scale, dtype=np.float64)
    dm = None if damping is None else np.ascontiguousarray(damping, dtype=np.float64)
    g = model_struct.gravity_z if gravity_z is None else gravity_z
    qdd, minv = np.zeros(n), np.zeros((n, n))
    rw, ow = np.zeros((n, 9)), np.zeros((n, 3))
    lib().orc_dynamics(C.byref(model_struct), C.c_double(dt), _p(ms), _p(dm), C.c_double(g),
                       _p(q), _p(qd), _p(tau), _p(qdd), _p(minv), _p(rw), _p(ow))
    return qdd, minv, rw.reshape(n, 3, 3), ow


def contact_points(model_struct, rw, ow, margin=0.0):
    n = model_struct.nq
    rw = np.ascontiguousarray(rw, dtype=np.float64).reshape(n, 9)
    ow = np.ascontiguousarray(ow, dtype=np.float64)
    active = np.zeros(n, dtype=np.int32)
    pw, depth = np.zeros((n, 3)), np.zeros(n)
    lib().orc_contact_points(C.byref(model_struct), C.c_double(margin), _p(rw), _p(ow), _p(active), _p(pw), _p(depth))
    return active.astype(bool), pw, -depth


CONTACT_CENTROID, CONTACT_PER_VERTEX = 0, 1   # os2r_oracle.h: the specification / the oracle-only comparison model


def substep(cfg, q, qd, tau2, mass_scale=None, damping=None, friction=None, mu=None, gravity_z=None,
            contact_model=CONTACT_CENTROID):
    q = np.array(q, dtype=np.float64)
    qd = np.array(qd, dtype=np.float64)
    t = np.ascontiguousarray(tau2, dtype=np.float64)
    arrs = [None if a is None else np.ascontiguousarray(a, dtype=np.float64)
            for a in (mass_scale, damping, friction, mu)]
    g = cfg.model.gravity_z if gravity_z is None else gravity_z
    lib().orc_substep_model(C.byref(cfg), C.c_int(contact_model), _p(arrs[0]), _p(arrs[1]), _p(arrs[2]), _p(arrs[3]),
                            C.c_double(g), _p(q), _p(qd), _p(t))
    return q, qd


def contact_problem(cfg, q, qd, tau2, mass_scale=None, damping=None, friction=None, mu=None, gravity_z=None,
                    contact_model=CONTACT_CENTROID, max_rows=3 * 192 + 5):
    """The boxed LCP of one physics iteration at (q, qd) and the oracle's solution of it with cfg's sweep
    counts (os2r_oracle.c: orc_contact_problem).  -> dict of numpy arrays."""
    n = cfg.model.nq
    q = np.ascontiguousarray(q, dtype=np.float64)
    qd = np.ascontiguousarray(qd, dtype=np.float64)
    t = np.ascontiguousarray(tau2, dtype=np.float64)
    arrs = [None if a is None else np.ascontiguousarray(a, dtype=np.float64)
            for a in (mass_scale, damping, friction, mu)]
    g = cfg.model.gravity_z if gravity_z is None else gravity_z
    vstar, minv, v_out = np.zeros(n), np.zeros((n, n)), np.zeros(n)
    J, target = np.zeros((max_rows, n)), np.zeros(max_rows)
    kind, nrow, body = (np.zeros(max_rows, dtype=np.int32) for _ in range(3))
    bound, box, lam, point = np.zeros(max_rows), np.zeros(max_rows), np.zeros(max_rows), np.zeros((max_rows, 3))
    f = lib().orc_contact_problem
    f.restype = C.c_int
    nr = f(C.byref(cfg), C.c_int(contact_model), _p(arrs[0]), _p(arrs[1]), _p(arrs[2]), _p(arrs[3]), C.c_double(g),
           _p(q), _p(qd), _p(t), C.c_int(max_rows), _p(vstar), _p(minv), _p(J), _p(target), _p(kind), _p(nrow),
           _p(body), _p(bound), _p(box), _p(lam), _p(point), _p(v_out))
    if nr < 0:
        raise RuntimeError("orc_contact_problem: more rows than max_rows")
    return {"nr": nr, "vstar": vstar, "minv": minv, "J": J[:nr], "target": target[:nr], "kind": kind[:nr],
            "normal_row": nrow[:nr], "body": body[:nr], "bound": bound[:nr], "box": box[:nr], "lambda": lam[:nr],
            "point": point[:nr], "v": v_out}


class OracleSim:
    """Batched oracle with the semantics of the C-ABI (host numpy arrays, SoA state)."""

    def __init__(self, cfg, threads: int = 1):
        self.cfg = cfg
        self.N = int(cfg.num_envs)
        self.nq = int(cfg.model.nq)
        self.D = int(cfg.task.obs_dim)
        self._L = lib()                                  # (the checker, or the laboratory inside `with laboratory():`)
        self._h = C.c_void_p()
        rc = self._L.orc_create(C.byref(cfg), C.byref(self._h))
        if rc != 0:
            raise RuntimeError(f"orc_create failed: {rc}")
        self._L.orc_set_threads(self._h, int(threads))

    def set_contact_model(self, model: int):
        """CONTACT_CENTROID (the specification, default) or CONTACT_PER_VERTEX (comparison model)."""
        self._L.orc_set_contact_model(self._h, int(model))

    def solver_counts(self):
        """Diagnostics: (sweeps, solves), int8 [substeps, N] -- the phase-2 sweeps and exact solves of every physics
        iteration of the last step.  The first call switches the recording on and returns None."""
        n = int(self.cfg.substeps)
        sw = np.zeros((n, self.N), dtype=np.int8)
        so = np.zeros((n, self.N), dtype=np.int8)
        if self._L.orc_get_solver_counts(self._h, _p(sw), _p(so)):
            return None
        return sw, so

    def small_solve_counts(self):
        """int8 [substeps, N]: how many of the exact solves of solver_counts() were dual solves of a small free set."""
        sm = np.zeros((int(self.cfg.substeps), self.N), dtype=np.int8)
        if self._L.orc_get_small_solve_counts(self._h, _p(sm)):
            return None
        return sm

    def close(self):
        if self._h:
            self._L.orc_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def reset(self, mask=None):
        obs = np.zeros((self.N, self.D))
        m = None if mask is None else np.ascontiguousarray(mask, dtype=np.uint8)
        self._L.orc_reset(self._h, _p(m), _p(obs))
        return obs

    def step(self, actions=None):
        a = None if actions is None else np.ascontiguousarray(actions, dtype=np.float64)
        obs, term = np.zeros((self.N, self.D)), np.zeros((self.N, self.D))
        rew, done_ = np.zeros(self.N), np.zeros(self.N, dtype=np.uint8)
        self._L.orc_step(self._h, _p(a), _p(obs), _p(rew), _p(done_), _p(term))
        return obs, rew, done_, term

    def get_state(self):
        q, qd = np.zeros((self.nq, self.N)), np.zeros((self.nq, self.N))
        self._L.orc_get_state(self._h, _p(q), _p(qd))
        return q, qd

    def set_state(self, q, qd):
        q = np.ascontiguousarray(q, dtype=np.float64)
        qd = np.ascontiguousarray(qd, dtype=np.float64)
        assert q.shape == (self.nq, self.N) and qd.shape == (self.nq, self.N)
        self._L.orc_set_state(self._h, _p(q), _p(qd))

    def get_solver_state(self):
        """(lam [4*nq, N], flags uint32 [N]): the impulses that ended every environment's last physics iteration, in the
        layout of os2r_get_solver_state (include/os2r.h)."""
        lam, flags = np.zeros((4 * self.nq, self.N)), np.zeros(self.N, dtype=np.uint32)
        self._L.orc_get_solver_state(self._h, _p(lam), _p(flags))
        return lam, flags

    def set_solver_state(self, lam, flags):
        lam = np.ascontiguousarray(lam, dtype=np.float64)
        flags = np.ascontiguousarray(flags, dtype=np.uint32)
        assert lam.shape == (4 * self.nq, self.N) and flags.shape == (self.N,)
        self._L.orc_set_solver_state(self._h, _p(lam), _p(flags))

    def get_action_history(self, which):
        out = np.zeros((2, self.N))
        self._L.orc_get_action_history(self._h, int(which), _p(out))
        return out

    def set_action_history(self, which, arr):
        arr = np.ascontiguousarray(arr, dtype=np.float64)
        assert arr.shape == (2, self.N)
        self._L.orc_set_action_history(self._h, int(which), _p(arr))

    def get_params(self, field):
        count = 1 if field == 4 else self.nq
        out = np.zeros((count, self.N))
        self._L.orc_get_params(self._h, int(field), _p(out))
        return out

    def set_params(self, field, arr):
        count = 1 if field == 4 else self.nq
        arr = np.ascontiguousarray(arr, dtype=np.float64).reshape(count, self.N)
        self._L.orc_set_params(self._h, int(field), _p(arr))

    def episode_info(self):
        steps = np.zeros(self.N, dtype=np.int32)
        epi = np.zeros(self.N, dtype=np.uint32)
        pose = np.zeros(self.N, dtype=np.uint8)
        self._L.orc_get_episode_info(self._h, _p(steps), _p(epi), _p(pose))
        return steps, epi, pose

    def set_episode_info(self, steps=None, episode=None, pose=None):
        a = [None if steps is None else np.ascontiguousarray(steps, dtype=np.int32),
             None if episode is None else np.ascontiguousarray(episode, dtype=np.uint32),
             None if pose is None else np.ascontiguousarray(pose, dtype=np.uint8)]
        assert all(x is None or x.shape == (self.N,) for x in a)
        self._L.orc_set_episode_info(self._h, _p(a[0]), _p(a[1]), _p(a[2]))

    def set_step_count(self, v):
        self.step_count = v

    @property
    def step_count(self):
        return int(self._L.orc_get_step_count(self._h))

    @step_count.setter
    def step_count(self, v):
        self._L.orc_set_step_count(self._h, C.c_uint64(int(v)))
