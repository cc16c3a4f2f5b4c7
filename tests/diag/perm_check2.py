"""Diagnostic: for an env whose result depends on its wave company, which company is the odd one?"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch

from helpers import make_config
from gym_os2r_amd import abi
from gym_os2r_amd.sim import HipSim
from oracle import oracle_py as o

n = 4096
cfg, task, model = make_config("free_hip", "BalancingV1", True, num_envs=n, contact=True, auto_reset=False, dtype=abi.F64, substeps=1)
rng = np.random.default_rng(31)
nq = model["nq"]
q = rng.uniform(-1.0, 1.0, (nq, n)); qd = rng.normal(0, 1.0, (nq, n))
q[1] = rng.uniform(-0.05, 0.25, n); q[3] = rng.uniform(0.2, 1.6, n); q[4] = rng.uniform(-2.8, -0.4, n)
act = rng.uniform(-1, 1, (n, 2))


def run(q_, qd_, a_):
    c2, _, _ = make_config("free_hip", "BalancingV1", True, num_envs=q_.shape[1], contact=True, auto_reset=False, dtype=abi.F64, substeps=1)
    sim = HipSim(c2)
    sim.set_state(q_, qd_)
    sim.step(torch.as_tensor(np.ascontiguousarray(a_)))
    out = [t.cpu().numpy() for t in sim.get_state()]
    sim.close()
    return out


base = run(q, qd, act)
perm = rng.permutation(n)
shuf = run(q[:, perm], qd[:, perm], act[perm])
inv = np.argsort(perm)
dv = np.abs(base[1] - shuf[1][:, inv]).max(axis=0)
bad = np.where(dv > 0)[0]
print("differing envs:", len(bad))
ms = cfg.model
for e in bad[:6]:
    alone = run(np.repeat(q[:, e:e + 1], 64, axis=1), np.repeat(qd[:, e:e + 1], 64, axis=1), np.repeat(act[e:e + 1], 64, axis=0))
    _, _, rw, ow = o.dynamics(ms, q[:, e], qd[:, e], np.zeros(nq))
    a, _, gap = o.contact_points(ms, rw, ow, cfg.contact_margin)
    w_base = e // 64; w_shuf = inv[e] // 64
    def wave_set(qq, w):
        s = np.zeros(nq, bool)
        for l in range(64):
            _, _, rw_, ow_ = o.dynamics(ms, qq[:, w * 64 + l], np.zeros(nq), np.zeros(nq))
            s |= o.contact_points(ms, rw_, ow_, cfg.contact_margin)[0][:nq]
        return s.astype(int)
    print(f"env {e}: own contacts {a[:nq].astype(int)}; wave sets base {wave_set(q, w_base)} shuf {wave_set(q[:, perm], w_shuf)}; "
          f"alone==base {np.array_equal(alone[1][:, 0], base[1][:, e])} alone==shuf {np.array_equal(alone[1][:, 0], shuf[1][:, inv[e]])} "
          f"|base-shuf| {dv[e]:.1e}")
