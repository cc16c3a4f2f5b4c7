"""Diagnostic: is a lane's result independent of its wave company?  Permute the environments and compare."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch

from helpers import make_config
from gym_os2r_amd import abi
from gym_os2r_amd.sim import HipSim

n = 4096
for contact, substeps, steps in ((False, 1, 1), (False, 10, 1), (True, 1, 1), (True, 10, 1), (True, 10, 3)):
    cfg, task, model = make_config("free_hip", "BalancingV1", True, num_envs=n, contact=contact, auto_reset=False,
                                   dtype=abi.F64, substeps=substeps)
    rng = np.random.default_rng(31)
    nq = model["nq"]
    q = rng.uniform(-1.0, 1.0, (nq, n)); qd = rng.normal(0, 1.0, (nq, n))
    q[1] = rng.uniform(-0.05, 0.25, n); q[3] = rng.uniform(0.2, 1.6, n); q[4] = rng.uniform(-2.8, -0.4, n)
    acts = [rng.uniform(-1, 1, (n, 2)) for _ in range(steps)]

    def run(q_, qd_, acts_):
        sim = HipSim(cfg)
        sim.set_state(q_, qd_)
        for a in acts_:
            sim.step(torch.as_tensor(np.ascontiguousarray(a)))
        out = [t.cpu().numpy() for t in sim.get_state()]
        sim.close()
        return out

    base = run(q, qd, acts)
    perm = rng.permutation(n)
    shuf = run(q[:, perm], qd[:, perm], [a[perm] for a in acts])
    dq = np.abs(base[0][:, perm] - shuf[0]); dv = np.abs(base[1][:, perm] - shuf[1])
    bad = (dq.max(axis=0) > 0) | (dv.max(axis=0) > 0)
    print(f"contact={contact} substeps={substeps} steps={steps}: differing envs {int(bad.sum())}/{n}, max |dq| {dq.max():.2e}, max |dqd| {dv.max():.2e}")
