"""round 3 diagnostic: one env-step of 512 random free_hip states on the GPU with several solver settings, dumped for a
CPU-side comparison with the oracle (default solve and converged solve)."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from helpers import make_config
from gym_os2r_amd import abi
from gym_os2r_amd.sim import HipSim
from test_gpu_parity import _random_states
n = 512
out = {}
for name, kw in (("default", {}), ("tight", dict(pgs_iters=300, pgs_exact=100, pgs_tol=0.0)), ("sweeps300", dict(pgs_iters=300, pgs_exact=0, pgs_tol=0.0)),
                 ("cap1", dict(pgs_exact=1)), ("tol0", dict(pgs_tol=0.0))):
    cfg, task, model = make_config("free_hip", "BalancingV2", True, num_envs=n, contact=True, auto_reset=False, dtype=abi.F64, **kw)
    rng = np.random.default_rng(11)
    q, qd = _random_states(model, n, rng)
    act = rng.uniform(-1, 1, (n, 2))
    sim = HipSim(cfg)
    sim.set_state(q, qd)
    sim.step(torch.as_tensor(act))
    q2, qd2 = (t.cpu().numpy() for t in sim.get_state())
    out["q2_" + name], out["qd2_" + name] = q2, qd2
np.savez(os.path.join(ROOT, "gpurun_out", "r3_onestep.npz"), q=q, qd=qd, act=act, **out)
print("dumped")
