#!/usr/bin/env python3
"""Writes tests/golden/spec_trajectories.npz: states the CPU oracle reaches from the reset, for a few configurations.

NOT a reference pin (the reference's physics backend is absent): a guard against *unintended* changes of the
stepper's specification.  The oracle carries experimental switches (orc_set_experimental_*) and every round edits the solver;
this fixture says what the default configuration produced when it was written, so that a change of the default shows up
as a test failure that has to be acknowledged by regenerating the file (python tools/gen_spec_fixture.py) and saying why.
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

CASES = [  # (name, task mode, domain randomisation, env-steps)
    ("free_hip_dr", "free_hip", True, 300),
    ("free_hip_nominal", "free_hip", False, 300),
    ("fixed_hip_simple_dr", "fixed_hip_simple", True, 200),
    ("fixed_nominal", "fixed", False, 200),
]


def run(case):
    from helpers import make_config
    from gym_os2r_amd import abi
    from oracle import oracle_py
    name, mode, dr, steps = case
    cfg, _, _ = make_config(mode, "BalancingV1", True, num_envs=24, contact=True, auto_reset=True, seed=7, max_episode_steps=120,
                            reset_mode=abi.RESET_RANDOM if dr else abi.RESET_FIXED, randomize_params=dr)
    o = oracle_py.OracleSim(cfg, threads=1)
    rews, dones = [], []
    for _ in range(steps):
        _, r, d, _ = o.step(None)
        rews.append(r.copy()); dones.append(d.copy())
    q, qd = o.get_state()
    lam, fl = o.get_solver_state()
    o.close()
    return {f"{name}/q": q, f"{name}/qd": qd, f"{name}/solver_lambda": lam, f"{name}/solver_flags": fl,
            f"{name}/reward_sum": np.sum(rews, axis=0), f"{name}/done_count": np.sum(np.array(dones) != 0, axis=0)}


def toolchain():
    """What built and runs the oracle here: compiler and C library (the fixture records it; on another toolchain 1-ulp differences
    of libm's sin / cos can flip a done flag of a flailing robot, and the drift test allows for that)."""
    import platform
    import subprocess
    try:
        cc = subprocess.run(["gcc", "--version"], capture_output=True, text=True).stdout.splitlines()[0]
    except Exception:
        cc = "gcc ?"
    return f"{cc}; libc {'-'.join(platform.libc_ver())}; {platform.machine()}"


def main():
    from oracle import oracle_py
    oracle_py.build()
    out = {"toolchain": np.array(toolchain())}
    for c in CASES:
        out.update(run(c))
    path = os.path.join(ROOT, "tests", "golden", "spec_trajectories.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, {k: v.shape for k, v in out.items() if k.endswith("/q")})


if __name__ == "__main__":
    main()
