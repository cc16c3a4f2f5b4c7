#!/usr/bin/env python3
"""Round 5: what a change of the exact finish does to the rounds of a launch's worst lanes and to the accuracy (oracle laboratory).

  python tests/diag/r5_rounds.py [--workload C4|C3|V1] [--envs 16384] [--steps 10] [--closed-loop] VARIANT [VARIANT ...]

A VARIANT is `name` or `name:switch=value,switch=value` with the laboratory's switches (oracle/os2r_oracle.h, ORC_EXPERIMENTS), e.g.
    spec   later8:prox_later=8   small2:small=2   both:small=2,prox_later=8
Per variant: the (environment, iteration) histogram of solves, the solves of a wave (64 consecutive environments: a wave runs as many
rounds in an iteration as its slowest lane) per env-step -- mean / p99 / max --, the modelled phase-2 ticks of a wave and env-step
(4200 a regularised solve, 1400 a dual solve of a small set -- when the variant has them --, 1050 a sweep), and with --closed-loop
the deviation after 1000 balancing env-steps (64 environments, DR) from the converged solve (every cap lifted), the yardstick of
tests/test_oracle_contact.py::test_closed_loop_solver_error_after_1000_balancing_steps."""
import argparse
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import bench                          # noqa: E402
from oracle import oracle_py as O     # noqa: E402

SWITCHES = {"small": ("orc_set_experimental_small", int, 0), "prox": ("orc_set_experimental_prox", int, 2),
            "prox_later": ("orc_set_experimental_prox_later", int, 0), "incons": ("orc_set_experimental_incons", float, 1e-4),
            "stall": ("orc_set_experimental_stall", float, 0.0), "clamp_all": ("orc_set_experimental_clamp_all", int, 0),
            "incons_once": ("orc_set_experimental_incons_once", int, 0), "pivot": ("orc_set_experimental_pivot", int, 0), "equil": ("orc_set_experimental_equil", int, 1), "repin": ("orc_set_experimental_repin", int, 0), "multicut": ("orc_set_experimental_multicut", int, 0), "snap": ("orc_set_experimental_snap", int, 0), "solve_first": ("orc_set_experimental_solve_first", int, 0), "warm_p0": ("orc_set_experimental_warm_p0", int, 0), "margin": ("orc_set_experimental_margin", float, 0.0)}


def apply(L, settings):
    import ctypes
    for k, (fn, typ, default) in SWITCHES.items():
        if not hasattr(L, fn):
            continue
        v = settings.get(k, default)
        getattr(L, fn)(ctypes.c_double(v) if typ is float else int(v))
    L.orc_set_experimental_warm(1, int(settings.get("first", 3)))   # `first=k`: k warm-started sweeps before the first check


def rounds(L, a, settings):
    class A:
        workload = a.workload; envs_per_gpu = a.envs; dtype = "f64"; seed = 42
        pgs_iters = None; pgs_normal_iters = None; pgs_tol = None; pgs_exact = None; runtime_model = False
    cfg, model, spec = bench.build_config(A, 0, 1)
    p1 = int(settings.get("p1", 2))    # (a configuration value: `p1=k` normal sweeps in phase 1; the preroll runs with the default, 2)
    if "tol" in settings:               # (a configuration value, not a switch: `tol=1e-20`)
        cfg.pgs_tol = float(settings["tol"])
    apply(L, {})                       # the preroll is the specification's for every variant: the same states
    o = O.OracleSim(cfg, threads=os.cpu_count() or 1)
    for _ in range(a.preroll):
        o.step(None)
    if p1 != 2:                        # the same states in a handle with another number of normal sweeps
        import copy
        cfg2 = copy.copy(cfg); cfg2.pgs_normal_iters = p1
        o2 = O.OracleSim(cfg2, threads=os.cpu_count() or 1)
        o2.set_state(*o.get_state()); o2.set_solver_state(*o.get_solver_state())
        for w in (0, 1):
            o2.set_action_history(w, o.get_action_history(w))
        for f in range(5):
            o2.set_params(f, o.get_params(f))
        o2.set_episode_info(*o.episode_info()); o2.step_count = o.step_count
        o.close(); o = o2
    apply(L, settings)
    o.solver_counts()
    has_small = hasattr(o, "small_solve_counts") and settings.get("small", 0)
    W = cfg.num_envs // 64
    hist = np.zeros(16, dtype=np.int64)
    T, M = [], []
    for _ in range(a.steps):
        o.step(None)
        sw, so = o.solver_counts()
        sm = o.small_solve_counts() if has_small else np.zeros_like(so)
        so_, sw_, sm_ = (x.reshape(x.shape[0], W, 64).astype(int) for x in (so, sw, sm))
        hist += np.bincount(so.ravel().astype(int), minlength=16)[:16]
        T.append(so_.max(axis=2).sum(axis=0))
        # a wave's rounds of an iteration: as many solves as its slowest lane; a round is priced as a dual solve when every lane
        # that takes part in it solves in the dual -- approximated by: the lane with the most solves decides the kind of each
        big = (so_ - sm_).max(axis=2)             # regularised solves of the wave's worst lane of that kind
        allr = so_.max(axis=2)
        M.append((1050.0 * sw_.max(axis=2) + 4200.0 * big + 1400.0 * np.maximum(allr - big, 0)).sum(axis=0) + 680.0 * p1 * so_.shape[0])
    launch_T, launch_M = np.mean([t.max() for t in T]), np.mean([m.max() for m in M])
    T, M = np.concatenate(T), np.concatenate(M)
    o.close()
    top = int(np.max(np.nonzero(hist))) if hist.any() else 0
    print(f"  (env, iteration) pairs by solves: {hist[:top + 1]}   P(>=1) {hist[1:].sum() / hist.sum():.4f}  P(>=2) {hist[2:].sum() / hist.sum():.5f}  P(>=6) {hist[6:].sum() / hist.sum():.6f}")
    print(f"  solves per wave and env-step: mean {T.mean():.2f}  p99 {np.percentile(T, 99):.0f}  max {T.max()};   modelled phase-2 ticks: mean {M.mean() / 1e3:.1f} k  p99 {np.percentile(M, 99) / 1e3:.1f} k  max {M.max() / 1e3:.1f} k")
    print(f"  the slowest wave of a launch, mean over the launches: {launch_T:.1f} solves, {launch_M / 1e3:.1f} k modelled ticks")


def closed_loop(L, settings):
    from helpers import make_config
    from gym_os2r_amd import abi
    from test_oracle_contact import _pd_policy
    n, steps = 64, 1000

    def run(sw, **kw):
        apply(L, sw)
        cfg, task, model = make_config("free_hip", "BalancingV2", True, num_envs=n, contact=True, auto_reset=False, seed=42,
                                       reset_mode=abi.RESET_RANDOM, randomize_params=True, **kw)
        o = O.OracleSim(cfg, threads=os.cpu_count() or 1)
        ih, ik = model["act_dof"]
        q0, _ = o.get_state()
        rng = np.random.default_rng(2)
        for _ in range(steps):
            q, qd = o.get_state()
            o.step(_pd_policy(q, qd, q0, ih, ik, 0.1 * rng.uniform(-1, 1, (n, 2))))
        x = np.concatenate(o.get_state())
        o.close()
        return x
    global _REF
    rule = (int(settings.get("warm_p0", 0)), int(settings.get("p1", 2)))   # (what defines the friction box: the converged reference keeps it)
    if "_REF" not in globals() or _REF[0] != rule:
        rs = {k: settings[k] for k in ("warm_p0",) if k in settings}
        _REF = (rule, run(rs, pgs_iters=300, pgs_exact=100, pgs_tol=0.0, pgs_normal_iters=rule[1]))
    kw = {"pgs_tol": float(settings["tol"])} if "tol" in settings else {}
    if "p1" in settings:
        kw["pgs_normal_iters"] = int(settings["p1"])
    e = np.max(np.abs(run(settings, **kw) - _REF[1]) / np.maximum(np.abs(_REF[1]), 1.0), axis=0)
    print(f"  closed loop, 1000 balancing env-steps vs the converged solve: median {np.median(e):.1e}  p90 {np.percentile(e, 90):.1e}  p99 {np.percentile(e, 99):.1e}  max {e.max():.1e}")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="C4")
    ap.add_argument("--envs", type=int, default=16384)
    ap.add_argument("--preroll", type=int, default=600)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--closed-loop", action="store_true")
    ap.add_argument("variants", nargs="+")
    a = ap.parse_args()
    L = O.use_laboratory()
    for v in a.variants:
        name, _, rest = v.partition(":")
        settings = {}
        for kv in filter(None, rest.split(",")):
            k, _, val = kv.partition("=")
            settings[k] = float(val) if k == "tol" else (int(val) if k in ("first", "p1") else SWITCHES[k][1](val))
        print(f"{name}  {settings}  [{a.workload}, {a.envs} envs, {a.steps} env-steps after {a.preroll}]")
        rounds(L, a, settings)
        if a.closed_loop:
            closed_loop(L, settings)
    apply(L, {})


if __name__ == "__main__":
    main()
