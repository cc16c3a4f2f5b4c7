"""Per physics-iteration index of an env-step: the share of environments that need an exact solve (one / two or more), and a
wave's solves and sweeps (oracle, bench workload in its stationary regime).  python tests/diag/iter_stats.py MODE [C4|C3|V1]
MODE: 1 the specification (solver state carried across env-steps), 2 round 3 (forgotten between env-steps), 0 no warm start."""
import sys, os
ROOT=os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0,ROOT); sys.path.insert(0,os.path.join(ROOT,'tests'))
import numpy as np, bench
from oracle import oracle_py as O
mode=int(sys.argv[1]); wl=sys.argv[2] if len(sys.argv)>2 else 'C4'
class A:
    workload=wl; envs_per_gpu=8192; dtype="f64"; seed=42
    pgs_iters=None; pgs_normal_iters=None; pgs_tol=None; pgs_exact=None; runtime_model=False
cfg,_,_=bench.build_config(A,0,1)
O.use_laboratory().orc_set_experimental_warm(mode,0)   # the laboratory build: oracle/Makefile
o=O.OracleSim(cfg,threads=8)
for _ in range(400): o.step(None)
o.solver_counts()
S=[];W=[]
for _ in range(10):
    o.step(None); sw,so=o.solver_counts(); S.append(so.astype(int)); W.append(sw.astype(int))
S=np.stack(S); W=np.stack(W)   # [step, iter, env]
n=S.shape[2]//64
Sw=S.reshape(10,10,n,64); Ww=W.reshape(10,10,n,64)
print(f"mode {mode} {wl}: per iteration index: P(lane solves>=1), P(>=2), wave max solves mean, wave max sweeps mean")
for it in range(10):
    s=Sw[:,it]; w=Ww[:,it]
    print(it, f"{(s>=1).mean():.4f} {(s>=2).mean():.5f}  {s.max(axis=-1).mean():.3f}  {w.max(axis=-1).mean():.3f}")
