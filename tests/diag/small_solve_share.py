"""Share of the exact solves that a dual solve of small free sets would serve (oracle experiment, docs/studies/round4_solver.md):
python tests/diag/small_solve_share.py [C4|C3|V1]"""
import sys, os
ROOT=os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0,ROOT); sys.path.insert(0,os.path.join(ROOT,'tests'))
import numpy as np, bench
from oracle import oracle_py as O
wl=sys.argv[1]
class A:
    workload=wl; envs_per_gpu=8192; dtype="f64"; seed=42
    pgs_iters=None; pgs_normal_iters=None; pgs_tol=None; pgs_exact=None; runtime_model=False
cfg,_,_=bench.build_config(A,0,1)
O.use_laboratory().orc_set_experimental_small(1)   # the laboratory build: oracle/Makefile
o=O.OracleSim(cfg,threads=8)
for _ in range(400): o.step(None)
o.solver_counts()
tot=gen=0; wave_any=0; wave_gen=0; W=cfg.num_envs//64; ws=0
for _ in range(10):
    o.step(None); sw,so=o.solver_counts(); sm=o.small_solve_counts()
    so=so.astype(int); sm=sm.astype(int); g=so-sm
    tot+=so.sum(); gen+=g.sum()
    gw=g.reshape(10,W,64); sw_=so.reshape(10,W,64)
    wave_any+=(sw_.max(axis=2)>0).sum(); wave_gen+=(gw.max(axis=2)>0).sum(); ws+=10*W
print(wl, "solves", tot, "of which regularised", gen, f"({100*gen/tot:.2f} %);  wave-iterations with a solve {wave_any/ws:.3f}, with a regularised solve {wave_gen/ws:.4f}")
