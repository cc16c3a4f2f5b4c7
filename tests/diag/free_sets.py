"""Which rows are free (strictly inside their box) at the solution of the contact problem, for the environments that needed an
exact solve and for those that did not (oracle, bench workload in its stationary regime): python tests/diag/free_sets.py [C4|C3|V1]"""
import sys, os, collections
ROOT=os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0,ROOT); sys.path.insert(0,os.path.join(ROOT,'tests'))
import numpy as np, bench
from oracle import oracle_py as O
wl=sys.argv[1] if len(sys.argv)>1 else 'C4'
class A:
    workload=wl; envs_per_gpu=4096; dtype="f64"; seed=42
    pgs_iters=None; pgs_normal_iters=None; pgs_tol=None; pgs_exact=None; runtime_model=False
cfg,_,_=bench.build_config(A,0,1)
O.build()
o=O.OracleSim(cfg,threads=8)
for _ in range(400): o.step(None)
o.solver_counts()
o.step(None)
sw,so=o.solver_counts()           # [iter, env] of the last step
q,qd=o.get_state()
hist=o.get_action_history(0)      # last applied action (normalised)
P=[o.get_params(f) for f in range(5)]
hard=(so[-1]>=1)                   # needed a solve in the last iteration
print(wl,"hard fraction in last iteration", hard.mean())
pat=collections.Counter(); pat_easy=collections.Counter()
for e in range(cfg.num_envs):
    tau=[2.5*hist[0,e],2.5*hist[1,e]]
    pr=O.contact_problem(cfg,q[:,e],qd[:,e],tau,P[0][:,e],P[1][:,e],P[2][:,e],P[3][:,e],float(P[4][0,e]))
    nr=pr["nr"]; lam=pr["lambda"]; box=pr["box"]; kind=pr["kind"]; body=pr["body"]
    free=[]
    for r in range(nr):
        lo=0.0 if kind[r]==0 else -box[r]; hi=np.inf if kind[r]==0 else box[r]
        if lam[r]>lo and lam[r]<hi: free.append(r)
    nf_n=sum(1 for r in free if kind[r]==0); nf_t=sum(1 for r in free if kind[r]==1); nf_j=sum(1 for r in free if kind[r]==2)
    ncont=int((kind[:nr]==0).sum())
    key=(ncont,nf_n,nf_t,nf_j)
    (pat if hard[e] else pat_easy)[key]+=1
print("hard lanes: (contacts, free normal, free tangential, free joint) -> count")
for k,v in pat.most_common(12): print("  ",k,v)
print("easy lanes:")
for k,v in pat_easy.most_common(12): print("  ",k,v)
