import sys; sys.path.insert(0,'.'); sys.path.insert(0,'tests')
import numpy as np, torch
from helpers import make_config
from gym_os2r_amd import abi
from gym_os2r_amd.sim import HipSim
from oracle import oracle_py as o
n=512
cfg,task,model=make_config('fixed','BalancingV2',True,num_envs=n,contact=True,auto_reset=False,dtype=abi.F64)
rng=np.random.default_rng(11)
nq=model['nq']
q=rng.uniform(-1.2,1.2,(nq,n)); q[0]=rng.uniform(-0.04,0.3,n); qd=rng.uniform(-8,8,(nq,n))
act=rng.uniform(-1,1,(n,2))
for sub in (1,2,10):
    cfg.substeps=sub
    sim,orc=HipSim(cfg),o.OracleSim(cfg)
    sim.set_state(q,qd); orc.set_state(q,qd)
    sim.step(torch.as_tensor(act)); orc.step(act)
    q2,qd2=(t.cpu().numpy() for t in sim.get_state()); oq,oqd=orc.get_state()
    err=np.max(np.abs(qd2-oqd)/np.maximum(np.abs(oqd),1),axis=0)
    bad=np.argsort(err)[-5:]
    print('substeps',sub,'worst envs',bad,err[bad])
    for e in bad[-2:]:
        print(' env',e,'q0',q[:,e],'qd0',qd[:,e],'act',act[e]); print('   gpu qd',qd2[:,e],' orc qd',oqd[:,e])
        _,_,rw,ow=o.dynamics(cfg.model,q[:,e],qd[:,e],np.zeros(nq)); a,pw,dep=o.contact_points(cfg.model,rw,ow); print('   active',a,'depth',dep)
