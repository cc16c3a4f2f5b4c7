"""round 3 experiment: does splitting the batch over S handles on S streams -- each split advancing on
its own, so that a split waits for its own slowest wave only -- raise the throughput of the bench workload?
  python tools/sessions/r3_splits.py [envs] [steps]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
import bench
from gym_os2r_amd.sim import HipSim

N = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
K = int(sys.argv[2]) if len(sys.argv) > 2 else 500


def run(S, order="round_robin"):
    class A:
        workload = "C4"; envs_per_gpu = N // S; dtype = "f64"; seed = 42; pgs_iters = None; pgs_exact = None; pgs_normal_iters = None
        pgs_tol = None; runtime_model = False
    sims, streams, bufs = [], [], []
    for i in range(S):
        cfg, _, _ = bench.build_config(A, i, S)          # rank i of S: env_offset = i * N/S
        sims.append(HipSim(cfg))
        streams.append(torch.cuda.Stream())
        n, D = cfg.num_envs, cfg.task.obs_dim
        bufs.append([torch.empty(n, D, dtype=torch.float64, device="cuda"), torch.empty(n, dtype=torch.float64, device="cuda"),
                     torch.empty(n, dtype=torch.uint8, device="cuda"), torch.empty(n, D, dtype=torch.float64, device="cuda")])
    for i in range(S):
        with torch.cuda.stream(streams[i]):
            for _ in range(1200):
                sims[i].step_into(None, *bufs[i])
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    if order == "round_robin":
        for _ in range(K):
            for i in range(S):
                with torch.cuda.stream(streams[i]):
                    sims[i].step_into(None, *bufs[i])
    else:                                                 # all of a split's steps at once
        for i in range(S):
            with torch.cuda.stream(streams[i]):
                for _ in range(K):
                    sims[i].step_into(None, *bufs[i])
    t_enq = time.perf_counter() - t0
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    for s in sims:
        s.close()
    return N * K / dt / 1e6, dt / K * 1e6, t_enq / K * 1e6


for S in [int(x) for x in os.environ.get("SPLITS", "1,2,4,8,16").split(",")]:
    v, us, enq = run(S)
    print(f"splits {S:3d}: {v:7.1f} M env-steps/s, {us:7.2f} us per step of the whole batch (host enqueue {enq:6.2f} us per step)", flush=True)
