import os, sys
import numpy as np, torch
sys.path.insert(0, os.getcwd())
import bench
from mpcgpu_amd import PcgSolver, pcg_config, synth
N,B=128,1024
sol=PcgSolver(N,max_batch=B); sol.set_option("cluster",0)
dS,dP,dg=bench.build_inputs(sol,N,B,0,"ss",torch.device("cuda",0))
lam=torch.zeros(B,14*N,device="cuda")
cfg=pcg_config(pcg_exit_tol=0.0,pcg_max_iter=167)
ts=[]
for i in range(6):
    lam.zero_(); e0,e1=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
    e0.record(); it,ex=sol.solve(dS,dP,dg,lam,cfg); e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1))
print("fixed 167 iterations: %.3f ms, its %d"%(np.median(ts[1:]), it.sum().item()))
