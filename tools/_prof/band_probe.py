import sys, os, numpy as np, torch
R=os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0,R); sys.path.insert(0,os.path.join(R,'tests'))
import test_gpu_scipy as T
from mpcgpu_amd import PcgSolver, pcg_config, synth
from util import relinf
n=14
for N in (32,128,256,512):
  for pc in ("ss","jacobi"):
    k = synth.make_kkt(N, 1, 8200 + N); S,P,g=(a[0] for a in synth.form_schur(k, precond=pc))
    lam0=np.zeros(n*N,np.float32); KM=50
    xs=T.scipy_iterates(S,P,g,lam0,N,pc,KM)
    rng=np.random.default_rng(1); pert=[]
    for t in range(4):
        Sp=(S.astype(np.float64)*(1+6e-8*rng.standard_normal(S.shape))); gp=(g.astype(np.float64)*(1+6e-8*rng.standard_normal(g.shape)))
        pert.append(T.scipy_iterates(Sp,P,gp,lam0,N,pc,KM))
    sol=PcgSolver(N,max_batch=1); dS,dP,dg=(torch.from_numpy(a.reshape(1,-1).copy()).cuda() for a in (S,P,g))
    for K in (3,10,25,50):
        lam=torch.zeros(1,n*N,device="cuda"); sol.solve(dS,dP,dg,lam,pcg_config(pcg_exit_tol=0.0,pcg_max_iter=K),pc); torch.cuda.synchronize()
        e=relinf(lam.cpu().numpy()[0],xs[K-1]); band=max(relinf(p_[K-1],xs[K-1]) for p_ in pert)
        print(f"N={N} {pc} K={K}: E {e:.2e} band(1-ulp input perturbation, scipy) {band:.2e} ratio {e/band:.1f}")
