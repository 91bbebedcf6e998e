import os, sys, numpy as np, torch
R=os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0,R); sys.path.insert(0,R+'/tests'); sys.path.insert(0,R+'/oracle')
import oracle as orc; orc.build()
from mpcgpu_amd import PcgSolver, pcg_config, synth
from util import relinf
dev=lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
rng=np.random.default_rng(3)
import sys as _s
CASES=((7,"ss"),) if len(_s.argv)>1 else ((5,"ss"),(5,"jacobi"),(9,"ss"))
for N,pc in CASES:
  for seed in range(8 if len(_s.argv)>1 else 3):
    k=synth.make_kkt(N,1,1000+seed); S,P,g=synth.form_schur(k,precond=pc,dtype=np.float64); S,P,g=S[0],P[0],g[0]
    lam0=np.zeros(14*N)
    h=np.abs(orc.pcg(S,P,g,lam0,N,45,0.0,pc,hist=True)["eta_hist"])
    sol=PcgSolver(N,max_batch=1)
    for K in ((18,20,21,22,23,24,26) if len(_s.argv)>1 else (10,20,25,30,35,40)):
        lam=dev(lam0.reshape(1,-1).copy()); sol.solve_f64(dev(S.reshape(1,-1)),dev(P.reshape(1,-1)),dev(g.reshape(1,-1)),lam,pcg_config(pcg_exit_tol=0.0,pcg_max_iter=K),pc); torch.cuda.synchronize()
        ref=orc.pcg(S,P,g,lam0,N,K,0.0,pc)["lam"]
        band=max(relinf(orc.pcg(S,P,g*(1+1.1e-16*rng.standard_normal(g.shape)),lam0,N,K,0.0,pc)["lam"],ref) for _ in range(24 if len(_s.argv)>1 else 10))
        print(f"N={N} {pc} seed {seed} K={K}: eta[K]/eta[0] {h[K]/h[0]:.1e}  GPU-vs-oracle {relinf(lam.cpu().numpy()[0],ref):.1e}  oracle's own 1-ulp band (10 trials) {band:.1e}")
