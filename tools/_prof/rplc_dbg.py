import os, sys, time, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from mpcgpu_amd import _lib as _L
if os.environ.get('AB_LIB'): _L.LIB_PATH = os.environ['AB_LIB']
from mpcgpu_amd import PcgSolver, pcg_config, synth
N=int(sys.argv[1]) if len(sys.argv)>1 else 64; B=int(sys.argv[2]) if len(sys.argv)>2 else 1
k=synth.make_kkt(N,B,1); S,P,g=synth.form_schur(k,dtype=np.float64)
dev=lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
dS,dP,dg=dev(S),dev(P),dev(g)
for fix,l2 in ((0,1),):
    sol=PcgSolver(N,max_batch=B); sol.set_option("cluster_fixup",fix); sol.set_option("cluster_l2",l2)
    lam=torch.zeros(B,14*N,dtype=torch.float64,device="cuda")
    for rep in range(1):
        lam.zero_(); torch.cuda.synchronize(); t0=time.perf_counter()
        it,ex=sol.solve_f64(dS,dP,dg,lam,pcg_config(pcg_exit_tol=0.0,pcg_max_iter=40)); torch.cuda.synchronize()
        print("fixup",fix,"l2",l2,"ms %.3f"%((time.perf_counter()-t0)*1e3),"iters",it.cpu().numpy()[:4],"exit",ex.cpu().numpy()[:4],"fam",sol.get_option("last_kernel_family"))

np.set_printoptions(linewidth=200, suppress=True)
L=lam.cpu().numpy()[0][:512].reshape(2,64,4)
for g_ in range(2):
    print('member',g_); print(L[g_][:48].T)
