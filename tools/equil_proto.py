"""BASELINE config 5, fp16 matrix storage: would symmetric equilibration before rounding make it usable at rho = 1e-3?  (VERDICT r04 #8.)
CPU prototype on the oracle: the float32 PCG restatement on the ROUNDED matrices (what mpcg_pcg_solve_f16 computes: tests/fuzz_cases.py checks the
register-resident kernels bit for bit against the fp32 solve of the rounded matrices) — plain rounding, scalar (diagonal) equilibration, and
block-diagonal equilibration S_eq = E S E^T with E_k = chol(-S_kk)^-1 (unit diagonal blocks, stored exactly; the symmetric-stair preconditioner
of S_eq is then I - offdiag(S_eq): no second matrix to round); true relative residual against the ORIGINAL fp32 system.
   python tools/equil_proto.py        (CPU only; profiles/r05_f16_equilibration.txt)"""
import os, sys, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'oracle')); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import oracle as orc; orc.build()
from mpcgpu_amd import synth
n=14
def blocks(M,N): return np.nan_to_num(np.asarray(M,np.float64)).reshape(N,3,n,n).transpose(0,1,3,2).copy()   # [k][col] row-major matrices
def pack(Bk,N): return Bk.transpose(0,1,3,2).reshape(-1).copy()
def run(N, rho, seed, K, tol=0.0):
    k=synth.make_kkt(N,1,seed); S,P,g=synth.form_schur(k,rho=rho,precond="ss"); S,P,g=S[0],P[0],g[0]
    Sd=synth.bd_to_dense(S.astype(np.float64),N)
    tr=lambda lam: float(np.linalg.norm(g-Sd@np.asarray(lam,np.float64))/np.linalg.norm(g))
    z=np.zeros(n*N,np.float32)
    out={}
    r=orc.pcg(S,P,g,z,N,K,tol,"ss"); out['f32']=(tr(r['lam']),r['iters'])
    h=lambda a: a.astype(np.float16).astype(np.float32)
    r=orc.pcg(h(S),h(P),g,z,N,K,tol,"ss"); out['f16 plain']=(tr(r['lam']),r['iters'])
    # diagonal (Jacobi scalar) equilibration
    Sb=blocks(S,N); Pb=blocks(P,N)
    d=np.array([1/np.sqrt(-np.diag(Sb[k_,1])) for k_ in range(N)])      # [N][n]
    Sq=Sb.copy(); Pq=Pb.copy()
    for k_ in range(N):
        for col,kk in ((0,k_-1),(1,k_),(2,k_+1)):
            if 0<=kk<N:
                Sq[k_,col]=d[k_][:,None]*Sb[k_,col]*d[kk][None,:]
                Pq[k_,col]=Pb[k_,col]/d[k_][:,None]/d[kk][None,:]
    gq=(g.reshape(N,n)*d).reshape(-1).astype(np.float32)
    r=orc.pcg(h(pack(Sq,N).astype(np.float32)),h(pack(Pq,N).astype(np.float32)),gq,z,N,K,tol,"ss")
    lam=(r['lam'].reshape(N,n)*d).reshape(-1); out['f16 diag-equilibrated']=(tr(lam),r['iters'])
    # block equilibration: -S_kk = L L^T, E = L^-1
    E=[np.linalg.inv(np.linalg.cholesky(-Sb[k_,1])) for k_ in range(N)]
    Se=np.zeros_like(Sb); Pe=np.zeros_like(Sb)
    for k_ in range(N):
        Se[k_,1]=-np.eye(n); Pe[k_,1]=-np.eye(n)
        if k_>0:
            Se[k_,0]=E[k_]@Sb[k_,0]@E[k_-1].T; Pe[k_,0]=-Se[k_,0]
        if k_<N-1:
            Se[k_,2]=E[k_]@Sb[k_,2]@E[k_+1].T; Pe[k_,2]=-Se[k_,2]
    ge=np.concatenate([E[k_]@g.reshape(N,n)[k_] for k_ in range(N)]).astype(np.float32)
    for name,cast in (('f32 block-equilibrated',lambda a:a),('f16 block-equilibrated',h)):
        r=orc.pcg(cast(pack(Se,N).astype(np.float32)),cast(pack(Pe,N).astype(np.float32)),ge,z,N,K,tol,"ss")
        lam=np.concatenate([E[k_].T@r['lam'].reshape(N,n)[k_] for k_ in range(N)]); out[name]=(tr(lam),r['iters'])
    return out
for N,K in ((128,167),(512,67)):
  for rho in (1e-3,1e-2,1.0):
    for seed in (900000, 900001):
        o=run(N,rho,seed,K)
        print(f"N={N} rho={rho} seed={seed} K={K}: "+" | ".join(f"{a}: {b[0]:.3g}" for a,b in o.items()))
  o=run(N,1e-3,900000,3000,1e-5); print(f"N={N} rho=1e-3 tol 1e-5 cap 3000: "+" | ".join(f"{a}: res {b[0]:.3g} it {b[1]}" for a,b in o.items()))
