"""Does the ORDER of the trajectories in a call matter when they leave the loop at different iterations?  1024 synthetic N=128 systems,
lambda0 = solution + noise of amplitude 1e-6..1e-1 (iterations 5..167): kernel time in the given order, sorted by descending iteration
count (long solves first), ascending (long solves last) — all with the library's dispatch-order hint off — and in the given order with the
hint on (option "sched_hint": the previous call's counts order the dispatch)."""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench
from mpcgpu_amd import PcgSolver, pcg_config
N = int(sys.argv[1]) if len(sys.argv) > 1 else 128
B = 1024
B = 4096 if N <= 32 else (2048 if N <= 64 else 1024)
cap = {32: 173, 128: 167, 256: 118, 512: 67}.get(N, 167)
dev = torch.device("cuda", 0)
sol = PcgSolver(N, max_batch=B)
dS, dP, dg = bench.build_inputs(sol, N, B, 0, "ss", dev)
lam_star = torch.zeros(B, 14 * N, device=dev)
sol.solve(dS, dP, dg, lam_star, pcg_config(pcg_exit_tol=1e-9, pcg_max_iter=4000))
gen = torch.Generator(device=dev); gen.manual_seed(1)
for lo, hi, label in ((-6.0, -1.0, "amplitude 1e-6..1e-1"), (-7.0, -3.5, "amplitude 1e-7..3e-4 (mostly short, a few long)")):
    amp = 10 ** (lo + (hi - lo) * torch.rand(B, 1, device=dev, generator=gen))
    lam_w = lam_star + amp * lam_star.abs().amax(dim=1, keepdim=True) * torch.randn(B, 14 * N, device=dev, generator=gen)
    cfg = pcg_config(pcg_exit_tol=1e-4, pcg_max_iter=cap)
    lam = lam_w.clone()
    it, ex = sol.solve(dS, dP, dg, lam, cfg)
    iters = it.cpu().numpy().astype(np.int64)
    print(f"{label}: iterations min {iters.min()} mean {iters.mean():.1f} max {iters.max()}, {int((iters == cap).sum())} at the cap")
    for name, perm in (("given order, no hint", np.arange(B)), ("given order, hinted", np.arange(B)), ("long solves first", np.argsort(-iters, kind="stable")),
                       ("long solves last", np.argsort(iters, kind="stable"))):
        sol.set_option("sched_hint", 1 if name.endswith("hinted") else 0)     # (the library's own dispatch order from the previous call's counts)
        p = torch.from_numpy(perm).to(dev)
        S_, P_, g_, l0 = dS[p].contiguous(), dP[p].contiguous(), dg[p].contiguous(), lam_w[p].contiguous()
        l_ = l0.clone()
        def go():
            l_.copy_(l0); sol.solve(S_, P_, g_, l_, cfg)
        ms = bench.timed(go, 5, warm=2) - bench.timed(lambda: l_.copy_(l0), 5, warm=2)
        print(f"   {name:22s}: {ms*1e3:7.1f} us   ({B/ms/1e3:.2f} M linsolves/s)", flush=True)
        del S_, P_, g_, l0, l_
