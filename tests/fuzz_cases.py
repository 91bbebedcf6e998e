"""Randomised differential run of the PCG kernel families — shared by tools/fuzz_families.py (any number of cases, on the GPU box) and
tests/test_gpu_fuzz.py (a seeded slice inside the `-m gpu` suite, VERDICT r04 #2a).
For random (N, batch, preconditioner, warm start, iteration cap) the default kernel of the call is compared with the float64 oracle iterate
after the same number of iterations (inside the float32 band of tests/util.py), iteration counts and exit flags checked; every kernel family
the launch policy can pick shows up (5 row-per-lane, 6 lane-pair, 7 clustered lane-pair).  The same case then runs with fp16 matrix storage:
on the register-resident families bit-identical to the fp32 solve of the rounded matrices by the same kernel."""
import collections
import time

import numpy as np
import torch

import oracle as orc
from mpcgpu_amd import PcgSolver, pcg_config, synth
from util import fp32_band, relinf


def run(cases=120, seed=7, verbose=True):
    orc.build()
    rng = np.random.default_rng(seed)
    dev = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
    fam_count, worst, bad, breakdown, marginal = collections.Counter(), 0.0, 0, 0, 0
    f16_count = collections.Counter()
    worst16 = 0.0
    f16_bitwise = 0
    f16_chaotic = 0
    warm_cases = 0
    t0 = time.time()
    for ci in range(cases):
        N = int(rng.choice([rng.integers(2, 33), rng.integers(33, 129), rng.integers(129, 400)], p=[0.4, 0.4, 0.2]))
        B = int(rng.integers(1, 7))
        pc = str(rng.choice(["ss", "jacobi"]))
        K = int(rng.integers(1, min(40, 14 * N)))      # (never past the dimension of the system: exact convergence, then 0 / 0 in float32 with exit_tol = 0)
        k = synth.make_kkt(N, B, int(rng.integers(1 << 30)))
        S, P, g = synth.form_schur(k, precond=pc, dtype=np.float32, poison_unused=True)
        warm = rng.random() < 0.5
        warm_cases += int(warm)
        lam0 = (0.1 * rng.standard_normal((B, 14 * N))).astype(np.float32) if warm else np.zeros((B, 14 * N), np.float32)
        sol = PcgSolver(N, max_batch=B)
        if N <= 32 and rng.random() < 0.25:
            sol.set_option("pcg_lpk", 1)                   # the lane-pair kernel's half build (the policy's choice for throughput-sized calls at 16 < N <= 32)
        lam = dev(lam0.copy())
        it, ex = sol.solve(dev(S), dev(P), dev(g), lam, pcg_config(pcg_exit_tol=0.0, pcg_max_iter=K), pc)
        torch.cuda.synchronize()
        fam = sol.get_option("last_kernel_family")
        fam_count[fam] += 1
        lam_h, it_h, ex_h = lam.cpu().numpy(), it.cpu().numpy(), ex.cpu().numpy()
        ok = (it_h == K).all() and (ex_h == 1).all()
        for b in range(B):
            Sz, Pz = np.nan_to_num(S[b]), np.nan_to_num(P[b])
            # float32 CG iterated far past convergence with exit_tol = 0 breaks down by construction (eta underflows to 0, alpha = 0 / 0): if
            # the CPU float32 restatement does, the case says nothing about the kernel
            if not np.isfinite(orc.pcg(Sz, Pz, g[b], lam0[b], N, K, 0.0, pc)["lam"]).all():
                breakdown += 1
                continue
            ok = ok and bool(np.isfinite(lam_h[b]).all())
            ref = orc.pcg(Sz.astype(np.float64), Pz.astype(np.float64), g[b].astype(np.float64), lam0[b].astype(np.float64), N, K, 0.0, pc)["lam"]
            band = fp32_band(orc, Sz, Pz, g[b], lam0[b], N, K, pc, ref)
            e = relinf(lam_h[b], ref)
            tol = max(2e-5 if K <= 3 else 1e-3, 4 * band)
            worst = max(worst, e / tol)
            marginal += int(tol < e <= 2 * tol)               # (the test-suite tolerance is a heuristic: report, fail from 2x)
            ok = ok and e <= 2 * tol
        # fp16 matrix storage on the same case: on the register-resident families the blocks are converted once at the load, so the solve must be
        # BIT-identical to the fp32 solve of the rounded matrices; elsewhere (round-1 kernel with mixed-precision FMAs) inside the band
        S16, P16 = sol.to_f16(dev(np.nan_to_num(S))), sol.to_f16(dev(np.nan_to_num(P)))
        l16, l32 = dev(lam0.copy()), dev(lam0.copy())
        cfgk = pcg_config(pcg_exit_tol=0.0, pcg_max_iter=K)
        sol.solve_f16(S16, P16, dev(g), l16, cfgk, pc)
        fam16 = sol.get_option("last_kernel_family")
        sol.set_option("pcg_lqb", 0)                       # (fp16 storage lives in the lane-pair kernels: the fp32 twin of this comparison runs there too)
        sol.solve(S16.float().contiguous(), P16.float().contiguous(), dev(g), l32, cfgk, pc)
        torch.cuda.synchronize()
        f16_count[fam16] += 1
        fam32 = sol.get_option("last_kernel_family")
        a16, a32 = l16.cpu().numpy(), l32.cpu().numpy()
        Sr, Pr = S16.float().cpu().numpy(), P16.float().cpu().numpy()
        same_kernel = fam16 in (6, 7) and fam32 == fam16 and np.isfinite(a32).all()
        if same_kernel and not np.array_equal(a16, a32):
            ok = False
            if verbose: print(f"  fp16 storage differs from the fp32 solve of the rounded matrices (family {fam16})")
        f16_bitwise += int(same_kernel)
        for b in range(B):
            if same_kernel:
                break           # bit-identical to the fp32 kernel on the rounded matrices: that kernel's own check (above, on the caller's system) is the check
            cpu32 = orc.pcg(Sr[b], Pr[b], g[b], lam0[b], N, K, 0.0, pc)["lam"]
            if not np.isfinite(cpu32).all():
                continue
            r64 = orc.pcg(Sr[b].astype(np.float64), Pr[b].astype(np.float64), g[b].astype(np.float64), lam0[b].astype(np.float64), N, K, 0.0, pc, hist=True)
            ref = r64["lam"]
            # Rounding to fp16 can make S or Pinv INDEFINITE (profiles/r05_fuzz.txt: case 2143 of seed 4041, N = 4: largest eigenvalue of the rounded
            # -S +0.15).  CG then passes near-breakdowns — eta grows by orders of magnitude in one iteration, in float64 too — and the iterates behind
            # one are not comparable between any two arithmetics (the CPU float32 and float64 restatements differ by 38 % there): inconclusive.
            eh = np.abs(np.asarray(r64["eta_hist"], np.float64))
            if len(eh) > 1 and (eh[1:] > 30.0 * np.maximum(eh[:-1], 1e-300)).any():
                f16_chaotic += 1
                continue
            band = fp32_band(orc, Sr[b], Pr[b], g[b], lam0[b], N, K, pc, ref)
            e = relinf(a16[b], ref)
            # the ROUNDED system can be much worse conditioned than the caller's (rho = 1e-3: entries lose 11 bits): where the CPU float32 restatement
            # itself leaves the band on it, the band says nothing — the yardstick is then three times that restatement's own error
            tol = max(2e-5 if K <= 3 else 1e-3, 4 * band, 3.0 * relinf(cpu32, ref))
            worst16 = max(worst16, e / tol)
            if not (np.isfinite(a16[b]).all() and e <= 4 * tol):      # (chaotic on the badly conditioned rounded systems: this check is for gross errors)
                ok = False
                if verbose: print(f"  fp16 storage: trajectory {b} off the float64 iterate of the rounded system by {e:.2e} (tolerance {tol:.2e}, family {fam16})")
        if not ok:
            bad += 1
            if verbose: print(f"MISMATCH case {ci}: N={N} B={B} {pc} K={K} family {fam}: iters {it_h.tolist()} exit {ex_h.tolist()} finite {bool(np.isfinite(lam_h).all())}", flush=True)
    return {"cases": cases, "seed": seed, "seconds": time.time() - t0, "families": dict(sorted(fam_count.items())), "worst_error_over_tolerance": worst,
            "marginal_trajectories": marginal, "breakdown_skipped": breakdown, "f16_families": dict(sorted(f16_count.items())), "f16_bitwise_cases": f16_bitwise,
            "f16_worst_error_over_tolerance": worst16, "f16_near_breakdown_skipped": f16_chaotic, "warm_start_cases": warm_cases, "mismatches": bad}
