"""Real IIWA-14 Schur systems at the BASELINE horizon N = 128 (cond(-S) ~ 1e7: the regime bench.py's iiwa_run reports), committed as
tests/golden/iiwa_kkt_N128.npz by tests/make_iiwa_golden.py from the reference's own trajectory data (examples/trajfiles/0_0_*), with
float64 answers made by an independent dense numpy PCG (tests/make_golden.py:dense_pcg — no code shared with oracle/ or the kernels).

 * CPU: the C oracle against those answers; and an independent THIRD-PARTY cross-check of the oracle's PCG: scipy.sparse.linalg.cg with
   the symmetric-stair operator as preconditioner M (VERDICT r2 #6d — removes "the same author wrote both sides" from the oracle itself);
 * GPU: the default kernel at N = 128 against the float64 iterates inside the fp32 band (cold and warm start), and — for the warm-start
   regime the bench reports — the TRUE residual of every solved trajectory does not grow."""
import os

import numpy as np
import pytest

from conftest import GOLDEN
from mpcgpu_amd import synth
from util import fp32_band, relinf, rel_residual

n, N = 14, 128


@pytest.fixture(scope="module")
def G():
    d = np.load(os.path.join(GOLDEN, "iiwa_kkt_N128.npz"))
    return {k: d[k] for k in d.files}


def test_oracle_on_real_n128_systems(orc, G):
    for i in range(2):
        S, P, g = G[f"s{i}_S"], G[f"s{i}_Pinv"], G[f"s{i}_gamma"]
        assert 5e6 < float(G[f"s{i}_cond"]) < 5e7
        for start, lam0 in (("cold", np.zeros(n * N)), ("warm", G[f"s{i}_lam_warm0"].astype(np.float64))):
            for K in (10, 40):
                want = G[f"s{i}_lam_{start}_K{K}"]
                got = orc.pcg(S.astype(np.float64), P.astype(np.float64), g.astype(np.float64), lam0, N, K, 0.0, "ss")["lam"]
                assert relinf(got, want) < 1e-9, (i, start, K, relinf(got, want))
        # the float64 direct solution solves the system
        assert rel_residual(S, g, G[f"s{i}_lam_direct"], N) < 1e-9


@pytest.mark.parametrize("which", ["synthetic N=32 golden", "real IIWA N=128"])
def test_oracle_pcg_against_scipy_cg(orc, G, which):
    """scipy's preconditioned CG (float64) on the SAME operator pair, iterate by iterate.  S and Pinv are stored negated (negative definite):
    scipy gets A = -S, M = -Pinv, b = -gamma — the same linear system and the same Krylov recurrences."""
    import scipy.sparse.linalg as sla
    from util import golden
    if which.startswith("synthetic"):
        g_ = golden(32)
        S, P, g, Nn = g_["S"], g_["Pinv"], g_["gamma"], 32
    else:
        S, P, g, Nn = G["s0_S"], G["s0_Pinv"], G["s0_gamma"], N
    Sd = synth.bd_to_dense(np.nan_to_num(S).astype(np.float64), Nn)
    Pd = synth.bd_to_dense(np.nan_to_num(P).astype(np.float64), Nn)
    A = sla.aslinearoperator(-Sd)
    M = sla.aslinearoperator(-Pd)
    b = -g.astype(np.float64)
    for K in (1, 5, 20):
        xs = []
        x, info = sla.cg(A, b, x0=np.zeros(n * Nn), rtol=0.0, atol=0.0, maxiter=K, M=M, callback=lambda xk: xs.append(xk.copy()))
        assert len(xs) == K
        ours = orc.pcg(S.astype(np.float64), P.astype(np.float64), g.astype(np.float64), np.zeros(n * Nn), Nn, K, 0.0, "ss")
        assert relinf(ours["lam"], xs[-1]) < 1e-8, (which, K, relinf(ours["lam"], xs[-1]))
        assert ours["iters"] == K


@pytest.mark.gpu
def test_default_kernel_on_real_n128_systems(orc, G):
    import torch
    from mpcgpu_amd import PcgSolver, pcg_config
    dev = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
    for i in range(2):
        S, P, g = G[f"s{i}_S"], G[f"s{i}_Pinv"], G[f"s{i}_gamma"]
        for start, lam0 in (("cold", np.zeros(n * N, np.float32)), ("warm", G[f"s{i}_lam_warm0"])):
            for K in (10, 40):
                sol = PcgSolver(N, max_batch=1)
                lam = dev(lam0.reshape(1, -1).astype(np.float32))
                it, ex = sol.solve(dev(S.reshape(1, -1)), dev(P.reshape(1, -1)), dev(g.reshape(1, -1)), lam, pcg_config(pcg_exit_tol=0.0, pcg_max_iter=K), "ss")
                torch.cuda.synchronize()
                assert sol.get_option("last_kernel_family") == 11 and int(it.item()) == K
                want = G[f"s{i}_lam_{start}_K{K}"]
                band = fp32_band(orc, S, P, g, lam0, N, K, "ss", want)
                err = relinf(lam.cpu().numpy()[0], want)
                assert err <= max(1e-3, 4 * band), (i, start, K, err, band)


@pytest.mark.gpu
def test_warm_started_solves_do_not_increase_the_true_residual(G):
    """The regime bench.py reports (iiwa_run.warm): lambda0 = multipliers of a slightly different system, cap 167, exit_tol 1e-4.  The
    |eta| test is loose on these systems (cond 1e7), so check what it does not: per trajectory, ||gamma - S lambda|| after the solve is
    not larger than before it (float64 evaluation on the dense matrix), for a batch of perturbed warm starts."""
    import torch
    from mpcgpu_amd import PcgSolver, pcg_config
    rng = np.random.default_rng(5)
    B = 24
    S = np.stack([G[f"s{b % 2}_S"] for b in range(B)])
    P = np.stack([G[f"s{b % 2}_Pinv"] for b in range(B)])
    g = np.stack([G[f"s{b % 2}_gamma"] for b in range(B)])
    lam_star = [G[f"s{b % 2}_lam_direct"] for b in range(B)]
    amp = np.logspace(-3, -0.5, B)
    lam0 = np.stack([(lam_star[b] * (1 + amp[b] * rng.standard_normal(n * N))).astype(np.float32) for b in range(B)])
    dev = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
    sol = PcgSolver(N, max_batch=B)
    lam = dev(lam0)
    it, ex = sol.solve(dev(S), dev(P), dev(g), lam, pcg_config(pcg_exit_tol=1e-4, pcg_max_iter=167), "ss")
    torch.cuda.synchronize()
    lam = lam.cpu().numpy()
    it = it.cpu().numpy()
    grew = []
    for b in range(B):
        r0, r1 = rel_residual(S[b], g[b], lam0[b], N), rel_residual(S[b], g[b], lam[b], N)
        if it[b] > 0 and r1 > r0 * (1 + 1e-3):
            grew.append((b, r0, r1, int(it[b])))
    assert not grew, grew
