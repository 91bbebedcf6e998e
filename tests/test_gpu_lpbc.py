"""GPU: the clustered register-resident kernels for fp32 horizons beyond 128 knots — round 3's clustered lane-PAIR kernel
(pcg_lpk_cluster.hip.h, the default: family 7) and round 2's clustered lane-per-block kernel (pcg_lpb_cluster.hip.h, "cluster_lpk" = 0:
family 4), every test on both: G = ceil(N / 128) workgroups per trajectory, one hand-off per matrix pass.  tests/test_gpu_cluster.py runs it with forced
member counts next to the row-triple cluster kernel; here: the automatic policy on ragged horizons, batches that need several
launches, determinism and batch-composition independence."""
import numpy as np
import pytest
import torch

from mpcgpu_amd import synth
from util import fp32_band, relinf, rel_residual

pytestmark = pytest.mark.gpu
n = 14


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


@pytest.fixture(params=["lpkc", "lpbc"])
def kern(request):
    """(value of the "cluster_lpk" option, expected "last_kernel_family")"""
    return (-1, 7) if request.param == "lpkc" else (0, 4)


@pytest.mark.parametrize("N", [129, 131, 255, 257, 300, 512, 640])
@pytest.mark.parametrize("pc", ["ss", "jacobi"])
@pytest.mark.parametrize("l2", [1, 0])
def test_auto_policy_ragged_horizons_vs_oracle(orc, kern, N, pc, l2):
    """Default handle, no knobs: N > 128 runs family 4 with G = ceil(N / 128) members of floor/ceil(N / G) knots."""
    from mpcgpu_amd import PcgSolver, pcg_config
    B = 2
    k = synth.make_kkt(N, B, 9100 + N)
    S, Pinv, g = synth.form_schur(k, poison_unused=True)
    lam0 = np.random.default_rng(N).normal(0, 0.2, (B, n * N)).astype(np.float32)
    sol = PcgSolver(N, max_batch=B)
    sol.set_option("cluster_lpk", kern[0])
    sol.set_option("cluster_l2", l2)               # 1 (default): L2-resident hand-offs inside an XCD; 0: write-through hand-offs
    dS, dP, dg = dev(S), dev(Pinv), dev(g)
    for K in (2, 25):
        lam = dev(lam0)
        it, ex = sol.solve(dS, dP, dg, lam, pcg_config(pcg_exit_tol=0.0, pcg_max_iter=K), pc)
        torch.cuda.synchronize()
        assert sol.get_option("last_kernel_family") == kern[1] and sol.get_option("last_kernel_cluster") == (N + 127) // 128
        assert kern[1] != 7 or sol.get_option("last_kernel_lds_bytes") == sol.lib.mpcg_pcg_lds_bytes(14, N)
        assert (it.cpu().numpy() == K).all() and (ex.cpu().numpy() == 1).all()
        for t in range(B):
            Sz, Pz = np.nan_to_num(S[t]), np.nan_to_num(Pinv[t])
            r64 = orc.pcg(Sz.astype(np.float64), Pz.astype(np.float64), g[t].astype(np.float64), lam0[t].astype(np.float64), N, K, 0.0, pc)
            band = fp32_band(orc, Sz, Pz, g[t], lam0[t], N, K, pc, r64["lam"], trials=2)
            assert relinf(lam.cpu().numpy()[t], r64["lam"]) <= max(2e-5 if K == 2 else 1e-3, 4 * band)


def test_full_batch_in_several_launches_is_deterministic_and_composition_independent(orc, kern):
    """N = 256, 300 trajectories: 128 clusters fit the chip, so the call is three launches (128 + 128 + 44), each followed by its
    fix-up launch.  Same answer twice, the same answer for a sub-batch, tolerance exits and counts like the CPU restatement."""
    from mpcgpu_amd import PcgSolver, pcg_config
    N, B = 256, 300
    k = synth.make_kkt(N, B, 77)
    S, Pinv, g = synth.form_schur(k, poison_unused=True)
    dS, dP, dg = dev(S), dev(Pinv), dev(g)
    sol = PcgSolver(N, max_batch=B)
    sol.set_option("cluster_lpk", kern[0])
    cfg = pcg_config(pcg_exit_tol=1e-4, pcg_max_iter=60)
    runs = []
    for _ in range(2):
        lam = torch.zeros(B, n * N, device="cuda")
        it, ex = sol.solve(dS, dP, dg, lam, cfg, "ss")
        torch.cuda.synchronize()
        runs.append((lam.cpu().numpy(), it.cpu().numpy(), ex.cpu().numpy()))
    assert sol.get_option("last_kernel_family") == kern[1] and sol.get_option("last_kernel_cluster") == 2
    np.testing.assert_array_equal(runs[0][0], runs[1][0])
    np.testing.assert_array_equal(runs[0][1], runs[1][1])
    assert (runs[0][2] <= 1).all() and (runs[0][1] <= 60).all()          # no cluster gave up (flag 2 / 0xFFFFFFFF)
    sub = [0, 127, 128, 255, 256, 299]
    lam_s = torch.zeros(len(sub), n * N, device="cuda")
    it_s, _ = sol.solve(dev(S[sub]), dev(Pinv[sub]), dev(g[sub]), lam_s, cfg, "ss")
    torch.cuda.synchronize()
    np.testing.assert_array_equal(lam_s.cpu().numpy(), runs[0][0][sub])
    np.testing.assert_array_equal(it_s.cpu().numpy(), runs[0][1][sub])
    for t in (0, 128, 299):
        Sz, Pz = np.nan_to_num(S[t]), np.nan_to_num(Pinv[t])
        r32 = orc.pcg(Sz, Pz, g[t], np.zeros(n * N, np.float32), N, 60, 1e-4, "ss")
        assert abs(int(runs[0][1][t]) - int(r32["iters"])) <= max(3, 0.12 * r32["iters"]), (t, runs[0][1][t], r32["iters"])
        assert rel_residual(S[t], g[t], runs[0][0][t], N) <= 2 * rel_residual(S[t], g[t], r32["lam"], N) + 1e-6


def test_warm_start_and_zero_iterations(kern):
    """lambda in/out: a second call starting from the converged lambda exits at once with 0 iterations and leaves lambda alone."""
    from mpcgpu_amd import PcgSolver, pcg_config
    N, B = 384, 3
    k = synth.make_kkt(N, B, 5)
    S, Pinv, g = synth.form_schur(k)
    sol = PcgSolver(N, max_batch=B)
    sol.set_option("cluster_lpk", kern[0])
    dS, dP, dg = dev(S), dev(Pinv), dev(g)
    lam = torch.zeros(B, n * N, device="cuda")
    it, ex = sol.solve(dS, dP, dg, lam, pcg_config(pcg_exit_tol=1e-2, pcg_max_iter=2000), "ss")
    torch.cuda.synchronize()
    assert sol.get_option("last_kernel_family") == kern[1] and (ex.cpu().numpy() == 0).all() and (it.cpu().numpy() > 0).all()
    before = lam.clone()
    it2, ex2 = sol.solve(dS, dP, dg, lam, pcg_config(pcg_exit_tol=1e-2, pcg_max_iter=2000), "ss")
    torch.cuda.synchronize()
    assert (it2.cpu().numpy() == 0).all() and (ex2.cpu().numpy() == 0).all()
    assert torch.equal(lam, before)


def test_options_round_trip_and_selection(kern):
    """cluster_lpb: -1 auto (on) / 0 (row-triple cluster kernel) / 1; invalid values are refused and leave the handle usable."""
    from mpcgpu_amd import PcgSolver, pcg_config
    from mpcgpu_amd._lib import MpcgError
    N = 256
    k = synth.make_kkt(N, 1, 2)
    S, Pinv, g = synth.form_schur(k)
    sol = PcgSolver(N, max_batch=1)
    assert sol.get_option("cluster_lpk") == -1
    sol.set_option("cluster_lpk", kern[0])
    assert sol.get_option("cluster_lpb") == -1 and sol.get_option("cluster_fixup") == 1
    with pytest.raises(MpcgError):
        sol.set_option("cluster_lpb", 2)
    fam = {}
    for v in (-1, 0, 1):
        sol.set_option("cluster_lpb", v)
        assert sol.get_option("cluster_lpb") == v
        lam = torch.zeros(1, n * N, device="cuda")
        it, ex = sol.solve(dev(S), dev(Pinv), dev(g), lam, pcg_config(pcg_exit_tol=0.0, pcg_max_iter=7), "ss")
        torch.cuda.synchronize()
        fam[v] = (sol.get_option("last_kernel_family"), int(it.item()), lam.cpu().numpy())
    assert fam[-1][0] == kern[1] and fam[1][0] == kern[1] and fam[0][0] == 1 and all(f[1] == 7 for f in fam.values())
    np.testing.assert_array_equal(fam[-1][2], fam[1][2])
    # hand-offs through the XCD's L2 (default, when the members of a cluster share an XCD) or write-through: same arithmetic, same bits
    assert sol.get_option("cluster_l2") == 1
    sol.set_option("cluster_l2", 0)
    lam = torch.zeros(1, n * N, device="cuda")
    sol.solve(dev(S), dev(Pinv), dev(g), lam, pcg_config(pcg_exit_tol=0.0, pcg_max_iter=7), "ss")
    torch.cuda.synchronize()
    assert sol.get_option("last_kernel_family") == kern[1]
    np.testing.assert_array_equal(lam.cpu().numpy(), fam[1][2])
    assert relinf(fam[0][2][0], fam[1][2][0]) <= 2e-3          # two kernels, same PCG, 7 iterations: fp32 round-off of the products and inner products
