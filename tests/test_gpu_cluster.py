"""GPU: the cluster kernel — G workgroups on G CUs solve one trajectory together (long horizons, small
batches).  Same PCG, inner products summed per workgroup then across workgroups, so iterates match the
single-workgroup kernel / the float64 oracle within the fp32 band; flags, counts and lambda conventions
are identical; results are deterministic."""
import numpy as np
import pytest
import torch

from mpcgpu_amd import synth
from util import fp32_band, relinf

pytestmark = pytest.mark.gpu
n = 14


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


@pytest.mark.parametrize("N,G,B", [(128, 2, 3), (200, 3, 2), (256, 4, 2), (512, 8, 2), (512, 16, 1), (64, 2, 1),
                                   (512, -411, 40), (256, -406, 70)])      # G < 0: 4-wave members, -(400 + G), two per CU
@pytest.mark.parametrize("pc", ["ss", "jacobi"])
@pytest.mark.parametrize("lpbc", [1, 0])       # 1: clustered register-resident kernel (round 3: lane-pair, family 7), 0: row-triple cluster kernel (family 1)
def test_cluster_matches_oracle_and_single_workgroup(orc, N, G, B, pc, lpbc):
    from mpcgpu_amd import PcgSolver, pcg_config
    waves4 = G < 0
    if waves4:
        G = -G - 400
    k = synth.make_kkt(N, B, 7700 + N + G)
    S, Pinv, g = synth.form_schur(k, poison_unused=True)
    lam0 = np.random.default_rng(N).normal(0, 0.2, (B, n * N)).astype(np.float32)
    dS, dP, dg = dev(S), dev(Pinv), dev(g)
    out = {}
    for mode in ("cluster", "single"):
        sol = PcgSolver(N, max_batch=B)
        sol.set_option("cluster", G if mode == "cluster" else 0)
        sol.set_option("cluster_waves", 4 if waves4 else 8)
        sol.set_option("cluster_lpb", lpbc)
        for K, tol in ((3, 0.0), (30, 0.0), (400, 1e-3)):
            lam = dev(lam0)
            it, ex = sol.solve(dS, dP, dg, lam, pcg_config(pcg_exit_tol=tol, pcg_max_iter=K), pc)
            torch.cuda.synchronize()
            out[(mode, K)] = (lam.cpu().numpy(), it.cpu().numpy().astype(np.int64), ex.cpu().numpy())
            if mode == "cluster":                      # the kernel under test really ran
                # (the clustered lane-per-block kernel takes up to 8 members; beyond, the row-triple cluster kernel runs)
                assert sol.get_option("last_kernel_family") == (7 if lpbc and G <= 8 else 1) and sol.get_option("last_kernel_cluster") == G
                assert sol.get_option("last_kernel_waves") == (4 if waves4 else 8)
        if mode == "cluster":      # deterministic: bitwise identical on a second run
            lam = dev(lam0)
            sol.solve(dS, dP, dg, lam, pcg_config(pcg_exit_tol=1e-3, pcg_max_iter=400), pc)
            torch.cuda.synchronize()
            np.testing.assert_array_equal(lam.cpu().numpy(), out[("cluster", 400)][0])
    for K in (3, 30):
        c, s1 = out[("cluster", K)], out[("single", K)]
        assert (c[1] == K).all() and (c[2] == 1).all(), (c[1], c[2])          # no timeout (would be 0xFFFFFFFF / 2)
        for t in (range(B) if B <= 3 else (0, B // 2, B - 1)):
            r64 = orc.pcg(np.nan_to_num(S[t]).astype(np.float64), np.nan_to_num(Pinv[t]).astype(np.float64),
                          g[t].astype(np.float64), lam0[t].astype(np.float64), N, K, 0.0, pc)
            band = fp32_band(orc, S[t], Pinv[t], g[t], lam0[t], N, K, pc, r64["lam"], trials=2)
            tol = max(2e-5 if K == 3 else 1e-3, 4 * band)
            assert relinf(c[0][t], r64["lam"]) <= tol and relinf(c[0][t], s1[0][t]) <= tol
    c, s1 = out[("cluster", 400)], out[("single", 400)]
    assert (c[2] == s1[2]).all() or (np.abs(c[1] - s1[1]) > 0).any()
    assert (np.abs(c[1] - s1[1]) <= np.maximum(3, 0.12 * s1[1])).all(), (c[1], s1[1])


def test_cluster_falls_back_when_it_does_not_apply():
    """batch * G > #CUs (peers could not all be resident) or G > #triples: the single-workgroup kernel runs."""
    from mpcgpu_amd import PcgSolver, pcg_config
    N, B = 32, 200
    k = synth.make_kkt(N, B, 3)
    S, Pinv, g = synth.form_schur(k)
    sol = PcgSolver(N, max_batch=B)
    sol.set_option("cluster", 4)                       # 200 * 4 > 256 CUs
    lam = torch.zeros(B, n * N, device="cuda")
    it, ex = sol.solve(dev(S), dev(Pinv), dev(g), lam, pcg_config(pcg_exit_tol=0.0, pcg_max_iter=5))
    torch.cuda.synchronize()
    assert (it.cpu().numpy() == 5).all() and np.isfinite(lam.cpu().numpy()).all()
