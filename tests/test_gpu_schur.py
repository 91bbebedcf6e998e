"""GPU: the steps either side of the PCG through the C ABI:
mpcg_form_schur == the oracle's knot-by-knot restatement of include/pcg/linsys_setup.cuh BIT FOR BIT
(same operation order, contraction off on both sides), mpcg_compute_dz likewise vs include/common/dz.cuh;
plus the whole chain KKT blocks -> Schur -> PCG -> dz against a float64 KKT solve."""
import numpy as np
import pytest
import torch

from mpcgpu_amd import synth
from util import relinf

pytestmark = pytest.mark.gpu
n, m = 14, 7


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


@pytest.mark.parametrize("N", [2, 3, 8, 33, 128])
@pytest.mark.parametrize("precond", ["ss", "jacobi"])
@pytest.mark.parametrize("formation", ["auto", "walk16", "walk5", "walk1", "lds"])
def test_form_schur_bit_exact_vs_oracle(orc, N, precond, formation):
    """The register-resident formation (schur_walk.hip.h: chunk-walking kernel + seam kernel) with L block rows per chunk — auto: what a
    call of this size gets by itself (L = 1 here: every row a seam), 16 (one chunk up to N = 17, eight at N = 128), 5 (ragged last chunk, many
    seams), 1 — and the LDS kernels of schur_kernels.hip.h ("schur_dpp" = 0).  All the oracle's bits."""
    from mpcgpu_amd import PcgSolver
    B = 5                                             # (a wavefront of four chunks straddles trajectories)
    k = synth.make_kkt(N, B, 555 + N)
    G, C, g, c = synth.pack_kkt_dense(k, np.float32)
    sol = PcgSolver(N, max_batch=B)
    if formation.startswith("walk"):
        sol.set_option("schur_chunk", int(formation[4:]))
    elif formation == "lds":
        sol.set_option("schur_dpp", 0)
    dG = dev(G)
    poison = float("nan")
    S = torch.full((B, 3 * n * n * N), poison, device="cuda")
    P = torch.full((B, 3 * n * n * N), poison, device="cuda")
    gam = torch.full((B, n * N), poison, device="cuda")
    sol.form_schur(dG, dev(C), dev(g), dev(c), 1e-3, precond, S=S, Pinv=P, gamma=gam)
    torch.cuda.synchronize()
    assert sol.get_option("last_schur_chunk") == {"auto": 1, "lds": 0}.get(formation, int(formation[4:]) if formation.startswith("walk") else -1)
    S, P, gam, Ginv = S.cpu().numpy(), P.cpu().numpy(), gam.cpu().numpy(), dG.cpu().numpy()
    for b in range(B):
        So, Po, go, Go = orc.form_schur(G[b], C[b], g[b], c[b], N, np.float32(1e-3), ss=(precond == "ss"))
        # identical bits, including WHICH slots are left unwritten (NaN poison on both sides)
        np.testing.assert_array_equal(S[b], So)
        np.testing.assert_array_equal(P[b], Po)
        np.testing.assert_array_equal(gam[b], go)
        np.testing.assert_array_equal(Ginv[b], Go)
    # and close to the float64 numpy builder
    Sn, Pn, gn = synth.form_schur(k, precond=precond, dtype=np.float64)
    mS = ~np.isnan(S)
    assert relinf(S[mS], Sn[mS]) < 2e-3 and relinf(gam, gn) < 2e-3
    mP = ~np.isnan(P)
    assert relinf(P[mP], Pn[mP]) < 2e-3


@pytest.mark.parametrize("chunk", [0, 4])
def test_form_schur_without_preconditioner_leaves_pinv_alone(chunk):
    """precond = MPCG_PRECOND_NONE (what mpcg_block_solve needs): S, gamma and G^-1 as with a preconditioner, d_Pinv untouched."""
    from mpcgpu_amd import PcgSolver
    N, B = 9, 5
    k = synth.make_kkt(N, B, 31)
    G, C, g, c = synth.pack_kkt_dense(k, np.float32)
    sol = PcgSolver(N, max_batch=B)
    sol.set_option("schur_chunk", chunk)
    res = {}
    for pc in ("jacobi", "none"):
        dG = dev(G)
        S = torch.full((B, 3 * n * n * N), float("nan"), device="cuda")
        P = torch.full((B, 3 * n * n * N), float("nan"), device="cuda")
        gam = torch.full((B, n * N), float("nan"), device="cuda")
        sol.form_schur(dG, dev(C), dev(g), dev(c), 1e-3, pc, S=S, Pinv=P, gamma=gam)
        torch.cuda.synchronize()
        res[pc] = [t.cpu().numpy() for t in (S, gam, dG, P)]
    for a0, a1 in zip(res["jacobi"][:3], res["none"][:3]):
        np.testing.assert_array_equal(a0, a1)
    assert np.isnan(res["none"][3]).all() and not np.isnan(res["jacobi"][3]).all()


def test_form_schur_formations_agree_at_throughput_sized_batches():
    """The automatic chunk length of calls large enough to be throughput-bound (L = 4 at 700 x 64 knots, where the grid-stride loop wraps
    and the last wavefront is ragged; the handle's seam buffer is the one sized at its first, smaller call) against one row per chunk and
    against the LDS kernels: the same bits in S, Pinv, gamma and G^-1."""
    from mpcgpu_amd import PcgSolver
    N, B = 64, 700
    k = synth.make_kkt(N, 50, 4242)
    G, C, g, c = (np.tile(a, (B // 50, 1)) for a in synth.pack_kkt_dense(k, np.float32))
    sol = PcgSolver(N, max_batch=B)
    outs, chunks = [], []
    for mode in ("small", "auto", "one", "lds"):
        sol.set_option("schur_dpp", 0 if mode == "lds" else 1)
        sol.set_option("schur_chunk", 1 if mode == "one" else 0)
        b = 96 if mode == "small" else B
        dG = dev(G[:b])
        S = torch.full((b, 3 * n * n * N), float("nan"), device="cuda")
        P = torch.full((b, 3 * n * n * N), float("nan"), device="cuda")
        gam = torch.full((b, n * N), float("nan"), device="cuda")
        sol.form_schur(dG, dev(C[:b]), dev(g[:b]), dev(c[:b]), 1e-3, "ss", S=S, Pinv=P, gamma=gam)
        torch.cuda.synchronize()
        chunks.append(sol.get_option("last_schur_chunk"))
        outs.append([t.cpu().numpy() for t in (S, P, gam, dG)])
    assert chunks[1] > 1 and chunks[2] == 1 and chunks[3] == 0, chunks
    for other in (outs[2], outs[3]):
        for a0, a1 in zip(outs[1], other):
            np.testing.assert_array_equal(a0, a1)
    for a0, a1 in zip(outs[0], outs[1]):
        np.testing.assert_array_equal(a0, a1[:96])


@pytest.mark.parametrize("dz_dpp", [1, 0])
@pytest.mark.parametrize("N", [2, 3, 9, 128])
def test_compute_dz_bit_exact_vs_oracle(orc, N, dz_dpp):
    """dz_dpp = 1: the four-knots-per-wavefront kernel (round 4, default), 0: the one-workgroup-per-knot LDS kernel."""
    from mpcgpu_amd import PcgSolver
    B = 3
    k = synth.make_kkt(N, B, 77 + N)
    G, C, g, c = synth.pack_kkt_dense(k, np.float32)
    lam = np.random.default_rng(N).normal(size=(B, n * N)).astype(np.float32)
    sol = PcgSolver(N, max_batch=B)
    sol.set_option("dz_dpp", dz_dpp)
    dG = dev(G)
    sol.form_schur(dG, dev(C), dev(g), dev(c), 1e-3)
    dz = sol.compute_dz(dG, dev(C), dev(g), dev(lam))
    torch.cuda.synchronize()
    dz = dz.cpu().numpy()
    Ginv = dG.cpu().numpy()
    for b in range(B):
        np.testing.assert_array_equal(dz[b], orc.compute_dz(Ginv[b], C[b], g[b], lam[b], N))


def test_kkt_to_step_pipeline_vs_float64_kkt_solve(orc):
    """KKT blocks -> mpcg_form_schur -> mpcg_pcg_solve -> mpcg_compute_dz, all on the GPU, against the
    primal step of a dense float64 solve of the regularised KKT system  [G C^T; C 0][dz; lam] = [g; c]
    (sign conventions of include/common/dz.cuh / linsys_setup.cuh: dz = G^-1 (g - C^T lam))."""
    from mpcgpu_amd import PcgSolver, pcg_config
    N, B, rho = 16, 2, 1e-1
    k = synth.make_kkt(N, B, 31337)
    G, C, g, c = synth.pack_kkt_dense(k, np.float32)
    sol = PcgSolver(N, max_batch=B)
    dG, dC, dg, dc = dev(G), dev(C), dev(g), dev(c)
    S, P, gam = sol.form_schur(dG, dC, dg, dc, rho, "ss")
    lam = torch.zeros(B, n * N, device="cuda")
    it, ex = sol.solve(S, P, gam, lam, pcg_config(pcg_exit_tol=1e-9, pcg_max_iter=2000), "ss")
    dz = sol.compute_dz(dG, dC, dg, lam)
    torch.cuda.synchronize()
    assert (ex.cpu().numpy() == 0).all()
    dz, lam = dz.cpu().numpy(), lam.cpu().numpy()
    nz = (n + m) * N - m
    for b in range(B):
        Cm, Gm, gz = np.zeros((n * N, nz)), np.zeros((nz, nz)), np.zeros(nz)
        Cm[:n, :n] = np.eye(n)
        for kk in range(N):
            o = kk * (n + m)
            Gm[o:o + n, o:o + n] = k.Q[b, kk] + rho * np.eye(n)
            gz[o:o + n] = k.q[b, kk]
            if kk < N - 1:
                Gm[o + n:o + n + m, o + n:o + n + m] = k.R[b, kk] + rho * np.eye(m)
                gz[o + n:o + n + m] = k.r[b, kk]
            if kk > 0:
                po = (kk - 1) * (n + m)
                Cm[kk * n:(kk + 1) * n, po:po + n] = -k.A[b, kk - 1]
                Cm[kk * n:(kk + 1) * n, po + n:po + n + m] = -k.Bm[b, kk - 1]
                Cm[kk * n:(kk + 1) * n, o:o + n] = np.eye(n)
        K = np.block([[Gm, Cm.T], [Cm, np.zeros((n * N, n * N))]])
        sol64 = np.linalg.solve(K, np.concatenate([gz, k.c[b].reshape(-1)]))
        assert relinf(lam[b], sol64[nz:]) < 5e-3
        assert relinf(dz[b], sol64[:nz]) < 5e-3
        # the step satisfies the linearised constraints C dz = c
        assert np.abs(Cm @ dz[b].astype(np.float64) - k.c[b].reshape(-1)).max() < 5e-3


@pytest.mark.parametrize("N", [2, 5, 32, 128])
def test_csr_emitter_matches_reference_pattern_and_oracle(orc, N):
    """QDLDL twin (include/utils/csr.cuh, include/qdldl/linsys_setup.cuh): pattern and values emitted on the
    GPU equal the oracle's restatement bit for bit, and the oracle's QDLDL-style LDL^T solves the system
    from them (config 1 of BASELINE.json, with the GPU as producer)."""
    from mpcgpu_amd import PcgSolver
    B = 3
    k = synth.make_kkt(N, B, 4000 + N)
    G, C, g, c = synth.pack_kkt_dense(k, np.float32)
    sol = PcgSolver(N, max_batch=B)
    S, P, gam = sol.form_schur(dev(G), dev(C), dev(g), dev(c), 1e-3)
    col_ptr, row_ind = sol.prep_csr()
    val = sol.bd_to_csr_lowertri(S)
    torch.cuda.synchronize()
    Ap, Ai = orc.prep_csr(N)
    np.testing.assert_array_equal(col_ptr.cpu().numpy(), Ap)
    np.testing.assert_array_equal(row_ind.cpu().numpy(), Ai)
    Sh, vh, gh = S.cpu().numpy(), val.cpu().numpy(), gam.cpu().numpy()
    assert vh.shape[1] == sol.csr_nnz() == len(Ai)
    L = orc.LdlSolver(N, np.float32)
    for b in range(B):
        np.testing.assert_array_equal(vh[b], orc.bd_to_csr_lowertri(np.nan_to_num(Sh[b]), N))
        x = L.solve(vh[b], gh[b])
        assert relinf(x, orc.direct_solve(Sh[b], gh[b], N)) < 5e-2      # float LDL^T at cond ~1e5


def test_cpp_sqp_linsys_chain_over_shim_headers():
    """examples/sqp_linsys_chain.cpp: form_schur_system -> PCG launch -> compute_dz with the reference's own
    names/signatures (include/pcg/sqp.cuh:207-259) over include/mpcgpu_compat + include/gbd_pcg_compat; the
    program verifies the KKT conditions of the resulting step on the CPU."""
    import json
    import os
    import subprocess
    from mpcgpu_amd import build
    exe = build.CHAIN_BIN if os.path.exists(build.CHAIN_BIN) else build.build_chain_example()
    for args in ([], ["--direct"]):                 # pcg<> launch, then the GPU twin of the QDLDL path
        r = subprocess.run([exe] + args, capture_output=True, text=True, timeout=120)
        assert r.returncode == 0, r.stdout + r.stderr
        out = json.loads(r.stdout.strip().splitlines()[-1])
        assert out["pcg_exit"] == 0 and out["constraint_err"] < 1e-3 and out["stationarity_err"] < 1e-3
    # the same source with -DUSE_DOUBLES (linsys_t = double): every step in double precision, KKT conditions to 1e-9
    exe64 = build.build_chain_example_f64()
    r = subprocess.run([exe64], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stdout + r.stderr
    out = json.loads(r.stdout.strip().splitlines()[-1])
    assert out["pcg_exit"] == 0 and out["constraint_err"] < 1e-9 and out["stationarity_err"] < 1e-9, out


@pytest.mark.parametrize("N", [2, 3, 9, 32, 128, 300])
def test_block_solve_bit_exact_vs_oracle_and_close_to_float64(orc, N):
    """mpcg_block_solve (the GPU counterpart of the reference's QDLDL path, include/qdldl/sqp.cuh:22-49): same bits as
    the oracle's restatement of the block LU sweep, for a batch that is not a multiple of the four trajectories
    a wavefront carries; against the float64 direct solve it is as good as fp32 elimination gets at cond ~1e5."""
    from mpcgpu_amd import PcgSolver
    B = 7
    k = synth.make_kkt(N, B, 6100 + N)
    S, Pinv, g = synth.form_schur(k, poison_unused=True)          # NaN in the two never-written blocks
    sol = PcgSolver(N, max_batch=B)
    sol.set_option("block_solve_wide", 0)            # four trajectories per wavefront
    lam = sol.block_solve(dev(S), dev(g))
    sol.set_option("block_solve_wide", 1)            # one trajectory per wavefront, columns dealt over the DPP rows
    lam_w = sol.block_solve(dev(S), dev(g))
    torch.cuda.synchronize()
    lam, lam_w = lam.cpu().numpy(), lam_w.cpu().numpy()
    assert np.isfinite(lam).all()
    np.testing.assert_array_equal(lam_w, lam)
    for b in range(B):
        np.testing.assert_array_equal(lam[b], orc.block_solve(S[b], g[b], N))
        x64 = orc.direct_solve(S[b], g[b], N)
        assert relinf(lam[b], x64) < 5e-3
        r = np.linalg.norm(g[b] - orc.bt_spmv(np.nan_to_num(S[b]).astype(np.float64), lam[b].astype(np.float64), N)) / np.linalg.norm(g[b])
        assert r < 5e-3


def test_block_solve_well_conditioned_is_accurate_and_agrees_with_pcg(orc):
    """With rho = 1 (cond ~1e2) both GPU solvers reach fp32 accuracy and agree with each other."""
    from mpcgpu_amd import PcgSolver, pcg_config
    N, B = 64, 5
    k = synth.make_kkt(N, B, 4321)
    S, Pinv, g = synth.form_schur(k, rho=1.0)
    sol = PcgSolver(N, max_batch=B)
    dS, dP, dg = dev(S), dev(Pinv), dev(g)
    lam_d = sol.block_solve(dS, dg)
    lam_p = torch.zeros(B, n * N, device="cuda")
    it, ex = sol.solve(dS, dP, dg, lam_p, pcg_config(pcg_exit_tol=1e-12, pcg_max_iter=2000))
    torch.cuda.synchronize()
    assert (ex.cpu().numpy() == 0).all()
    lam_d, lam_p = lam_d.cpu().numpy(), lam_p.cpu().numpy()
    for b in range(B):
        x64 = orc.direct_solve(S[b], g[b], N)
        assert relinf(lam_d[b], x64) < 2e-4 and relinf(lam_p[b], x64) < 2e-4


def test_form_schur_without_preconditioner_for_the_direct_solver(orc):
    """precond='none': the same S, gamma and G^-1 bit for bit, d_Pinv never touched — what mpcg_block_solve needs."""
    from mpcgpu_amd import PcgSolver
    N, B = 33, 5
    k = synth.make_kkt(N, B, 909)
    G, C, g, c = synth.pack_kkt_dense(k, np.float32)
    sol = PcgSolver(N, max_batch=B)
    for dpp in (1, 0):
        sol.set_option("schur_dpp", dpp)
        dG = dev(G)
        S = torch.full((B, 3 * n * n * N), float("nan"), device="cuda")
        P = torch.full((B, 3 * n * n * N), 123.0, device="cuda")
        gam = torch.empty(B, n * N, device="cuda")
        sol.form_schur(dG, dev(C), dev(g), dev(c), 1e-3, "none", S=S, Pinv=P, gamma=gam)
        lam = sol.block_solve(S, gam)
        torch.cuda.synchronize()
        assert (P.cpu().numpy() == 123.0).all()
        for b in range(B):
            So, Po, go, Go = orc.form_schur(G[b], C[b], g[b], c[b], N, np.float32(1e-3), ss=False)
            np.testing.assert_array_equal(S[b].cpu().numpy(), So)
            np.testing.assert_array_equal(gam[b].cpu().numpy(), go)
            np.testing.assert_array_equal(dG[b].cpu().numpy(), Go)
            np.testing.assert_array_equal(lam[b].cpu().numpy(), orc.block_solve(So, go, N))


def test_bd_layout_device_templates_vs_oracle(orc):
    """include/gbd_pcg_compat/utils.cuh: store_block_bd / load_block_bd / gato_memcpy instantiated in a kernel
    (examples/bd_utils_probe.cpp) the way the reference's Schur formation calls them, against the oracle's
    orc_store_block_bd / orc_load_block_bd — bit for bit, never-written slots untouched."""
    import ctypes as C
    import json
    import os
    import subprocess
    from mpcgpu_amd import build
    exe = build.UTILS_BIN if os.path.exists(build.UTILS_BIN) else build.build_utils_probe()
    r = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stdout + r.stderr
    out = json.loads(r.stdout)
    n_, N = out["n"], out["N"]
    nn = n_ * n_
    src = np.array(out["src"], np.float32)
    bd = np.array([np.nan if v is None else v for v in out["bd"]], np.float32)
    loaded = np.array(out["loaded"], np.float32)
    lib = orc.lib()
    fp = lambda a: a.ctypes.data_as(C.POINTER(C.c_float))
    want = np.full(3 * nn * N, np.nan, np.float32)
    for k in range(N):
        for col in range(3):
            if (k == 0 and col == 0) or (k == N - 1 and col == 2):
                continue
            blk = np.ascontiguousarray(src[(k * 3 + col) * nn:(k * 3 + col + 1) * nn])
            lib.orc_store_block_bd_f32(n_, N, fp(blk), fp(want), col, k, C.c_float(-1.0 if col == 1 else 1.0))
    np.testing.assert_array_equal(bd, want)                              # NaN == NaN positions included
    for k in range(N):
        got = loaded[2 * k * nn:(2 * k + 1) * nn]
        exp = np.zeros(nn, np.float32)
        lib.orc_load_block_bd_f32(n_, N, fp(want), fp(exp), 1, k, 0)
        np.testing.assert_array_equal(got, exp)
        if k > 0:
            got = loaded[(2 * k + 1) * nn:(2 * k + 2) * nn]
            lib.orc_load_block_bd_f32(n_, N, fp(want), fp(exp), 2, k - 1, 1)
            np.testing.assert_array_equal(got, exp)
    np.testing.assert_array_equal(np.array(out["copied"], np.float32).reshape(N, n_), src.reshape(N, 3 * nn)[:, :n_])


def test_mpcsim_entry_points_with_both_linsys_solvers():
    """simulateMPC -> sqpSolvePcg | sqpSolveQdldl with the reference's names, signatures and return tuples
    (include/mpcsim.cuh:147, include/pcg/sqp.cuh:22, include/qdldl/sqp.cuh:53) over this repo's shim headers, the same
    source built with -DLINSYS_SOLVE=1 and =0 (the reference's compile-time switch, include/mpcsim.cuh:21-25).  The program
    drives an LQ tracking problem through four control steps and checks the KKT conditions of every iterate itself."""
    import json
    import os
    import subprocess
    from mpcgpu_amd import build
    bins = build.DEMO_BINS if all(os.path.exists(p) for p in build.DEMO_BINS.values()) else build.build_mpcsim_demo()
    outs = {}
    for sel, exe in bins.items():
        r = subprocess.run([exe], capture_output=True, text=True, timeout=180)
        assert r.returncode == 0, (sel, r.stdout + r.stderr)
        outs[sel] = json.loads(r.stdout.strip().splitlines()[-1])
        assert outs[sel]["linsys_solve"] == sel and outs[sel]["control_steps"] == 4 and outs[sel]["linsolves"] == 4
        assert outs[sel]["dynamics_defect"] < 1e-3 and outs[sel]["stationarity_rel"] < 2e-2
        # the SQP time box (sqpTimecheck, reference include/pcg/sqp.cuh:161-169): off / already expired / generous
        tb = outs[sel]["time_box"]
        assert tb["ok"] is True and tb["off_iters"] == 5 and tb["generous_iters"] == 5
        assert tb["expired_iters"] == 0 and tb["expired_linsolves"] == 0 and tb["expired_sqp_time_exit"] == 1
    # the two solvers steer the same problem the same way
    assert abs(outs[0]["tracking_last"] - outs[1]["tracking_last"]) < 1e-2 * max(1.0, outs[1]["tracking_first"])


def test_qdldl_twin_on_device_buffers(orc):
    """LINSYS_SOLVE == 0 end to end through the C ABI: mpcg_form_schur (no preconditioner) -> mpcg_bd_to_csr_lowertri ->
    mpcg_qdldl_solve_schur (D2H, host LDL^T of the library, H2D) against the float64 ground truth and the GPU direct solver."""
    from mpcgpu_amd import PcgSolver, QdldlSolver
    N, B = 64, 2
    k = synth.make_kkt(N, B, 321)
    G, C, g, c = synth.pack_kkt_dense(k, np.float32)
    sol = PcgSolver(N, max_batch=B)
    S, _, gam = sol.form_schur(dev(G), dev(C), dev(g), dev(c), 1e-3, "none")
    val = sol.bd_to_csr_lowertri(S)
    q = QdldlSolver(N)
    lam_gpu = sol.block_solve(S, gam)
    torch.cuda.synchronize()
    for b in range(B):
        lam = torch.zeros(n * N, device="cuda")
        q.solve_schur(sol, val[b], gam[b], lam)
        exact = orc.direct_solve(S[b].cpu().numpy(), gam[b].cpu().numpy(), N)
        assert relinf(lam.cpu().numpy(), exact) < 5e-2                    # float LDL^T at cond ~1e5
        assert relinf(lam_gpu[b].cpu().numpy(), exact) < 5e-2                  # fp32 block elimination, same conditioning
        np.testing.assert_array_equal(lam.cpu().numpy(), q.solve_host(val[b].cpu().numpy(), gam[b].cpu().numpy()))


def test_mpcsim_on_a_real_trajectory_window_with_the_library_kkt_stage():
    """examples/mpcsim_iiwa_demo.cpp: simulateMPC -> sqpSolvePcg | sqpSolveQdldl over the shim headers on rows 0.. of the reference's
    0_0 trajectory, the KKT stage being the library's own mpcg_generate_kkt (mpcgpu_compat::use_mpcg_generate_kkt with
    mpcg_plant_create_iiwa14) and the step stage the reference's line search (eight step lengths, rho adaptation) on a constraint-violation
    merit.  The constraint violation of the perturbed start must come down and stay down over three control steps, with both linear-system solvers."""
    import json
    import os
    import subprocess
    from mpcgpu_amd import build
    bins = build.IIWA_DEMO_BINS if all(os.path.exists(p) for p in build.IIWA_DEMO_BINS.values()) else build.build_iiwa_demo()
    outs = {}
    for sel, exe in bins.items():
        r = subprocess.run([exe], capture_output=True, text=True, timeout=180)
        assert r.returncode == 0, (sel, r.stdout + r.stderr)
        o = outs[sel] = json.loads(r.stdout.strip().splitlines()[-1])
        assert o["ok"] is True and o["linsys_solve"] == sel and o["control_steps"] == 3 and o["linsolves"] == 12
        assert o["violation_start"] > 1e-2 and max(o["violation_after_step"]) < 0.6 * o["violation_start"]
        assert len(o["line_search_exponents"]) == 12 and sum(p >= 0 for p in o["line_search_exponents"]) >= 6
