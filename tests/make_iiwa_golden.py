#!/usr/bin/env python3
"""Runs in the BUILD container only (reads /root/reference): extracts the KUKA iiwa-14 model tables and a slice of the
reference's trajectory fixtures, and writes
  mpcgpu_amd/data/iiwa14_model.json   robot model DATA: spatial transforms X (constant entries + the entries that are
                                      coef*sin(q_j) / coef*cos(q_j)), spatial inertias I, homogeneous transforms Xhom, as tabulated in
                                      include/dynamics/iiwa/iiwa_eepos_grid.cuh (init_XImats :909-1640, load_update_XImats_helpers
                                      :1770-1845, load_update_XmatsHom_helpers :1857-1904)
  mpcgpu_amd/data/iiwa_traj_0_0.npz   first 400 rows of examples/trajfiles/0_0_traj.csv (x(14), u(7)) and 0_0_eepos.traj (6)
  tests/golden/iiwa_kkt_N32.npz       KKT blocks (G, C, g, c) of three N=32 windows produced by oracle/iiwa_ref.py:generate_kkt
                                      (float64 restatement of include/common/kkt.cuh:22-163), the Schur systems the oracle forms
                                      from them and float64 PCG statistics
and prints how PCG behaves on real IIWA systems (the numbers quoted in DESIGN.md §3.6)."""
import json
import os
import re
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
REF = "/root/reference"
GRID = os.path.join(REF, "include/dynamics/iiwa/iiwa_eepos_grid.cuh")


def extract_model():
    txt = open(GRID).read()
    const = np.zeros(728)
    sec = txt[txt.index("T* init_XImats()"):txt.index("robotModel<T>* init_robotModel()")]
    seen = 0
    for m_ in re.finditer(r"h_XImats\[(\d+)\] = static_cast<T>\(([^)]+)\);", sec):
        const[int(m_.group(1))] = float(m_.group(2))
        seen += 1
    assert seen == 728, seen

    def trig(signature, arr):
        start = txt.index(signature)
        body = txt[start:txt.index("__syncthreads();", txt.index("if(threadIdx.x == 0 && threadIdx.y == 0){", start))]
        out = []
        for m_ in re.finditer(re.escape(arr) + r"\[(\d+)\] = static_cast<T>\(([^;]+)\);", body):
            e = m_.group(2).strip()
            mm = re.fullmatch(r"(-?[0-9.]*)\*?s_temp\[(\d+)\]", e)
            assert mm, e
            cf = mm.group(1)
            cf = -1.0 if cf == "-" else (1.0 if cf == "" else float(cf))
            out.append([int(m_.group(1)), cf, int(mm.group(2))])       # value = cf * (sin(q_j) if j < 7 else cos(q_{j-7}))
        return out

    model = {
        "robot": "KUKA LBR iiwa 14 R820 (7 revolute joints about the local z axis, serial chain, base fixed)",
        "source": "numeric tables of the reference's GRiD-generated include/dynamics/iiwa/iiwa_eepos_grid.cuh (robot description data, not code)",
        "layout": "X, I: 7 blocks of 36 values, column-major 6x6 (Featherstone order [angular; linear]); Xhom: 7 blocks of 16, column-major 4x4 "
                  "(link frame -> parent frame); *_trig: [flat index, coefficient, j] -> value = coefficient * (sin(q_j) if j < 7 else cos(q_{j-7}))",
        "X_const": const[:252].tolist(), "I": const[252:504].tolist(), "Xhom_const": const[504:616].tolist(),
        "X_trig": trig("void load_update_XImats_helpers(T *s_XImats, const T *s_q, const robotModel<T> *d_robotModel, T *s_temp)", "s_XImats"),
        "Xhom_trig": trig("void load_update_XmatsHom_helpers(T *s_XmatsHom, const T *s_q, const robotModel<T> *d_robotModel, T *s_temp)", "s_XmatsHom"),
    }
    assert len(model["X_trig"]) == 50 and len(model["Xhom_trig"]) == 28, (len(model["X_trig"]), len(model["Xhom_trig"]))
    os.makedirs(os.path.join(ROOT, "mpcgpu_amd", "data"), exist_ok=True)
    json.dump(model, open(os.path.join(ROOT, "mpcgpu_amd", "data", "iiwa14_model.json"), "w"))
    return model


def main():
    extract_model()
    import iiwa_ref as iiwa
    from mpcgpu_amd import synth
    import oracle as orc
    orc.build()
    M = iiwa.Model()
    from mpcgpu_amd.iiwa import read_csv
    traj = read_csv(os.path.join(REF, "examples/trajfiles/0_0_traj.csv"))
    eep = read_csv(os.path.join(REF, "examples/trajfiles/0_0_eepos.traj"))
    assert traj.shape == (666, 21) and eep.shape == (666, 6)
    # sanity of the restated kinematics against the reference's own fixture: eepos.traj row t = end-effector position of traj row t
    err = max(np.abs(M.ee_pos(traj[t, :7]) - eep[t, :3]).max() for t in (0, 1, 50, 199, 400, 665))
    print("end-effector position vs 0_0_eepos.traj: max abs err", err)
    assert err < 2e-5
    np.savez_compressed(os.path.join(ROOT, "mpcgpu_amd", "data", "iiwa_traj_0_0.npz"), xu=traj[:400].astype(np.float32), eepos=eep[:400].astype(np.float32))

    n, m = 14, 7
    rng = np.random.default_rng(7)
    out = {}
    for N in (32, 128):
        print(f"---- N = {N} ----")
        scen = []
        for name, t0, goal_shift, pert in (("nominal+state noise", 0, 0, 0.02), ("window t=20, goals 8 steps ahead", 20, 8, 0.0),
                                           ("window t=60, goals 4 steps ahead + noise", 60, 4, 0.05)):
            xu = traj[t0:t0 + N].reshape(-1)[:(n + m) * N - m].copy()
            goals = eep[t0 + goal_shift:t0 + goal_shift + N]
            xs = xu[:n] + pert * rng.standard_normal(n)
            if pert:
                xu[:n] = xs                                   # the MPC loop starts the iterate at the measured state (c_0 = 0)
                xu = xu + pert * 0.3 * rng.standard_normal(xu.shape)
                xu[:n] = xs
            G, C, g, c = iiwa.generate_kkt(M, xu, goals, xs, N)
            G32, C32, g32, c32 = (a.astype(np.float32) for a in (G, C, g, c))
            S, P, gam, _ = orc.form_schur(G32.copy(), C32, g32, c32, N, synth.RHO_INIT, ss=True)
            S, P = np.nan_to_num(S), np.nan_to_num(P)
            Sd = synth.bd_to_dense(S.astype(np.float64), N)
            ev = np.linalg.eigvalsh(-Sd)
            cond = ev.max() / ev.min()
            stats = {}
            for pc in ("ss", "jacobi"):
                for tol in (1e-5, 1e-4, 1e-3):
                    r = orc.pcg(S, P, gam, np.zeros(n * N, np.float32), N, 5000, tol, pc)
                    res = float(np.linalg.norm(gam - Sd @ r["lam"].astype(np.float64)) / np.linalg.norm(gam))
                    stats[f"{pc}_tol{tol:g}"] = (r["iters"], res)
            cap = synth.pcg_max_iter(N)
            print(f"{name}: cond(-S) = {cond:.3e}, |gamma| = {np.linalg.norm(gam):.3e}, iteration cap {cap}; fp32 PCG from lambda0=0 (iters, true residual):")
            for k_, v in stats.items():
                print(f"     {k_}: {v[0]} iters, residual {v[1]:.2e}")
            scen.append(dict(name=name, xu=xu.astype(np.float32), goals=goals.astype(np.float32), xs=xs.astype(np.float32), G=G32, C=C32, g=g32, c=c32,
                             S=S.astype(np.float32), Pinv=P.astype(np.float32), gamma=gam.astype(np.float32), cond=cond,
                             iters_ss_1e4=stats["ss_tol0.0001"][0]))
        if N == 128:
            # two of the N = 128 windows (cond(-S) ~ 1e7: the regime bench.py's iiwa_run reports) with float64 answers made HERE by dense numpy
            # (tests/make_golden.py:dense_pcg — no code shared with oracle/ or the kernels): iterate after K iterations from a cold and from an
            # MPC-style warm start (the direct solution of a slightly different right-hand side), direct solution
            from make_golden import dense_pcg
            flat = {}
            for i, sc in enumerate(scen[:2]):
                S64, P64, g64 = (sc[k_].astype(np.float64) for k_ in ("S", "Pinv", "gamma"))
                Sd, Pd = synth.bd_to_dense(S64, N), synth.bd_to_dense(P64, N)
                lam_star = np.linalg.solve(Sd, g64)
                warm = np.linalg.solve(Sd, g64 * (1 + 0.02 * rng.standard_normal(g64.shape)))      # previous SQP iterate's multipliers
                for k_ in ("S", "Pinv", "gamma", "cond"):
                    flat[f"s{i}_{k_}"] = sc[k_]
                flat[f"s{i}_lam_direct"] = lam_star
                flat[f"s{i}_lam_warm0"] = warm.astype(np.float32)
                for K in (10, 40):
                    flat[f"s{i}_lam_cold_K{K}"] = dense_pcg(Sd, Pd, g64, np.zeros(n * N), K)[0]
                    flat[f"s{i}_lam_warm_K{K}"] = dense_pcg(Sd, Pd, g64, warm.astype(np.float32).astype(np.float64), K)[0]
            np.savez_compressed(os.path.join(ROOT, "tests", "golden", "iiwa_kkt_N128.npz"), **flat)
        if N == 32:
            flat = {}
            for i, sc in enumerate(scen):
                for k_, v in sc.items():
                    flat[f"s{i}_{k_}"] = v
            np.savez_compressed(os.path.join(ROOT, "tests", "golden", "iiwa_kkt_N32.npz"), **flat)


if __name__ == "__main__":
    main()
