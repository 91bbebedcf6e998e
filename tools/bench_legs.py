"""The side legs of bench.py (everything that is NOT the timed region): each takes the bench context `cx` (a SimpleNamespace made by
bench.py: sol, dev, N, B, args, cfg, max_iter, the device arrays of the timed workload) and returns a dict for the FULL record
(`gpurun_out/bench_full.json`).  bench.py picks a few scalars out of them for the one compact line the driver parses.

Every leg is wrapped by bench.py's `run_leg`: a failing leg is reported as {"error": ...} and never costs the headline line.
Only `cpu_baseline`, `parity_sample` and nothing else here touch oracle/ (the checker), after the timed region, as DESIGN.md §4 allows.
"""
from __future__ import annotations

import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from mpcgpu_amd import PcgSolver, pcg_config, synth  # noqa: E402

HBM_PEAK_GBS = 8000.0      # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s
FP32_VALU_PEAK_TF = 157.3  # same guide: 256 CUs x 4 SIMDs x 64 flop/clk x 2.4 GHz
FP64_VALU_PEAK_TF = 78.6
RAMP_S = 0.02              # every timed leg first runs its function back to back for this long (steady clocks)

# Algorithmic bytes and arithmetic of the producer kernels, per knot (n = 14, m = 7):
#   form_schur (ss): reads  G 245 + C 294 + g 21 + c 14 floats, writes S 3x196 + Pinv 3x196 + G^-1 245 + gamma 14  = 2,009 floats = 8,036 B
#   compute_dz:      reads  G^-1 245 + C 294 + g 21 + lambda 14 (+14 of the next knot, L2), writes dz 21             = 2,380 B
#   generate_kkt:    reads  x,u,x+ 35 + goals 12 floats, writes G 245 + C 294 + g 21 + c 14                         = 2,484 B
PRODUCER_MODEL = {
    "form_schur": {"bytes_per_unit": 8036, "flops_per_unit": 2 * (1323 + 5831 + 4 * 196), "dtype": "f32"},
    "compute_dz": {"bytes_per_unit": 2380, "flops_per_unit": 2 * (2 * 196 + 98 + 49), "dtype": "f32"},
    "generate_kkt": {"bytes_per_unit": 2484, "flops_per_unit": 25 * 2800 + 2000, "dtype": "f64"},
}


def timed(fn, reps, warm=1):
    """Median HIP-event time (ms) of `fn` over `reps` runs on the current stream (= the stream the library launches on), after RAMP_S of
    back-to-back runs: a few sub-millisecond launches after a host-side pause are otherwise timed below the steady clocks."""
    t_end = time.perf_counter() + RAMP_S
    while time.perf_counter() < t_end:
        fn()
        torch.cuda.synchronize()
    ts = []
    for i in range(reps + warm):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        fn()
        e1.record()
        torch.cuda.synchronize()
        if i >= warm:
            ts.append(e0.elapsed_time(e1))
    return float(np.median(ts))


def build_inputs(sol, N, batch, seed0, precond, dev, chunk=128, rho=synth.RHO_INIT):
    """Synthetic IIWA-shaped KKT blocks (host, numpy) -> Schur systems on the GPU with the library's own mpcg_form_schur (the reference's
    form_schur_system step).  Returns device tensors (S, Pinv, gamma); input generation is outside every timed region."""
    S = torch.empty(batch, 3 * 196 * N, device=dev)
    P = torch.empty_like(S)
    g = torch.empty(batch, 14 * N, device=dev)
    for lo in range(0, batch, chunk):
        hi = min(batch, lo + chunk)
        k = synth.make_kkt(N, hi - lo, 900000 + seed0 + lo)          # trajectory b of make_kkt(seed) depends only on (seed, b)
        Gd, Cd, gd, cd = (torch.from_numpy(a).to(dev) for a in synth.pack_kkt_dense(k, np.float32))
        sol.form_schur(Gd, Cd, gd, cd, rho, precond, S=S[lo:hi], Pinv=P[lo:hi], gamma=g[lo:hi])
    torch.cuda.synchronize()
    return S, P, g


def cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def _oracle():
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import oracle as orc
    orc.build()
    return orc


class SclkSampler:
    """Shader clock and socket power while a leg runs: `rocm-smi --showclocks --showpower` from a thread."""

    def __init__(self, card_index=0):
        import shutil
        self.exe = shutil.which("rocm-smi") or ("/opt/rocm/bin/rocm-smi" if os.path.exists("/opt/rocm/bin/rocm-smi") else None)
        self.card = card_index
        self.sclk, self.power, self._stop, self._thr = [], [], False, None

    def _run(self):
        import re
        import subprocess
        while not self._stop:
            try:
                txt = subprocess.run([self.exe, "-d", str(self.card), "--showclocks", "--showpower"], capture_output=True, text=True, timeout=10).stdout
            except Exception:
                return
            for ln in txt.splitlines():
                if "sclk" in ln:
                    m = re.search(r"\((\d+)\s*Mhz\)", ln, re.I)
                    if m:
                        self.sclk.append(int(m.group(1)))
                elif "Power" in ln and "(W)" in ln:
                    m = re.search(r"([0-9.]+)\s*$", ln.strip())
                    if m:
                        self.power.append(float(m.group(1)))
            time.sleep(0.2)

    def __enter__(self):
        if self.exe:
            import threading
            self._thr = threading.Thread(target=self._run, daemon=True)
            self._thr.start()
        return self

    def __exit__(self, *a):
        self._stop = True
        if self._thr:
            self._thr.join(timeout=12.0)

    def summary(self):
        if not self.sclk:
            return None
        return {"min": int(min(self.sclk)), "median": int(np.median(self.sclk)), "max": int(max(self.sclk)), "samples": len(self.sclk),
                "socket_power_w_median": float(np.median(self.power)) if self.power else None}


# --------------------------------------------------------------------------------------------------------------------------------
# traffic: committed PMC passes (default) or opt-in child passes of this run
# --------------------------------------------------------------------------------------------------------------------------------

def load_traffic(kernel_key):
    """HBM-side traffic of one launch from this round's committed PMC passes (profiles/traffic.json, written by tools/profile_round.sh from
    `rocprofv3 --kernel-trace --pmc FETCH_SIZE` / `WRITE_SIZE` runs of this same bench command)."""
    import json
    try:
        t = json.load(open(os.path.join(ROOT, "profiles", "traffic.json")))
        return t.get("kernels", {}).get(kernel_key), t.get("source")
    except (OSError, ValueError):
        return None, None


def inrun_pmc(argv_tail, timeout_s=120):
    """OPT-IN (--inrun-pmc): HBM-side traffic of THIS run's kernels by two rocprofv3 child runs of bench.py --profile-mini (one --pmc
    FETCH_SIZE pass, one --pmc WRITE_SIZE pass, counters only with --kernel-trace, as /opt/skills/guides/MI355X_MICROARCH.md prescribes;
    FETCH_SIZE doubled: gfx950 tallies 128-byte read requests at 64 bytes).  Returns ({kernel name: {...}}, note) or (None, why not)."""
    import shutil
    import sqlite3
    import subprocess
    import tempfile
    exe = shutil.which("rocprofv3") or ("/opt/rocm/bin/rocprofv3" if os.path.exists("/opt/rocm/bin/rocprofv3") else None)
    if not exe:
        return None, "rocprofv3 not found on this box"
    if os.environ.get("MPCG_BENCH_CHILD") == "1":
        return None, "child run"
    kern = {}
    env = dict(os.environ, MPCG_BENCH_CHILD="1", TMPDIR="/tmp")
    t0 = time.perf_counter()
    for counter in ("FETCH_SIZE", "WRITE_SIZE"):
        d = tempfile.mkdtemp(prefix="mpcg_pmc_", dir="/tmp")
        cmd = [exe, "--kernel-trace", "--pmc", counter, "-d", d, "-o", "p", "--", sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "3", "--warmup", "1",
               "--profile-mini"] + argv_tail
        try:
            r = subprocess.run(cmd, cwd="/tmp", env=env, stdout=subprocess.DEVNULL, stderr=subprocess.PIPE, timeout=timeout_s)
        except (subprocess.TimeoutExpired, OSError) as e:
            shutil.rmtree(d, ignore_errors=True)
            return None, f"rocprofv3 {counter} pass failed: {e!r}"
        dbs = [os.path.join(dp, f) for dp, _, fs in os.walk(d) for f in fs if f.endswith(".db")]
        if r.returncode != 0 or not dbs:
            shutil.rmtree(d, ignore_errors=True)
            return None, f"rocprofv3 {counter} pass: rc {r.returncode}, {len(dbs)} result files: {r.stderr.decode(errors='replace')[-200:]}"
        try:
            c = sqlite3.connect(dbs[0])
            rows = list(c.execute("select e.name, d.grid_size_x, count(*), avg(e.counter_value) from pmc_events e "
                                  "left join rocpd_kernel_dispatch d on d.dispatch_id = e.dispatch_id "
                                  "where e.name like '%mpcg%' and e.counter_name = ? group by e.name, d.grid_size_x", (counter,)))
        except sqlite3.Error:
            try:
                rows = [(nm, None, n, avg) for nm, n, avg in c.execute(
                    "select name, count(*), avg(counter_value) from pmc_events where name like '%mpcg%' and counter_name = ? group by name", (counter,))]
            except sqlite3.Error as e:
                shutil.rmtree(d, ignore_errors=True)
                return None, f"rocprofv3 {counter} pass: cannot read {dbs[0]}: {e!r}"
        for nm, grid, n, avg in rows:
            k = kern.setdefault((nm.replace("void ", ""), grid), {"launches": n})
            k[counter] = avg * 1024.0
        shutil.rmtree(d, ignore_errors=True)
    out = {}
    for (nm, grid), k in kern.items():
        if "FETCH_SIZE" in k and "WRITE_SIZE" in k:
            e = {"kernel": nm, "grid": grid, "launches_averaged": k["launches"], "fetch_bytes_corrected": 2.0 * k["FETCH_SIZE"], "write_bytes": k["WRITE_SIZE"],
                 "hbm_traffic_bytes_per_launch": 2.0 * k["FETCH_SIZE"] + k["WRITE_SIZE"]}
            if nm not in out or (grid or 0) > (out[nm]["grid"] or 0):          # several launch shapes of one kernel: keep the largest grid
                out[nm] = e
    return out, f"measured in this run: 2 rocprofv3 --kernel-trace --pmc child passes of `bench.py --profile-mini` ({time.perf_counter() - t0:.0f} s)"


def find_traffic(pmc, needle):
    for nm, e in (pmc or {}).items():
        if needle in nm:
            return e
    return None


# --------------------------------------------------------------------------------------------------------------------------------
# legs
# --------------------------------------------------------------------------------------------------------------------------------

def spmv_roofline(cx):
    """HBM roofline kernel of the path: the stand-alone block-tridiagonal SpMV (north_star's >= 60 % target), S streamed from
    HBM (1.2 GB >> the 256 MiB Infinity Cache).  NOT in the timed region and not called by the PCG kernels, which keep S in registers."""
    sol, dev, N, B, args = cx.sol, cx.dev, cx.N, cx.B, cx.args
    Bs = max(B, args.spmv_batch)
    reps = (Bs + B - 1) // B
    S_big = cx.d_S.repeat(reps, 1)[:Bs].contiguous() if reps > 1 else cx.d_S
    x = torch.randn(Bs, 14 * N, device=dev)
    y = torch.empty_like(x)

    def spmv20():
        for _ in range(20):
            sol.bt_spmv(S_big, x, y)
    ms = timed(spmv20, 3, warm=1) / 20          # 20 back-to-back launches per timing, like the rocprofv3 average
    b_unit = synth.algorithmic_bytes(N)["spmv"]
    b_sp = b_unit * Bs
    sp = {"bound": "hbm", "kernel": "bt_spmv_kernel", "achieved": b_sp / (ms * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
          "frac": b_sp / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, "kernel_ms": ms, "traffic": None,
          "algorithmic_bytes_per_launch": b_sp, "bytes_per_unit": b_unit, "units_per_launch": Bs,
          "unit_of_work": "one block-tridiagonal SpMV of one trajectory (SURVEY §8d)", "working_set_mb": Bs * 3 * 196 * N * 4 / 1e6,
          "trajectory_spmv_per_sec": Bs / (ms * 1e-3)}
    try:        # what THIS box's HBM delivers to a pure read of the same array (the library's probe kernel: 16-byte nontemporal loads, nothing else)
        sink = torch.zeros(1, device=dev)

        def rd20():
            for _ in range(20):
                sol.probe_hbm_read(S_big, sink)
        ms_rd = timed(rd20, 3, warm=1) / 20
        sp["read_ceiling_gbs_this_run"] = S_big.numel() * 4 / (ms_rd * 1e-3) / 1e9
        sp["frac_of_read_ceiling_this_run"] = sp["achieved"] / sp["read_ceiling_gbs_this_run"]
    except Exception:
        sp["read_ceiling_gbs_this_run"] = None
    tr, src = load_traffic(f"bt_spmv_kernel|N{N}_B{Bs}")
    if tr:
        sp["traffic"] = tr["hbm_traffic_bytes_per_launch"]
        sp["traffic_source"] = src
        sp["traffic_over_algorithmic"] = sp["traffic"] / b_sp
    if args.spmv_mfma:
        sol.set_option("spmv_mfma", 1)           # config 5's MFMA block-GEMV experiment, same launch shape
        ms_mfma = timed(spmv20, 3, warm=1) / 20
        sol.set_option("spmv_mfma", 0)
        sp["mfma_experiment"] = {"kernel": "bt_spmv_mfma_kernel", "ms": ms_mfma, "achieved": b_sp / (ms_mfma * 1e-3) / 1e9, "unit": "GB/s",
                                 "frac": b_sp / (ms_mfma * 1e-3) / 1e9 / HBM_PEAK_GBS,
                                 "mfma_flops_issued_per_launch": 2 * 16 * 16 * 4 * 12 * Bs * N, "useful_flops_per_launch": 2 * (3 * N - 2) * 196 * Bs}
    return sp


def fp32_check(orc, S, P, g, lam_gpu, it_gpu, N, pc):
    """One sampled trajectory against the oracle: the float64 iterate after the SAME number of iterations, the band the CPU float32
    restatement reaches on the same inputs, and the true residuals."""
    S, P = np.nan_to_num(S), np.nan_to_num(P)
    z = np.zeros(14 * N)
    r64 = orc.pcg(S.astype(np.float64), P.astype(np.float64), g.astype(np.float64), z, N, it_gpu, 0.0, pc)["lam"]
    r32 = orc.pcg(S, P, g, z.astype(np.float32), N, it_gpu, 0.0, pc)["lam"]
    den = max(np.abs(r64).max(), 1e-300)
    err, band = float(np.abs(lam_gpu - r64).max() / den), float(np.abs(r32 - r64).max() / den)
    Sd = synth.bd_to_dense(S, N)
    res = lambda v: float(np.linalg.norm(g - Sd @ np.asarray(v, np.float64)) / np.linalg.norm(g))
    return {"iters": int(it_gpu), "rel_err_vs_f64_same_iters": err, "cpu_f32_band": band, "ok": bool(err <= max(1e-3, 4 * band)),
            "true_residual_gpu": res(lam_gpu), "true_residual_cpu_f32": res(r32)}


def parity_sample(cx):
    """Sampled trajectories of the workload that was just timed, against the CPU oracle (checker only, after the timed region)."""
    orc = _oracle()
    B, N = cx.B, cx.N
    lam_h = cx.d_lam.cpu().numpy()
    idx = sorted({0, B // 3, (2 * B) // 3, B - 1})
    checks = []
    for b_ in idx:
        c = fp32_check(orc, cx.d_S[b_].cpu().numpy(), cx.d_P[b_].cpu().numpy(), cx.d_g[b_].cpu().numpy(), lam_h[b_], int(cx.it_host[b_]), N, cx.args.precond)
        c["trajectory"] = int(b_)
        checks.append(c)
    return {"against": "oracle/ (CPU, float64 iterate after the same number of iterations; tolerance max(1e-3, 4 x CPU float32 band))",
            "checked": len(checks), "all_ok": bool(all(c["ok"] for c in checks)), "samples": checks}


def warm_start_synthetic(cx):
    """The synthetic workload warm-started: converged solution + noise, so solves leave the loop at different iterations."""
    sol, dev, N, B = cx.sol, cx.dev, cx.N, cx.B
    lam_star = torch.zeros(B, 14 * N, device=dev)
    sol.solve(cx.d_S, cx.d_P, cx.d_g, lam_star, pcg_config(pcg_exit_tol=1e-9, pcg_max_iter=4000), cx.args.precond)
    gen = torch.Generator(device=dev).manual_seed(1234 + cx.rank)
    scale = lam_star.abs().amax(dim=1, keepdim=True)
    amp = torch.logspace(-4, -1, B, device=dev)[torch.randperm(B, device=dev, generator=gen)].unsqueeze(1)
    lam_w = lam_star + amp * scale * torch.randn(B, 14 * N, device=dev, generator=gen)

    def warm_solve():
        cx.d_lam.copy_(lam_w)
        cx.run_solve()
    ms_w = timed(warm_solve, 3, warm=1) - timed(lambda: cx.d_lam.copy_(lam_w), 3, warm=1)
    itw = cx.d_it.cpu().numpy().astype(np.int64)
    return {"lambda0": "converged solution + gaussian noise of relative amplitude 1e-4..1e-1 (log-uniform over the batch)",
            "mean_pcg_iters": float(itw.mean()), "min_pcg_iters": int(itw.min()), "max_pcg_iters": int(itw.max()),
            "max_iter_exit_rate": float(cx.d_ex.float().mean().item()), "kernel_ms": ms_w,
            "pcg_iterations_per_sec": float(itw.sum() / (ms_w * 1e-3)), "linsolves_per_sec": B / (ms_w * 1e-3)}


def producers_mini(cx):
    """(--profile-mini, the rocprofv3 child: three launches of each producer kernel at the bench batch, nothing else)"""
    from mpcgpu_amd import Plant, iiwa
    sol, dev, N, B = cx.sol, cx.dev, cx.N, cx.B
    plant = Plant(device=dev.index)
    xu_h, goals_h, xs_h = iiwa.random_windows(N, B, 2024 + cx.rank)
    f32 = lambda a_: torch.from_numpy(np.ascontiguousarray(a_, np.float32)).to(dev)
    d_xu, d_goal, d_xs = f32(xu_h), f32(goals_h.reshape(B, -1)), f32(xs_h)
    for _ in range(3):
        Gk, Ck, gk, ck = sol.generate_kkt(plant, d_goal, d_xs, d_xu, iiwa.TIMESTEP, iiwa.QD_COST, iiwa.r_cost(N))
        sol.form_schur(Gk, Ck, gk, ck, synth.RHO_INIT, "ss")
        sol.compute_dz(Gk, Ck, gk, cx.d_lam)
    torch.cuda.synchronize()
    return {"launches": 3}


def mpc_window_sets(N, B, seed, steps, max_noise=0.05):
    """`steps` CONSECUTIVE control steps of `B` independent MPC loops on the reference trajectory: trajectory b at step s tracks the window
    that starts s rows after its step-0 window (the MPC loop shifts its horizon by one knot per control step, include/mpcsim.cuh:318-345),
    with the same goal lead and its own measurement noise.  So trajectory b's linear systems at consecutive steps are related the way a
    real loop's are, and the previous step's PCG iteration counts are a REALISTIC (not exact) prediction of this step's."""
    from mpcgpu_amd import iiwa
    d = np.load(iiwa.TRAJ_FIXTURE)
    traj, eep = d["xu"].astype(np.float64), d["eepos"].astype(np.float64)
    n, m = 14, 7
    rng = np.random.default_rng(seed)
    t0 = rng.integers(0, traj.shape[0] - N - 8 - steps, size=B)
    sh = rng.integers(0, 9, size=B)
    amp = max_noise * rng.random(B)
    sets = []
    for s in range(steps):
        xu = np.zeros((B, (n + m) * N - m))
        goals = np.zeros((B, N, 6))
        xs = np.zeros((B, n))
        for b in range(B):
            w = traj[t0[b] + s:t0[b] + s + N].reshape(-1)[:(n + m) * N - m].copy()
            xs[b] = w[:n] + amp[b] * rng.standard_normal(n)
            w += 0.3 * amp[b] * rng.standard_normal(w.shape)
            w[:n] = xs[b]
            xu[b], goals[b] = w, eep[t0[b] + s + sh[b]:t0[b] + s + sh[b] + N]
        sets.append((xu, goals, xs))
    return sets


def iiwa_run(cx):
    """The same path on REAL IIWA-14 systems, produced on the device like the reference's SQP iteration does (include/pcg/sqp.cuh:190-232):
    generate_kkt -> form_schur -> PCG (cold; warm-started the way the MPC loop warm-starts, include/mpcsim.cuh:186,267,337) -> dz; the
    producers' rooflines; the whole step as one hipGraph; the dispatch-order hint with an exact, a realistic and no prediction."""
    from mpcgpu_amd import Plant, iiwa
    sol, dev, N, B, cfg, args = cx.sol, cx.dev, cx.N, cx.B, cx.cfg, cx.args
    d_it, d_ex = cx.d_it, cx.d_ex
    plant = Plant(device=dev.index)
    f32 = lambda a_: torch.from_numpy(np.ascontiguousarray(a_, np.float32)).to(dev)
    STEPS = 4
    sets = mpc_window_sets(N, B, 2024 + cx.rank, STEPS)
    xu_h, goals_h, xs_h = sets[0]
    d_xu, d_goal, d_xs = f32(xu_h), f32(goals_h.reshape(B, -1)), f32(xs_h)
    rc = iiwa.r_cost(N)
    ms_kkt = timed(lambda: sol.generate_kkt(plant, d_goal, d_xs, d_xu, iiwa.TIMESTEP, iiwa.QD_COST, rc), 3, warm=1)
    # "kkt_f32" = 1 (opt-in): linsys_t's own arithmetic as in the reference's GRiD code, two knots per lane in packed float
    sol.set_option("kkt_f32", 1)
    ms_kkt_f32 = timed(lambda: sol.generate_kkt(plant, d_goal, d_xs, d_xu, iiwa.TIMESTEP, iiwa.QD_COST, rc), 3, warm=1)
    sol.set_option("kkt_f32", 0)
    Gk, Ck, gk, ck = sol.generate_kkt(plant, d_goal, d_xs, d_xu, iiwa.TIMESTEP, iiwa.QD_COST, rc)
    Gk0 = Gk.clone()
    ms_schur = timed(lambda: (Gk.copy_(Gk0), sol.form_schur(Gk, Ck, gk, ck, synth.RHO_INIT, "ss")), 3, warm=1) - timed(lambda: Gk.copy_(Gk0), 3, warm=1)
    Gk.copy_(Gk0)
    rS, rP, rg = sol.form_schur(Gk, Ck, gk, ck, synth.RHO_INIT, "ss")
    lam_any = torch.randn(B, 14 * N, device=dev)
    ms_dz = timed(lambda: sol.compute_dz(Gk, Ck, gk, lam_any), 5, warm=1)
    knots = B * N
    prod = {}
    for nm_, ms__ in (("generate_kkt", ms_kkt), ("form_schur", ms_schur), ("compute_dz", ms_dz)):
        mdl = PRODUCER_MODEL[nm_]
        gbs = knots * mdl["bytes_per_unit"] / (ms__ * 1e-3) / 1e9
        tfl = knots * mdl["flops_per_unit"] / (ms__ * 1e-3) / 1e12
        pk = FP64_VALU_PEAK_TF if mdl["dtype"] == "f64" else FP32_VALU_PEAK_TF
        prod[nm_] = {"frac": gbs / HBM_PEAK_GBS, "bound": "hbm", "kernel_ms": ms__, "achieved": gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "bytes_per_unit": mdl["bytes_per_unit"], "flops_per_unit": mdl["flops_per_unit"], "units_per_launch": knots, "unit_of_work": "one knot point of one trajectory",
                     "useful_tflops": tfl, "frac_of_valu_peak": tfl / pk, "valu_peak_tflops": pk, "arithmetic": mdl["dtype"], "traffic": None}
        tr, src = load_traffic(f"{nm_}|N{N}_B{B}")
        if tr:
            prod[nm_]["traffic"] = tr["hbm_traffic_bytes_per_launch"]
            prod[nm_]["traffic_over_algorithmic"] = tr["hbm_traffic_bytes_per_launch"] / (knots * mdl["bytes_per_unit"])
            prod[nm_]["traffic_source"] = src
    prod["generate_kkt"]["kernel_ms_kkt_f32"] = ms_kkt_f32
    prod["generate_kkt"]["frac_of_fp32_valu_peak_kkt_f32"] = knots * PRODUCER_MODEL["generate_kkt"]["flops_per_unit"] / (ms_kkt_f32 * 1e-3) / 1e12 / FP32_VALU_PEAK_TF
    prod["generate_kkt"]["kkt_f32"] = ("option \"kkt_f32\" = 1 (opt-in): float arithmetic (the reference's GRiD code: T = float), two knots per lane in v_pk_*_f32; "
                                       "outputs within 5e-6 of the float64 restatement instead of 2e-7")
    prod["generate_kkt"]["bound"] = "fp64 VALU issue: ~6,000 instructions per wavefront of four knots at 4 clocks each = 0.32 ms (its HBM floor is 0.04 ms)"
    prod["form_schur"]["kernels"] = "schur_walk_kernel + schur_seam_kernel (chunk length %d)" % sol.get_option("last_schur_chunk")
    prod["form_schur"]["bound"] = ("hbm: its output pattern alone (8,192 row streams of the bd layout, no arithmetic) takes 0.20 ms = 3.7 TB/s on this chip, "
                                   "~0.28 ms with the input reads (profiles/r04_walk_store_side.txt); VALU floor 0.27 ms")

    def previous_iterate_multipliers(d_xu_, d_goal_, d_xs_):
        # previous SQP iterate = this one plus a small change of the trajectory -> its multipliers (direct solve) are the warm start
        d_prev = d_xu_ + 2e-3 * torch.randn_like(d_xu_)
        d_prev[:, :14] = d_xu_[:, :14]
        Gp, Cp, gp, cp = sol.generate_kkt(plant, d_goal_, d_xs_, d_prev, iiwa.TIMESTEP, iiwa.QD_COST, rc)
        pS, _, pg = sol.form_schur(Gp, Cp, gp, cp, synth.RHO_INIT, "none")
        return sol.block_solve(pS, pg).clone()
    lam_prev = previous_iterate_multipliers(d_xu, d_goal, d_xs)
    resid = lambda S_, g_, l_: ((g_ - sol.bt_spmv(S_, l_)).double().norm(dim=1) / g_.double().norm(dim=1))
    res = {}
    for name, lam0 in (("cold", torch.zeros(B, 14 * N, device=dev)), ("warm", lam_prev)):
        l_ = lam0.clone()
        r0 = resid(rS, rg, l_)

        def go():
            l_.copy_(lam0)
            sol.solve(rS, rP, rg, l_, cfg, "ss", iters=d_it, exits=d_ex)
        ms_copy = timed(lambda: l_.copy_(lam0), 3, warm=1)
        ms_ = timed(go, 3, warm=1) - ms_copy               # (l_ now holds the solution of the last run)
        iti = d_it.cpu().numpy().astype(np.int64)
        r1 = resid(rS, rg, l_)
        res[name] = {"mean_pcg_iters": float(iti.mean()), "min_pcg_iters": int(iti.min()), "max_pcg_iters": int(iti.max()),
                     "max_iter_exit_rate": float(d_ex.float().mean().item()), "kernel_ms": ms_,
                     "pcg_iterations_per_sec": float(iti.sum() / (ms_ * 1e-3)), "linsolves_per_sec": B / (ms_ * 1e-3),
                     "true_rel_residual_before_median": float(r0.median().item()), "true_rel_residual_after_median": float(r1.median().item()),
                     "true_rel_residual_after_p90": float(torch.quantile(r1, 0.9).item()),
                     "true_rel_residual_after_max": float(r1.max().item()),
                     "worst_trajectory": int(r1.argmax().item()), "worst_trajectory_iters": int(iti[int(r1.argmax().item())]),
                     "worst_trajectory_residual_before": float(r0[int(r1.argmax().item())].item()),
                     "trajectories_whose_true_residual_grew": int((r1 > r0 * (1 + 1e-6)).sum().item())}
    # the dispatch-order hint predicts from the PREVIOUS call's iteration counts.  Exact: the timed repetitions replay one solve.  Off: no hint.
    sol.set_option("sched_hint", 0)
    l_nh = lam_prev.clone()
    ms_nh = timed(lambda: (l_nh.copy_(lam_prev), sol.solve(rS, rP, rg, l_nh, cfg, "ss", iters=d_it, exits=d_ex)), 3, warm=1) - timed(lambda: l_nh.copy_(lam_prev), 3, warm=1)
    sol.set_option("sched_hint", 1)
    res["warm"]["sched_hint"] = "on, with an exact prediction (the timed repetitions replay one solve)"
    res["warm"]["kernel_ms_sched_hint_off"] = ms_nh
    res["warm"]["linsolves_per_sec_sched_hint_off"] = B / (ms_nh * 1e-3)
    out = {"inputs": f"{B} windows of the reference trajectory examples/trajfiles/0_0_traj.csv (first 400 rows), random offset, goals 0..8 steps ahead, "
                     "state / iterate noise <= 0.05; KKT blocks by mpcg_generate_kkt (IIWA-14 dynamics on the device), Schur by mpcg_form_schur, rho = 1e-3",
           "pcg": {"max_iter": cx.max_iter, "exit_tol": args.exit_tol, "precond": "ss"}, "generate_kkt_ms": ms_kkt, "generate_kkt_f32_ms": ms_kkt_f32, "form_schur_ms": ms_schur, "compute_dz_ms": ms_dz,
           "cold_start": res["cold"], "warm_start_from_previous_sqp_iterate": res["warm"],
           "note": "cond(-S) of these systems is 1e7-2e7 at N=128 (synthetic generator: ~1e5): from lambda0 = 0 most solves hit the reference's iteration "
                   "cap, which presupposes the MPC loop's warm starts (tests/make_iiwa_golden.py prints the study)"}

    # ---- REALISTIC prediction: STEPS consecutive control steps of B MPC loops (mpc_window_sets); the timed loop walks
    # step 0, 1, 2, 3, 0, ... so that every solve's dispatch order was predicted from a DIFFERENT (the previous control step's) solve ----
    try:
        steps = [(rS, rP, rg, lam_prev)]
        for s in range(1, STEPS):
            xu_s, goals_s, xs_s = sets[s]
            dxu, dgo, dxs = f32(xu_s), f32(goals_s.reshape(B, -1)), f32(xs_s)
            Gs, Cs, gs, cs = sol.generate_kkt(plant, dgo, dxs, dxu, iiwa.TIMESTEP, iiwa.QD_COST, rc)
            sS, sP, sg = sol.form_schur(Gs, Cs, gs, cs, synth.RHO_INIT, "ss")
            steps.append((sS, sP, sg, previous_iterate_multipliers(dxu, dgo, dxs)))
        lws = [st[3].clone() for st in steps]
        its_steps = []

        def walk():
            for (S_, P_, g_, l0), lw in zip(steps, lws):
                lw.copy_(l0)
                sol.solve(S_, P_, g_, lw, cfg, "ss", iters=d_it, exits=d_ex)

        def copies():
            for (S_, P_, g_, l0), lw in zip(steps, lws):
                lw.copy_(l0)
        ms_walk = (timed(walk, 3, warm=1) - timed(copies, 3, warm=1)) / STEPS
        for (S_, P_, g_, l0), lw in zip(steps, lws):
            lw.copy_(l0)
            sol.solve(S_, P_, g_, lw, cfg, "ss", iters=d_it, exits=d_ex)
            its_steps.append(d_it.cpu().numpy().astype(np.int64))
        corr = float(np.mean([np.corrcoef(its_steps[s], its_steps[(s + 1) % STEPS])[0, 1] for s in range(STEPS)]))
        sol.set_option("sched_hint", 0)
        ms_walk_off = (timed(walk, 3, warm=1) - timed(copies, 3, warm=1)) / STEPS
        sol.set_option("sched_hint", 1)
        out["warm_start_consecutive_control_steps"] = {
            "what": f"{STEPS} consecutive control steps of {B} MPC loops (window shifted by one knot per step), walked cyclically: every solve's dispatch order "
                    "comes from the previous control step's iteration counts (a different solve of a related system)",
            "kernel_ms_per_step_hint_realistic": ms_walk, "linsolves_per_sec_hint_realistic": B / (ms_walk * 1e-3),
            "kernel_ms_per_step_hint_off": ms_walk_off, "linsolves_per_sec_hint_off": B / (ms_walk_off * 1e-3),
            "mean_pcg_iters": float(np.mean([i.mean() for i in its_steps])),
            "iteration_count_correlation_between_consecutive_steps": corr}
        del steps, lws
    except Exception as e_:
        out["warm_start_consecutive_control_steps"] = {"error": repr(e_)}

    # one whole linear-system step of an SQP iteration (include/pcg/sqp.cuh:190-259) as ONE hipGraph replay
    try:
        lam_g = lam_prev.clone()
        dz_g = torch.empty(B, 21 * N - 7, device=dev)

        def step():
            lam_g.copy_(lam_prev)
            G_, C_, g_, c_ = sol.generate_kkt(plant, d_goal, d_xs, d_xu, iiwa.TIMESTEP, iiwa.QD_COST, rc)
            S_, P_, gam_ = sol.form_schur(G_, C_, g_, c_, synth.RHO_INIT, "ss")
            sol.solve(S_, P_, gam_, lam_g, cfg, "ss", iters=d_it, exits=d_ex)
            sol.compute_dz(G_, C_, g_, lam_g, dz=dz_g)
        step()
        torch.cuda.synchronize()
        gr = torch.cuda.CUDAGraph()
        with torch.cuda.graph(gr):
            step()
        ms_step = timed(gr.replay, 4, warm=1)
        it_step = d_it.cpu().numpy().astype(np.int64)
        sol.set_option("kkt_f32", 1)
        try:
            step()
            torch.cuda.synchronize()
            gr32 = torch.cuda.CUDAGraph()
            with torch.cuda.graph(gr32):
                step()
            ms_step_f32 = timed(gr32.replay, 4, warm=1)
            it_step_f32 = float(d_it.cpu().numpy().astype(np.int64).mean())
            del gr32
        finally:
            sol.set_option("kkt_f32", 0)
        sqp_step = {"what": "generate_kkt -> form_schur (ss) -> PCG warm-started from the previous iterate's multipliers -> compute_dz, one hipGraph replay",
                    "ms_per_batch": ms_step, "batch": B, "sqp_linear_steps_per_sec": B / (ms_step * 1e-3), "us_per_trajectory_step": ms_step * 1e3 / B,
                    "mean_pcg_iters": float(it_step.mean()), "dz_finite": bool(torch.isfinite(dz_g).all().item()),
                    "ms_per_batch_kkt_f32": ms_step_f32, "mean_pcg_iters_kkt_f32": it_step_f32}
        del gr
    except Exception as e_:
        sqp_step = {"error": repr(e_)}
    out["sqp_linear_step_graph"] = sqp_step
    return out, prod


def producers_f64(cx):
    """linsys_t = double twins of the producers (include/common/settings.cuh:41-49): mpcg_form_schur_f64 / mpcg_compute_dz_f64 on the synthetic
    KKT blocks of the bench shape."""
    sol, dev, N = cx.sol, cx.dev, cx.N
    Bd = min(cx.B, 1024)
    k = synth.make_kkt(N, min(Bd, 64), 4242)
    Gh, Ch, gh, ch = synth.pack_kkt_dense(k, np.float64)
    rep = (Bd + Gh.shape[0] - 1) // Gh.shape[0]
    G0, C_, g_, c_ = (torch.from_numpy(np.tile(a, (rep, 1))[:Bd].copy()).to(dev) for a in (Gh, Ch, gh, ch))
    G = G0.clone()
    So = torch.empty(Bd, 3 * 196 * N, dtype=torch.float64, device=dev); Po = torch.empty_like(So)
    go = torch.empty(Bd, 14 * N, dtype=torch.float64, device=dev); dzo = torch.empty(Bd, 21 * N - 7, dtype=torch.float64, device=dev)
    ms_s = timed(lambda: (G.copy_(G0), sol.form_schur(G, C_, g_, c_, synth.RHO_INIT, "ss", S=So, Pinv=Po, gamma=go)), 3, warm=1) - timed(lambda: G.copy_(G0), 3, warm=1)
    lam = torch.randn(Bd, 14 * N, dtype=torch.float64, device=dev)
    ms_d = timed(lambda: sol.compute_dz(G, C_, g_, lam, dz=dzo), 3, warm=1)
    knots = Bd * N
    return {"batch": Bd, "knot_points": N,
            "form_schur_f64": {"kernel_ms": ms_s, "bytes_per_unit": 2 * 8036, "achieved": knots * 2 * 8036 / (ms_s * 1e-3) / 1e9, "unit": "GB/s",
                               "frac": knots * 2 * 8036 / (ms_s * 1e-3) / 1e9 / HBM_PEAK_GBS},
            "compute_dz_f64": {"kernel_ms": ms_d, "bytes_per_unit": 2 * 2380, "achieved": knots * 2 * 2380 / (ms_d * 1e-3) / 1e9, "unit": "GB/s",
                               "frac": knots * 2 * 2380 / (ms_d * 1e-3) / 1e9 / HBM_PEAK_GBS}}


def short_horizon(cx):
    """The short-horizon regime in throughput mode (N = 32, the reference's real-time horizon, 2048 trajectories: the lane-quad kernel's 32-knot build —
    two wavefronts, four workgroups per CU; until round 5 the lane-pair kernel's half build; the row-per-lane kernel serves latency-sized calls and N <= 16)."""
    sh = {}
    for pc_ in ("ss", "jacobi"):
        Ns, Bs = 32, 2048
        ss_ = PcgSolver(Ns, max_batch=Bs, device=cx.dev.index)
        S0, P0, g0 = build_inputs(ss_, Ns, 64, cx.seed0, pc_, cx.dev, chunk=64)
        Sl, Pl, gl = (t.repeat(Bs // 64, 1).contiguous() for t in (S0, P0, g0))
        cs = pcg_config(pcg_exit_tol=0.0, pcg_max_iter=synth.pcg_max_iter(Ns))
        ls = torch.zeros(Bs, 14 * Ns, device=cx.dev)
        is_ = torch.zeros(Bs, dtype=torch.int32, device=cx.dev)
        xs_ = torch.zeros(Bs, dtype=torch.uint8, device=cx.dev)

        def run_s():
            ls.zero_()
            ss_.solve(Sl, Pl, gl, ls, cs, pc_, iters=is_, exits=xs_)
        ms_s = timed(run_s, 5, warm=1)
        sh[pc_] = {"knots": Ns, "batch": Bs, "pcg_iters_per_solve": synth.pcg_max_iter(Ns), "kernel_ms": ms_s,
                   "pcg_iterations_per_sec": int(is_.sum().item()) / (ms_s * 1e-3), "kernel_family": ss_.get_option("last_kernel_family"),
                   "kernel_waves": ss_.get_option("last_kernel_waves")}
        del Sl, Pl, gl, ls, ss_
    return sh


def latency_config2(cx, reps=100):
    """BASELINE config 2: IIWA-14 N=32, ONE trajectory, block-Jacobi, max_iter 173 (settings.cuh:127), exit_tol 5e-6 (track_iiwa_pcg.cu:49),
    through the reference-shaped 12-argument entry.  Timed the way the reference times a linsolve (include/pcg/sqp.cuh:224-241): host
    monotonic clock around launch + the two D2H copies of (iters, exit), device-synchronised on both sides; plus the kernel by HIP events."""
    dev = cx.dev
    N = 32
    k = synth.make_kkt(N, 1, 1)
    S, P, g = synth.form_schur(k, precond="jacobi", poison_unused=True)
    sol = PcgSolver(N, max_batch=1, device=dev.index)
    d_S, d_P, d_g = (torch.from_numpy(a[0]).to(dev) for a in (S, P, g))
    d_lam = torch.zeros(14 * N, device=dev)
    d_it = torch.zeros(1, dtype=torch.int32, device=dev)
    d_ex = torch.zeros(1, dtype=torch.uint8, device=dev)
    cfg = pcg_config(pcg_exit_tol=5e-6, pcg_max_iter=synth.pcg_max_iter(N))
    wall, kern = [], []
    for i in range(reps + 10):
        d_lam.zero_()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0 = time.perf_counter()
        e0.record()
        sol.solve(d_S.view(1, -1), d_P.view(1, -1), d_g.view(1, -1), d_lam.view(1, -1), cfg, "jacobi", iters=d_it, exits=d_ex)
        e1.record()
        it = int(d_it.cpu().item())
        ex = int(d_ex.cpu().item())
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        if i >= 10:
            wall.append((t1 - t0) * 1e6)
            kern.append(e0.elapsed_time(e1) * 1e3)
    return {"workload": "IIWA-14 N=32, 1 trajectory, block-Jacobi, max_iter 173, exit_tol 5e-6 (BASELINE config 2)",
            "pcg_iters": it, "max_iter_exit": ex, "us_per_linsolve_wall_incl_2_d2h": float(np.median(wall)),
            "us_per_linsolve_kernel": float(np.median(kern)), "us_per_pcg_iter_kernel": float(np.median(kern)) / max(it, 1),
            "kernel_family": sol.get_option("last_kernel_family"), "kernel_waves": sol.get_option("last_kernel_waves")}


def batch1_sqp_step_latency(cx, horizons=(32, 64, 128)):
    """The reference's actual operating point (include/common/settings.cuh:161-163: a 2000 us SQP time box, ONE trajectory): the whole
    linear-system step of an SQP iteration — generate_kkt -> form_schur (ss) -> PCG warm-started from the previous iterate's multipliers ->
    compute_dz (include/pcg/sqp.cuh:190-259) — for one real IIWA-14 window, captured once as a hipGraph and replayed."""
    from mpcgpu_amd import Plant, iiwa
    dev, exit_tol = cx.dev, cx.args.exit_tol
    plant = Plant(device=dev.index)
    f32 = lambda a_: torch.from_numpy(np.ascontiguousarray(a_, np.float32)).to(dev)
    out = {}
    for N in horizons:
        try:
            sol = PcgSolver(N, max_batch=1, device=dev.index)
            W = 7                                                # windows sampled per horizon
            xu_h, goals_h, xs_h = iiwa.random_windows(N, W, 77 + N)
            gen = torch.Generator(device="cpu").manual_seed(1234 + N)
            rc = iiwa.r_cost(N)
            cfg = pcg_config(pcg_exit_tol=exit_tol, pcg_max_iter=synth.pcg_max_iter(N))
            d_xu, d_goal, d_xs = f32(xu_h[:1]), f32(goals_h[:1].reshape(1, -1)), f32(xs_h[:1])       # the graph's static inputs
            lam_prev = torch.zeros(1, 14 * N, device=dev)
            lam = lam_prev.clone()
            it = torch.zeros(1, dtype=torch.int32, device=dev)
            ex = torch.zeros(1, dtype=torch.uint8, device=dev)
            dz = torch.empty(1, 21 * N - 7, device=dev)

            def step():
                lam.copy_(lam_prev)
                G_, C_, g_, c_ = sol.generate_kkt(plant, d_goal, d_xs, d_xu, iiwa.TIMESTEP, iiwa.QD_COST, rc)
                S_, P_, gam_ = sol.form_schur(G_, C_, g_, c_, synth.RHO_INIT, "ss")
                sol.solve(S_, P_, gam_, lam, cfg, "ss", iters=it, exits=ex)
                sol.compute_dz(G_, C_, g_, lam, dz=dz)
            for _ in range(3):                                   # (scratch sizing; the symmetry latch settles)
                step()
                torch.cuda.synchronize()
            gr = torch.cuda.CUDAGraph()
            with torch.cuda.graph(gr):
                step()
            per = []
            for wdw in range(W):
                xw = f32(xu_h[wdw:wdw + 1])
                d_xu.copy_(xw); d_goal.copy_(f32(goals_h[wdw:wdw + 1].reshape(1, -1))); d_xs.copy_(f32(xs_h[wdw:wdw + 1]))
                xprev = xw + 2e-3 * torch.randn(xw.shape, generator=gen).to(dev)
                xprev[:, :14] = xw[:, :14]
                Gp, Cp, gp, cp = sol.generate_kkt(plant, d_goal, d_xs, xprev, iiwa.TIMESTEP, iiwa.QD_COST, rc)
                pS, _, pg = sol.form_schur(Gp, Cp, gp, cp, synth.RHO_INIT, "none")
                lam_prev.copy_(sol.block_solve(pS, pg))
                ms = timed(gr.replay, 20, warm=4)
                per.append({"us": ms * 1e3, "pcg_iters": int(it.item()), "max_iter_exit": int(ex.item()), "dz_finite": bool(torch.isfinite(dz).all().item())})
            us = sorted(p_["us"] for p_ in per)
            out[f"N{N}"] = {"us_per_step": us[W // 2], "us_min": us[0], "us_max": us[-1], "windows": W,
                            "pcg_iters": sorted(p_["pcg_iters"] for p_ in per), "max_iter_exits": sum(p_["max_iter_exit"] for p_ in per),
                            "dz_finite": all(p_["dz_finite"] for p_ in per),
                            "pcg_kernel_family": sol.get_option("last_kernel_family"), "schur_chunk": sol.get_option("last_schur_chunk"),
                            "fraction_of_the_2000us_sqp_time_box": us[W // 2] / 2000.0}
            del gr
        except Exception as e_:
            out[f"N{N}"] = {"error": repr(e_)}
    out["what"] = ("one trajectory: generate_kkt -> form_schur (ss) -> PCG from the previous iterate's multipliers -> compute_dz, one hipGraph replayed on 7 windows of the "
                   "reference trajectory (fixed seeds); us_per_step = the median window (each window: median of 20 replays)")
    return out


def single_trajectory_latency(cx):
    """The headline horizon as ONE trajectory (the reference's own mode of use): ms per SQP-linsolve."""
    N, dev = cx.N, cx.dev
    sol1 = PcgSolver(N, max_batch=1, device=dev.index)
    l1 = torch.zeros(1, 14 * N, device=dev)
    i1 = torch.zeros(1, dtype=torch.int32, device=dev)
    x1 = torch.zeros(1, dtype=torch.uint8, device=dev)

    def one():
        l1.zero_()
        sol1.solve(cx.d_S[:1], cx.d_P[:1], cx.d_g[:1], l1, cx.cfg, cx.args.precond, iters=i1, exits=x1)
    ms1 = timed(one, 25, warm=5)
    return {"workload": f"N={N}, {cx.args.precond}, ONE trajectory, max_iter={cx.max_iter}", "pcg_iters": int(i1.item()), "ms_per_linsolve": ms1,
            "us_per_pcg_iter": ms1 * 1e3 / max(int(i1.item()), 1),
            "kernel_family": sol1.get_option("last_kernel_family"), "kernel_waves": sol1.get_option("last_kernel_waves")}


def long_horizon(cx, lean=False):
    """Horizons one CU cannot hold (BASELINE config 5's N = 512, and N = 256): the clustered lane-pair kernel at the reference's iteration
    caps (settings.cuh:123-139); plus the all-streaming PCG kernel as an HBM
    roofline leg at N = 512.  Returns (long_horizon, roofline_pcg_streaming)."""
    dev, B = cx.dev, cx.B
    lh, stream = {}, None
    for Nl in ((512,) if lean else (256, 512)):
        sl = PcgSolver(Nl, max_batch=B, device=dev.index)
        S0, P0, g0 = build_inputs(sl, Nl, 32, cx.seed0, "ss", dev, chunk=32)
        rep = (B + 31) // 32
        Sl, Pl, gl = (t.repeat(rep, 1)[:B].contiguous() for t in (S0, P0, g0))
        del S0, P0
        cl = pcg_config(pcg_exit_tol=0.0, pcg_max_iter=synth.pcg_max_iter(Nl))
        ll = torch.zeros(B, 14 * Nl, device=dev)
        il = torch.zeros(B, dtype=torch.int32, device=dev)
        xl = torch.zeros(B, dtype=torch.uint8, device=dev)

        def run_b(nb):
            ll[:nb].zero_()
            sl.solve(Sl[:nb], Pl[:nb], gl[:nb], ll[:nb], cl, "ss", iters=il[:nb], exits=xl[:nb])
        if not lean:
            ms_b = timed(lambda: run_b(B), 5, warm=1)
            its = int(il.sum().item())
            ms_1 = timed(lambda: run_b(1), 15, warm=3)
            assert int(xl.max().item()) <= 1, "a cluster gave up"
            e = {"pcg_iters_per_solve": synth.pcg_max_iter(Nl), "batch": B, "kernel_ms": ms_b, "pcg_iterations_per_sec": its / (ms_b * 1e-3),
                 "ms_one_trajectory": ms_1, "us_per_pcg_iter_one_trajectory": ms_1 * 1e3 / synth.pcg_max_iter(Nl),
                 "kernel_family": sl.get_option("last_kernel_family"), "members_per_trajectory": sl.get_option("last_kernel_cluster"),
                 "cluster_fixups": sl.get_option("cluster_fixups")}
            lh[f"N{Nl}"] = e
        if Nl == 512:
            # the PCG solve as an HBM stream: nothing resident (pcg_traj_kernel<16,0,2>), every block re-read every iteration; one workgroup per
            # CU, the LIVE matrices are 256 x 2.41 MB = 616 MB >> the Infinity Cache: SURVEY §8d's model (2,577,344 B / iteration) applies as written
            ss = PcgSolver(Nl, max_batch=B, device=dev.index)
            ss.set_option("pcg_waves", 16); ss.set_option("pcg_reg_rows", 0); ss.set_option("pcg_lds_rows", 0)

            def stream_solve():
                ll.zero_()
                ss.solve(Sl, Pl, gl, ll, cl, "ss", iters=il, exits=xl)
            ms_s = timed(stream_solve, 3, warm=1) - timed(lambda: ll.zero_(), 3, warm=1)
            its_s = int(il.sum().item())
            b_it = synth.algorithmic_bytes(Nl, precond="ss")["pcg_iter"]
            ach = its_s * b_it / (ms_s * 1e-3) / 1e9
            assert ss.get_option("last_kernel_family") == 0 and ss.get_option("last_kernel_reg_rows") == 0
            occ = ss.checkPcgOccupancy()
            stream = {"bound": "hbm" if ach <= HBM_PEAK_GBS else "infinity cache + hbm (frac > 1: NOT an HBM roofline fraction)",
                      "kernel": "pcg_traj_kernel<16,0,2> (no resident rows)", "knot_points": Nl, "batch": B, "kernel_ms": ms_s, "achieved": ach,
                      "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS, "bytes_per_unit": b_it, "units_per_launch": its_s,
                      "resident_trajectories": occ, "live_matrix_set_mb": occ * 2 * 3 * 196 * Nl * 4 / 1e6, "infinity_cache_mb": 268.4,
                      "pcg_iterations_per_sec": its_s / (ms_s * 1e-3), "traffic": None}
            tr, src = load_traffic(f"pcg_traj_kernel<16,0,2>|N{Nl}_B{B}_ss_it{synth.pcg_max_iter(Nl)}_tol0")
            if tr:
                stream["traffic"] = tr["hbm_traffic_bytes_per_launch"]
                stream["traffic_source"] = src
            del ss
        del Sl, Pl, gl, ll, sl
    return lh, stream


def long_horizon_floor(cx, lh):
    """The clustered kernel against the floor of ITS design: classic PCG = 2 x (matrix pass + one L2 hand-off + barrier) per iteration
    (phase stamps: profiles/r04_lpkc_phases.txt, N = 256, shader clocks at 2.38 GHz)."""
    floor_ticks = 2 * (2090 + 635 + 140)
    roof = {"what": "clustered lane-pair kernel vs the floor of classic PCG on its decomposition: 2 x (matrix pass + one L2 hand-off + barrier) per iteration",
            "floor_us_per_iteration": floor_ticks / 2380.0, "floor_terms_shader_clocks": {"matrix_pass": 2090, "l2_handoff_min": 635, "barrier": 140},
            "above_the_floor_shader_clocks_per_half": {"fold_partials_and_halo_into_lds": 460, "operand_rebuild_and_scalar_chain": 360, "handoff_jitter_up_to": 640},
            "source": "profiles/r04_lpkc_phases.txt (tools/_prof/lpkc_phases.py, -DMPCG_PROF build)",
            # the single-reduction (Chronopoulos-Gear) recurrence, measured before building it — its hand-off after the Pinv pass
            # carries the neighbours' halo only; -DMPCG_CG_EMULATE builds exactly that hand-off into the shipping kernel (timing only)
            "single_reduction_ceiling": {"what": "the shipping kernel with the hand-off after the Pinv pass reduced to the halo granules (no partials polled, nothing folded): "
                                                 "what a Chronopoulos-Gear recurrence would save at best without overlap (tools/_prof/cg_emulate.py, profiles/r05_cg_emulate.txt)",
                                         "N256_batch1024_M_it_per_s": {"classic": 38.22, "emulated": 41.48, "target": 44.0},
                                         "N512_batch1024_M_it_per_s": {"classic": 17.76, "emulated": 19.16, "target": 20.5},
                                         "N256_one_trajectory_us_per_iteration": {"classic": 3.29, "emulated": 3.01, "target": 2.9},
                                         "gain": "+7.9 .. +9.3 %: below every target, before the variant's own extra vector recurrence (s = w + beta s) and with fixed-K iterates "
                                                 "that drift 5-10x more in float32: NOT BUILT, item closed"}}
    ncu_ = cx.sol.get_option("num_cus")
    for key_, v_ in lh.items():
        G_ = v_["members_per_trajectory"]
        resident = 8 * ((ncu_ // 8) // G_) if ncu_ >= 8 and ncu_ // 8 >= G_ else ncu_ // G_
        us_it = min(resident, cx.B) / (v_["pcg_iterations_per_sec"] * 1e-6)
        roof[key_] = {"pcg_iterations_per_sec": v_["pcg_iterations_per_sec"], "resident_clusters": resident, "us_per_iteration_full_batch": us_it,
                      "us_per_iteration_one_trajectory": v_["us_per_pcg_iter_one_trajectory"],
                      "frac_of_floor_full_batch": roof["floor_us_per_iteration"] / us_it,
                      "frac_of_floor_one_trajectory": roof["floor_us_per_iteration"] / v_["us_per_pcg_iter_one_trajectory"]}
    return roof


def double_precision(cx):
    """linsys_t = double (USE_DOUBLES of the reference): the bench workload's own systems in double.  Default policy beyond N = 32 with block-symmetric
    matrices: the lane-quad kernel (family 9: one CU to N = 64; family 10: ceil(N / 64) CUs of one XCD beyond) — the lower block triangle resident in
    registers; "pcg_lqk" = 0: the clustered row-per-lane kernel (family 8, full block rows on ceil(N / 32) CUs); "cluster" = 0: the streaming kernel
    (family 3), whose HBM roofline is reported next to it.  40 iterations per solve (a warm-started solve) and the reference's cap."""
    sol, dev, N, args = cx.sol, cx.dev, cx.N, cx.args
    Bd = min(cx.B, 1024)
    S64, P64, g64 = torch.nan_to_num(cx.d_S[:Bd]).double(), torch.nan_to_num(cx.d_P[:Bd]).double(), cx.d_g[:Bd].double()
    l64 = torch.zeros(Bd, 14 * N, dtype=torch.float64, device=dev)
    i64 = torch.zeros(Bd, dtype=torch.int32, device=dev); x64 = torch.zeros(Bd, dtype=torch.uint8, device=dev)
    t_zero = timed(lambda: l64.zero_(), 3, warm=1)

    def rate(K, reps=3):
        c64 = pcg_config(pcg_exit_tol=0.0, pcg_max_iter=K)
        go = lambda: (l64.zero_(), sol.solve_f64(S64, P64, g64, l64, c64, args.precond, iters=i64, exits=x64))
        go()
        ms = timed(go, reps, warm=1) - t_zero
        return ms, Bd * K / (ms * 1e-3)

    out = {"knot_points": N, "batch": Bd, "pcg_iters_per_solve": 40}
    ms, r40 = rate(40)
    fam = sol.get_option("last_kernel_family")
    bound = {10: "two cluster-wide hand-offs per iteration (lower block triangle of S and Pinv resident in the registers of ceil(N / 64) CUs)",
             9: "fp64 VALU issue / per-half latency, one active wavefront per SIMD (lower block triangle resident in one CU)",
             8: "two cluster-wide hand-offs per iteration (full block rows resident in the registers of ceil(N / 32) CUs)"}.get(fam, "hbm")
    out.update({"kernel_ms": ms, "pcg_iterations_per_sec": r40, "kernel_family": fam, "members_per_trajectory": sol.get_option("last_kernel_cluster"),
                "cluster_fixups": sol.get_option("cluster_fixups"), "bound": bound})
    # useful flop of one trajectory-iteration (bench.py's count for the headline kernel): two block-tridiagonal products + 2 inner products + 3 axpys
    flop = 2 * 196 * ((3 * N - 2) + ((3 * N - 2) if args.precond == "ss" else N)) + 10 * 14 * N
    ms_cap, r_cap = rate(cx.max_iter, reps=2)
    out.update({"flop_per_iteration": flop, "achieved_tflops": r40 * flop / 1e12, "frac_of_fp64_valu_peak": r40 * flop / 1e12 / FP64_VALU_PEAK_TF,
                "at_iteration_cap": {"pcg_iters_per_solve": cx.max_iter, "kernel_ms": ms_cap, "pcg_iterations_per_sec": r_cap,
                                     "frac_of_fp64_valu_peak": r_cap * flop / 1e12 / FP64_VALU_PEAK_TF}})
    try:
        if fam in (9, 10):
            sol.set_option("pcg_lqk", 0)
            ms_r, r_r = rate(40)
            out["row_per_lane_clusters"] = {"kernel_ms": ms_r, "pcg_iterations_per_sec": r_r, "kernel_family": sol.get_option("last_kernel_family"),
                                            "members_per_trajectory": sol.get_option("last_kernel_cluster")}
        sol.set_option("cluster", 0)
        ms_s, r_s = rate(40)                                                     # (first streaming call of a handle that does not know yet: the latch's one blocking check)
        cols64 = 2 if sol.get_option("symmetry_state") == 1 else 3
        by64 = 2 * cols64 * 196 * N * 8
        out["streaming_kernel"] = {"kernel_ms": ms_s, "pcg_iterations_per_sec": r_s, "kernel_family": sol.get_option("last_kernel_family"),
                                   "block_columns_read": cols64, "bytes_per_unit": by64, "achieved": r_s * by64 / 1e9, "peak": HBM_PEAK_GBS,
                                   "unit": "GB/s", "frac": r_s * by64 / 1e9 / HBM_PEAK_GBS, "bound": "hbm (S and Pinv re-read every iteration)"}
        out["speedup_over_streaming"] = ms_s / ms
    finally:
        sol.set_option("cluster", -1); sol.set_option("pcg_lqk", -1)
    # the longest horizon one CU holds in double
    if N > 64:
        sol64 = PcgSolver(64, max_batch=Bd, device=dev.index)
        S_, P_, g_ = (t.view(Bd, N, -1)[:, :64].reshape(Bd, -1).contiguous() for t in (S64, P64, g64))
        l_ = torch.zeros(Bd, 14 * 64, dtype=torch.float64, device=dev)
        c40 = pcg_config(pcg_exit_tol=0.0, pcg_max_iter=40)
        go64 = lambda: (l_.zero_(), sol64.solve_f64(S_, P_, g_, l_, c40, args.precond, iters=i64, exits=x64))
        go64()
        ms64 = timed(go64, 3, warm=1) - t_zero
        out["N64"] = {"kernel_ms": ms64, "pcg_iterations_per_sec": Bd * 40 / (ms64 * 1e-3), "kernel_family": sol64.get_option("last_kernel_family"),
                      "note": "the first 64 knots of the same systems (a leading principal block: still symmetric negative definite)"}
    return out


def scaling_expectation(cx):
    """What scaling to expect (measured here on one GPU, the multi-GPU curve itself is the driver's to measure)."""
    sol, B, cfg, args = cx.sol, cx.B, cx.cfg, cx.args
    Bq = max(1, B // 8)
    ms_q = timed(lambda: (cx.d_lam[:Bq].zero_(), sol.solve(cx.d_S[:Bq], cx.d_P[:Bq], cx.d_g[:Bq], cx.d_lam[:Bq], cfg, args.precond, iters=cx.d_it[:Bq], exits=cx.d_ex[:Bq])), 7, warm=2)
    ms_f = timed(lambda: (cx.d_lam.zero_(), cx.run_solve()), 5, warm=1)
    return {"weak": {"ceiling": "N x value(N=1): ranks are independent (one all-gather of 5 B per trajectory after the timed step)", "efficiency_expected": 1.0},
            "strong": {"global_batch": B, "batch_per_gpu_at_8": Bq, "ms_step_full_batch_1gpu": ms_f, "ms_step_at_batch_per_gpu_at_8": ms_q,
                       "speedup_ceiling_at_8_gpus": ms_f / ms_q, "efficiency_ceiling_at_8_gpus": ms_f / ms_q / 8.0,
                       "why": f"{Bq} trajectories = {Bq} workgroups on {sol.get_option('num_cus')} CUs: the step costs one trajectory's latency however few CUs are busy"}}


def block_solve(cx):
    """The other selectable solver on the same resident systems: batched block-tridiagonal direct solve (GPU twin of the QDLDL path)."""
    sol, dev, N, B = cx.sol, cx.dev, cx.N, cx.B
    lam_d = torch.empty(B, 14 * N, device=dev)
    ms_d = timed(lambda: sol.block_solve(cx.d_S, cx.d_g, lam_d), 5, warm=1)
    nb = min(4, B)
    Sd = cx.d_S[:nb].cpu().numpy()
    gd = cx.d_g[:nb].cpu().numpy()
    ld = lam_d[:nb].cpu().numpy().astype(np.float64)
    res = [float(np.linalg.norm(gd[b] - synth.bd_to_dense(np.nan_to_num(Sd[b]), N) @ ld[b]) / np.linalg.norm(gd[b])) for b in range(nb)]
    return {"kernel": "bt_block_solve_kernel (mpcg_block_solve)", "ms_per_batch": ms_d, "batch": B, "linsolves_per_sec": B / (ms_d * 1e-3),
            "us_per_linsolve_throughput": ms_d * 1e3 / B, "true_rel_residual_sample": res}


def cpu_baseline(cx, budget_s=12.0):
    """Reference CPU path (include/qdldl/sqp.cuh:22-49: numeric LDL^T factor + solve per linsolve, symbolic part amortised) — restated in
    oracle/ because the qdldl submodule is absent ("kind": "port").  Single thread: QDLDL is serial and the reference calls it from one thread."""
    orc = _oracle()
    N, S_host, P_host, g_host = cx.N, cx.S_h, cx.P_h, cx.g_h
    mean_iters = float(cx.it_host.mean())
    ns = min(32, S_host.shape[0])
    L = orc.LdlSolver(N, np.float32)
    vals = [orc.bd_to_csr_lowertri(np.nan_to_num(S_host[b]), N) for b in range(ns)]
    x = L.solve(vals[0], g_host[0])          # warm
    t0 = time.perf_counter()
    cnt = 0
    while time.perf_counter() - t0 < budget_s:
        for b in range(ns):
            x = L.solve(vals[b], g_host[b])
        cnt += ns
    dt = time.perf_counter() - t0
    resid = float(np.abs(orc.bt_spmv(np.nan_to_num(S_host[ns - 1]).astype(np.float64), x, N) - g_host[ns - 1]).max() / np.abs(g_host[ns - 1]).max())
    t1 = time.perf_counter()                 # same algorithm as the GPU (fp32 PCG, SS) on one CPU core, for a same-unit comparison
    it_cpu = 0
    nb = min(4, S_host.shape[0])
    for b in range(nb):
        r = orc.pcg(np.nan_to_num(S_host[b]), np.nan_to_num(P_host[b]), g_host[b], np.zeros(14 * N, np.float32), N, synth.pcg_max_iter(N), 1e-4, "ss")
        it_cpu += r["iters"]
    dt_pcg = time.perf_counter() - t1
    solves_per_s = cnt / dt
    ncore = os.cpu_count() or 1
    mt_cnt, mt_el = L.throughput(np.stack(vals), np.ascontiguousarray(g_host[:ns], np.float32), ncore, min(4.0, budget_s))
    mt_solves_per_s = mt_cnt / mt_el
    k32 = synth.make_kkt(32, 8, 32)          # BASELINE config 1: the reference's own CPU-runnable case, N=32
    S32, _, g32 = synth.form_schur(k32)
    L32 = orc.LdlSolver(32, np.float32)
    v32 = [orc.bd_to_csr_lowertri(S32[b], 32) for b in range(8)]
    t2 = time.perf_counter()
    c32 = 0
    while time.perf_counter() - t2 < 1.5:
        for b in range(8):
            L32.solve(v32[b], g32[b])
        c32 += 8
    us32 = (time.perf_counter() - t2) / c32 * 1e6
    return {"value": solves_per_s, "unit": "linsolves/s", "cores": 1, "kind": "port", "ms_per_linsolve": 1e3 / solves_per_s,
            "sample": f"{cnt} QDLDL-style f32 LDL^T factor+solve calls, first {ns} trajectories of the workload, {dt:.1f} s, 1 thread",
            "all_cores": {"value": mt_solves_per_s, "unit": "linsolves/s", "cores": ncore, "equiv_pcg_iters_per_sec": mt_solves_per_s * mean_iters},
            "config1_N32_us_per_linsolve": us32, "equiv_pcg_iters_per_sec": solves_per_s * mean_iters, "cpu_pcg_port_iters_per_sec": it_cpu / dt_pcg,
            "host_cpus": os.cpu_count(), "cpu_model": cpu_model(), "nnz": len(vals[0]), "dim": 14 * N,
            "rel_residual_float_ldl": resid, "meets_1e-4_residual_gate_of_BASELINE_md_3": bool(resid <= 1e-4),
            "residual_note": "QDLDL is built with float (Makefile:16 -DQDLDL_FLOAT=true); at cond(S) ~ 1e5 a float LDL^T cannot reach 1e-4 — "
                             "the port is timed as the reference runs it, the miss is reported, not hidden; the reference also pays D2H(values,gamma)+H2D(lambda) "
                             "per solve (include/qdldl/sqp.cuh:261-282), not included"}
