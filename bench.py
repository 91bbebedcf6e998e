#!/usr/bin/env python3
"""bench.py — headline benchmark of the PCG hot path (BASELINE.json metric).

  python bench.py [--gpus N] [--steps K] [--warmup W]
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
         --master-port P bench.py --gpus N --steps K --warmup W

One "step" = one batched SQP-linsolve: `batch` independent IIWA-14 trajectories (N=128 knots,
symmetric-stair preconditioner, lambda0 = 0, pcg_max_iter = 167, pcg_exit_tol = 1e-4) solved by ONE
launch of the persistent PCG kernel on every GPU.  Trajectories are independent, so ranks just get
their own `batch` (weak scaling) and the only collectives are the barrier, the max-over-ranks of
the time and the sum of the iteration counts (RCCL).  Inputs are synthetic (mpcgpu_amd.synth),
built on the host and resident in HBM before the timed region.

Prints ONE JSON line on rank 0: value = PCG iterations / second over all GPUs, plus
  roofline               the HBM-bound kernel of the path — the stand-alone block-tridiagonal SpMV (the
                         >= 60 % target) streamed over 1.2 GB of S (>> the 256 MiB Infinity Cache): frac <= 1 by construction
  roofline_pcg_streaming the PCG solve with NOTHING resident (every block re-read every iteration): the HBM model of
                         SURVEY §8d applies to it as written
  roofline_resident      the kernel `value` is measured on: register-resident, its HBM traffic is one read of the
                         lower block triangle per solve; bound = fp32 VALU, reported as useful TFLOP/s over 157.3
  parity_sample          sampled trajectories of the timed workload checked against the CPU oracle after the timed region
  cpu_baseline           (N=1 only) the reference's QDLDL CPU path, restated in oracle/, timed on this host.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from mpcgpu_amd import PcgSolver, pcg_config, synth  # noqa: E402
from mpcgpu_amd import dist as D  # noqa: E402

HBM_PEAK_GBS = 8000.0      # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured float4 copy)
FP32_VALU_PEAK_TF = 157.3  # same guide: peak fp32 vector = 256 CUs x 4 SIMDs x 64 flop/clk x 2.4 GHz


def build_inputs(sol, N, batch, seed0, precond, dev, chunk=128, rho=synth.RHO_INIT):
    """Synthetic IIWA-shaped KKT blocks (host, numpy) -> Schur systems on the GPU with the library's own
    mpcg_form_schur (the reference's form_schur_system step).  Returns device tensors (S, Pinv, gamma);
    input generation is outside every timed region."""
    S = torch.empty(batch, 3 * 196 * N, device=dev)
    P = torch.empty_like(S)
    g = torch.empty(batch, 14 * N, device=dev)
    for lo in range(0, batch, chunk):
        hi = min(batch, lo + chunk)
        # trajectory b of make_kkt(seed) depends only on (seed, b): offset through the seed
        k = synth.make_kkt(N, hi - lo, 900000 + seed0 + lo)
        Gd, Cd, gd, cd = (torch.from_numpy(a).to(dev) for a in synth.pack_kkt_dense(k, np.float32))
        sol.form_schur(Gd, Cd, gd, cd, rho, precond, S=S[lo:hi], Pinv=P[lo:hi], gamma=g[lo:hi])
    torch.cuda.synchronize()
    return S, P, g


def cpu_baseline(N, S_host, g_host, mean_iters, budget_s=12.0):
    """Reference CPU path (include/qdldl/sqp.cuh:22-49: numeric LDL^T factor + solve per linsolve,
    symbolic part amortised) — restated in oracle/ because the qdldl submodule is absent
    ("kind": "port").  Single thread: QDLDL is serial and the reference calls it from one thread."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import oracle as orc
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
    resid = float(np.abs(orc.bt_spmv(np.nan_to_num(S_host[ns - 1]).astype(np.float64), x, N) - g_host[ns - 1]).max()
                  / np.abs(g_host[ns - 1]).max())
    # same algorithm as the GPU (fp32 PCG, SS) on one CPU core, for a same-unit comparison
    t1 = time.perf_counter()
    it_cpu = 0
    nb = min(4, S_host.shape[0])
    for b in range(nb):
        r = orc.pcg(np.nan_to_num(S_host[b]), np.nan_to_num(P_HOST[b]), g_host[b], np.zeros(14 * N, np.float32),
                    N, synth.pcg_max_iter(N), 1e-4, "ss")
        it_cpu += r["iters"]
    dt_pcg = time.perf_counter() - t1
    solves_per_s = cnt / dt
    # the same port batch-parallel over every host core: one solver workspace per thread, trajectories
    # dealt round-robin (POSIX threads inside oracle/mpcg_oracle.c)
    ncore = os.cpu_count() or 1
    mt_cnt, mt_el = L.throughput(np.stack(vals), np.ascontiguousarray(g_host[:ns], np.float32), ncore, min(4.0, budget_s))
    mt_solves_per_s = mt_cnt / mt_el
    # BASELINE config 1: the reference's own CPU-runnable case, N=32 (include/common/settings.cuh:5-7 default)
    k32 = synth.make_kkt(32, 8, 32)
    S32, _, g32 = synth.form_schur(k32)
    L32 = orc.LdlSolver(32, np.float32)
    v32 = [orc.bd_to_csr_lowertri(S32[b], 32) for b in range(8)]
    t2 = time.perf_counter()
    c32 = 0
    while time.perf_counter() - t2 < 1.5:
        for b in range(8):
            x32 = L32.solve(v32[b], g32[b])
        c32 += 8
    us32 = (time.perf_counter() - t2) / c32 * 1e6
    return {
        "config1_N32_us_per_linsolve": us32,
        "value": solves_per_s, "unit": "linsolves/s", "cores": 1, "kind": "port",
        "ms_per_linsolve": 1e3 / solves_per_s,
        "all_cores": {"value": mt_solves_per_s, "unit": "linsolves/s", "cores": ncore,
                      "equiv_pcg_iters_per_sec": mt_solves_per_s * mean_iters},
        "equiv_pcg_iters_per_sec": solves_per_s * mean_iters,
        "cpu_pcg_port_iters_per_sec": it_cpu / dt_pcg,
        "sample": f"{cnt} QDLDL-style float32 LDL^T factor+solve calls over the first {ns} trajectories of the "
                  f"workload ({dt:.1f} s, 1 thread, nnz={len(vals[0])}, dim={14 * N}); rel. residual of the last float LDL^T solution against the fp32-built S {resid:.1e}; "
                  f"reference also pays D2H(values,gamma)+H2D(lambda) per solve (include/qdldl/sqp.cuh:261-282), not included",
        "host_cpus": os.cpu_count(), "cpu_model": cpu_model(),
        "rel_residual_float_ldl": resid, "meets_1e-4_residual_gate_of_BASELINE_md_3": bool(resid <= 1e-4),
        "residual_note": "QDLDL is built with float (Makefile:16 -DQDLDL_FLOAT=true); at cond(S) ~ 1e5 a float LDL^T cannot reach 1e-4 — "
                         "the port is timed as the reference runs it, the miss is reported, not hidden",
    }


class SclkSampler:
    """Shader clock and socket power while a leg runs: `rocm-smi --showclocks --showpower` from a thread (one call takes a few hundred ms; the
    sysfs pp_dpm_sclk table of this driver does not show the live clock of an MI300-class part)."""

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


def inrun_pmc(argv_tail, timeout_s=240):
    """HBM-side traffic of THIS run's kernels, measured by rocprofv3 in two child runs of this very script (`--profile-mini`: the
    timed region, the SpMV leg and the producer kernels, a few launches each): one --pmc FETCH_SIZE pass, one --pmc WRITE_SIZE pass,
    counters only with --kernel-trace as /opt/skills/guides/MI355X_MICROARCH.md prescribes (FETCH_SIZE doubled: gfx950 tallies
    128-byte read requests at 64 bytes; WRITE_SIZE as reported — it matches the known store bytes of form_schur to 0.4 %).
    Returns ({kernel name: {...}}, note) or (None, why not)."""
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
        cmd = [exe, "--kernel-trace", "--pmc", counter, "-d", d, "-o", "p", "--", sys.executable, os.path.abspath(__file__), "--steps", "3", "--warmup", "1",
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
            # several launch shapes of one kernel (the bench launches some kernels on more than one batch): keep the largest grid
            if nm not in out or (grid or 0) > (out[nm]["grid"] or 0):
                out[nm] = e
    return out, f"measured in this run: 2 rocprofv3 --kernel-trace --pmc child passes of `bench.py --profile-mini` ({time.perf_counter() - t0:.0f} s)"


def find_traffic(pmc, needle):
    if not pmc:
        return None
    for nm, e in pmc.items():
        if needle in nm:
            return e
    return None


# Algorithmic bytes and arithmetic of the producer kernels, per knot (n = 14, m = 7):
#   form_schur (ss): reads  G 245 + C 294 + g 21 + c 14 floats, writes S 3x196 + Pinv 3x196 + G^-1 245 + gamma 14  = 2,009 floats = 8,036 B
#   compute_dz:      reads  G^-1 245 + C 294 + g 21 + lambda 14 (+14 of the next knot, L2), writes dz 21             = 2,380 B
#   generate_kkt:    reads  x,u,x+ 35 + goals 12 floats, writes G 245 + C 294 + g 21 + c 14                         = 2,484 B
# flops: form_schur 1,323 multiply-adds in seven 14x14(x7) products + three Gauss-Jordan inversions (2 x 14^3 + 7^3 = 5,831 multiply-subtracts)
# + 4 matrix-vector products = ~14.7 kflop (fp32); generate_kkt 25 recursive Newton-Euler sweeps x ~2,800 flop + Cholesky/solves ~2 kflop = ~72 kflop (fp64).
PRODUCER_MODEL = {
    "form_schur": {"bytes_per_unit": 8036, "flops_per_unit": 2 * (1323 + 5831 + 4 * 196), "dtype": "f32"},
    "compute_dz": {"bytes_per_unit": 2380, "flops_per_unit": 2 * (2 * 196 + 98 + 49), "dtype": "f32"},
    "generate_kkt": {"bytes_per_unit": 2484, "flops_per_unit": 25 * 2800 + 2000, "dtype": "f64"},
}
FP64_VALU_PEAK_TF = 78.6   # MI355X fp64 vector peak (same guide)


P_HOST = None


def latency_config2(dev, reps=100):
    """BASELINE config 2: IIWA-14 N=32, ONE trajectory, block-Jacobi, max_iter 173 (settings.cuh:127),
    exit_tol 5e-6 (track_iiwa_pcg.cu:49), through the reference-shaped 12-argument entry.  Timed the way
    the reference times a linsolve (include/pcg/sqp.cuh:224-241): host monotonic clock around launch + the two
    D2H copies of (iters, exit), device-synchronised on both sides; plus the bare kernel by HIP events."""
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


def batch1_sqp_step_latency(dev, exit_tol, horizons=(32, 64, 128)):
    """The reference's actual operating point (include/common/settings.cuh:161-163: a 2000 us SQP time box, ONE trajectory): the whole linear-system
    step of an SQP iteration — generate_kkt -> form_schur (ss) -> PCG warm-started from the previous iterate's multipliers -> compute_dz
    (include/pcg/sqp.cuh:190-259) — for one real IIWA-14 window, captured once as a hipGraph and replayed: median latency by HIP events."""
    from mpcgpu_amd import Plant, iiwa
    plant = Plant(device=dev.index)
    f32 = lambda a_: torch.from_numpy(np.ascontiguousarray(a_, np.float32)).to(dev)
    out = {}
    for N in horizons:
        try:
            sol = PcgSolver(N, max_batch=1, device=dev.index)
            W = 7                                                # windows sampled per horizon (each: its own trajectory piece, goal, warm start)
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
                # the previous SQP iterate = this trajectory plus a small change; its multipliers (direct solve) are the warm start
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
        except Exception as e_:                                  # (reported, never fatal for the headline measurement)
            out[f"N{N}"] = {"error": repr(e_)}
    out["what"] = ("one trajectory: generate_kkt -> form_schur (ss) -> PCG from the previous iterate's multipliers -> compute_dz, one hipGraph replayed on 7 windows of the "
                   "reference trajectory (fixed seeds); us_per_step = the median window (each window: median of 20 replays); pcg_iters = the windows' counts")
    return out


def fp32_check(orc, S, P, g, lam_gpu, it_gpu, N, pc):
    """One sampled trajectory against the oracle: the float64 iterate after the SAME number of iterations, the band
    the CPU float32 restatement reaches on the same inputs (tests/util.py:fp32_band, no perturbation trials here),
    and the true residuals."""
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


def cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


RAMP_S = 0.02      # every timed leg first runs its function back to back for this long


def timed(fn, reps, warm=1):
    """Median HIP-event time (ms) of `fn` over `reps` runs on the current stream (= the stream the kernels are launched on).
    The function first runs back to back for RAMP_S seconds: a leg of a few sub-millisecond launches after a host-side pause is otherwise timed on a
    GPU that has not reached its steady clocks (the timed region itself: 125-129 M it/s with 10 steps after 2, 138-140 M with 200 after 20)."""
    t_end = time.perf_counter() + RAMP_S
    while time.perf_counter() < t_end:
        fn()
        torch.cuda.synchronize()            # (bounds the ramp in GPU time: unsynchronised, 20 ms of host time enqueue seconds of work)
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


def load_traffic(kernel_key):
    """HBM-side traffic of one launch from this round's committed PMC passes (profiles/traffic.json, written by
    tools/profile_round.sh from rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE runs of this same bench command; bench cannot
    run rocprofv3 on itself)."""
    try:
        t = json.load(open(os.path.join(ROOT, "profiles", "traffic.json")))
        return t.get("kernels", {}).get(kernel_key), t.get("source")
    except (OSError, ValueError):
        return None, None


def _free_port():
    import socket
    sk = socket.socket()
    sk.bind(("127.0.0.1", 0))
    port = sk.getsockname()[1]
    sk.close()
    return port


def self_launch(args):
    """`python bench.py --gpus N` (N > 1) without a launcher around it: re-run this very command line as N ranks under
    torch.distributed.run (one process per GPU, rendezvous on 127.0.0.1, a free port), stream their output through and
    return their exit status.  Fails loudly when the node has fewer than N devices — it never degrades to fewer ranks."""
    import subprocess
    n = args.gpus
    if not args.plumbing_only and not os.environ.get("MPCG_FORCE_DEVICE"):
        have = torch.cuda.device_count() if torch.cuda.is_available() else 0
        if have < n:
            print(f"bench.py: --gpus {n} requested but {have} HIP device(s) visible on this node; refusing to run fewer ranks", file=sys.stderr)
            return 2
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")      # dmabuf IPC: RCCL across processes needs it on this driver
    env["MPCG_BENCH_SELF_LAUNCHED"] = "1"
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.abspath(__file__), *sys.argv[1:]]
    print("bench.py: launching " + " ".join(cmd), file=sys.stderr, flush=True)
    return subprocess.call(cmd, env=env)


def plumbing_only(args, rank, world):
    """--plumbing-only: the distributed flow of main() with made-up per-trajectory results and no solve (CPU, gloo)."""
    B_global = args.batch if args.scaling == "strong" else args.batch * world
    lo, hi = D.shard_range(args.batch, rank, world) if args.scaling == "strong" else (rank * args.batch, (rank + 1) * args.batch)
    it = torch.arange(lo, hi, dtype=torch.int32) % 97 + 1          # "iterations" of trajectory b: b % 97 + 1
    ex = (torch.arange(lo, hi) % 5 == 0).to(torch.uint8)
    D.barrier()
    t_all = D.max_over_ranks(1.0 + rank)
    its_all = D.sum_over_ranks(float(it.sum()))
    per_rank = D.all_gather_floats(float(rank) + 0.5)
    g_it, g_ex = D.gather_results(it, ex, B_global if args.scaling == "strong" else None)
    want = torch.arange(B_global, dtype=torch.int32) % 97 + 1
    ok = bool(torch.equal(g_it.cpu(), want) and int(g_it.sum()) == int(its_all) and t_all == float(world))
    out = {"metric": "pcg_iterations_per_sec", "value": None, "unit": "iter/s", "n_gpus": world, "plumbing_only": True, "scaling": args.scaling,
           "self_launched": os.environ.get("MPCG_BENCH_SELF_LAUNCHED") == "1",
           "results_gather": {"backend": D.backend_name(), "trajectories": int(g_it.numel()), "consistent_with_allreduce_sum": ok},
           "per_rank": per_rank, "shard": [lo, hi]}
    if torch.distributed.is_initialized():
        torch.distributed.destroy_process_group()
    if rank == 0:
        print(json.dumps(out), flush=True)
    return 0 if ok else 1


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)     # (0.25 s of timed steps: the GPU's steady state; 10 after 2 measure the clock ramp, see timed())
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--knots", type=int, default=128)
    ap.add_argument("--batch", type=int, default=1024, help="trajectories per GPU (--scaling weak) or in total (--scaling strong)")
    ap.add_argument("--scaling", default="weak", choices=["weak", "strong"],
                    help="weak: every rank solves --batch trajectories; strong: --batch trajectories are sharded over the ranks "
                         "(BASELINE config 4 as written: 1024 over 8 GPUs = 128 each)")
    ap.add_argument("--precond", default="ss", choices=["ss", "jacobi"])
    ap.add_argument("--exit-tol", type=float, default=1e-4)
    ap.add_argument("--max-iter", type=int, default=0, help="0 = reference table (settings.cuh:123-139)")
    ap.add_argument("--pcg-waves", type=int, default=0)
    ap.add_argument("--reg-rows", type=int, default=-1)
    ap.add_argument("--lds-rows", type=int, default=-2)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    ap.add_argument("--no-extras", action="store_true", help="only the timed region (no roofline side kernels, parity sample, warm run, latency)")
    ap.add_argument("--profile-lean", action="store_true",
                    help="for rocprofv3 passes: only the timed region + the two HBM roofline kernels, so that per-kernel averages are not "
                         "diluted by the batch-1 latency launches, the 4000-iteration warm-start reference solve and the parity sample")
    ap.add_argument("--profile-mini", action="store_true",
                    help="what the in-run rocprofv3 passes execute: the timed region, the SpMV leg and the producer kernels (one launch shape each), nothing else")
    ap.add_argument("--no-inrun-pmc", action="store_true", help="do not spawn the two rocprofv3 --pmc child passes (roofline.traffic then comes from profiles/traffic.json)")
    ap.add_argument("--sustain-seconds", type=float, default=3.0, help="length of the sustained leg (back-to-back timed steps; 0 = skip)")
    ap.add_argument("--spmv-mfma", action="store_true", help="also time BASELINE config 5's MFMA block-GEMV experiment")
    ap.add_argument("--spmv-batch", type=int, default=4096, help="trajectories streamed by the SpMV roofline run (S = batch x 301 KB)")
    ap.add_argument("--storage", default="f32", choices=["f32", "f16"],
                    help="matrix storage of S/Pinv; f16 = BASELINE config 5's reduced-precision experiment (arithmetic stays fp32)")
    ap.add_argument("--plumbing-only", action="store_true",
                    help="launcher / collective dry run (CPU test of the N > 1 flow, tests/test_bench_launcher.py): every rank runs the whole "
                         "distributed flow of this file — rendezvous, barrier, shard ranges, all-reduces, all-gather of per-trajectory results — "
                         "with made-up iteration counts and NO solve; prints value = null")
    args = ap.parse_args()
    if args.profile_mini:
        args.profile_lean = True
        args.no_cpu_baseline = True

    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        # bare `python bench.py --gpus N`: this process becomes the launcher of N ranks (one per GPU, RCCL), it does not
        # degrade to one rank
        sys.exit(self_launch(args))
    rank, local_rank, world = D.init()
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world} — launch with --nproc-per-node {args.gpus} "
                         f"(or run `python bench.py --gpus {args.gpus}` without a launcher: it starts its own ranks)")
    if args.plumbing_only:
        return plumbing_only(args, rank, world)
    assert torch.cuda.is_available(), "bench.py needs a HIP device (the product has no CPU path)"
    if torch.cuda.device_count() < (1 if os.environ.get("MPCG_FORCE_DEVICE") else world):
        raise SystemExit(f"bench.py: {world} ranks but only {torch.cuda.device_count()} HIP device(s) visible")
    if os.environ.get("MPCG_FORCE_DEVICE"):          # dry-run of the N>1 flow on a 1-GPU box (with MPCG_DIST_BACKEND=gloo)
        local_rank = int(os.environ["MPCG_FORCE_DEVICE"])
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)

    N = args.knots
    if args.scaling == "strong":
        lo, hi = D.shard_range(args.batch, rank, world)
        B, seed0, B_global = hi - lo, lo, args.batch
    else:
        B, seed0, B_global = args.batch, rank * args.batch, args.batch * world
    assert B > 0, "more ranks than trajectories"
    max_iter = args.max_iter or synth.pcg_max_iter(N)
    cfg = pcg_config(pcg_exit_tol=args.exit_tol, pcg_max_iter=max_iter)

    global P_HOST
    sol = PcgSolver(N, max_batch=max(B, args.spmv_batch if not args.no_extras else B), device=local_rank)
    d_S, d_P, d_g = build_inputs(sol, N, B, seed0, args.precond, dev, chunk=B if args.profile_mini else 128)
    ns = min(32, B)
    S_h, P_h, g_h = (t[:ns].cpu().numpy() for t in (d_S, d_P, d_g))     # host copies of the CPU-baseline sample
    P_HOST = P_h
    d_lam = torch.zeros(B, 14 * N, device=dev)
    d_it = torch.zeros(B, dtype=torch.int32, device=dev)
    d_ex = torch.zeros(B, dtype=torch.uint8, device=dev)
    if args.pcg_waves:
        sol.set_option("pcg_waves", args.pcg_waves)
    if args.reg_rows >= 0:
        sol.set_option("pcg_reg_rows", args.reg_rows)
    if args.lds_rows >= -1:
        sol.set_option("pcg_lds_rows", args.lds_rows)

    if args.storage == "f16":
        d_S16, d_P16 = sol.to_f16(d_S), sol.to_f16(d_P)

        def run_solve():
            sol.solve_f16(d_S16, d_P16, d_g, d_lam, cfg, args.precond, iters=d_it, exits=d_ex)
    else:
        def run_solve():
            sol.solve(d_S, d_P, d_g, d_lam, cfg, args.precond, iters=d_it, exits=d_ex)

    # like every leg (timed()): the clocks first, then the caller's W warm-up steps, then exactly K timed steps
    t_ramp = time.perf_counter() + 2.5 * RAMP_S
    while time.perf_counter() < t_ramp:
        d_lam.zero_()
        run_solve()
        torch.cuda.synchronize()
    for _ in range(args.warmup):
        d_lam.zero_()                       # every step is the same cold-start solve
        run_solve()
    torch.cuda.synchronize()

    # --- timed region: exactly K steps, barrier + synchronize on both sides, max over ranks ---
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]
    # the reference's linsolve ends with the D2H copies of (pcg_iters, pcg_exit) (include/pcg/sqp.cuh:231-232): SURVEY §8d puts them
    # INSIDE the timed region — here batch x 5 bytes per step into pinned host memory, on the solve's stream
    h_it = torch.empty(B, dtype=torch.int32).pin_memory()
    h_ex = torch.empty(B, dtype=torch.uint8).pin_memory()
    D.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(args.steps):
        d_lam.zero_()
        ev[i][0].record()                   # HIP events on the stream the kernel is launched on
        run_solve()
        ev[i][1].record()
        h_it.copy_(d_it, non_blocking=True)
        h_ex.copy_(d_ex, non_blocking=True)
    torch.cuda.synchronize()
    D.barrier()
    t_local = time.perf_counter() - t0
    t_all = D.max_over_ranks(t_local, dev)

    fam = {0: "pcg_traj_kernel", 3: "pcg_generic_kernel", 5: "pcg_rpl_kernel", 6: "pcg_lpk_kernel", 7: "pcg_lpkc_kernel"}[sol.get_option("last_kernel_family")]
    kdesc = {"family": fam, **{k: sol.get_option("last_kernel_" + k) for k in ("waves", "reg_rows", "lds_rows", "stream_bufs", "cluster", "lds_bytes")}}
    it_host = h_it.numpy().astype(np.int64)          # what the timed region's last step copied back
    ex_host = h_ex.numpy().copy()
    iters_step_local = int(it_host.sum())
    iters_step_all = D.sum_over_ranks(iters_step_local, dev)
    kern_ms = float(np.mean([a.elapsed_time(b) for a, b in ev]))
    kern_ms_all = D.max_over_ranks(kern_ms, dev)
    kern_ms_ranks = D.all_gather_floats(kern_ms, dev)
    # the only collective of the path: per-trajectory (iters, exit flag) of every shard to every rank (RCCL all-gather)
    g_it, g_ex = D.gather_results(d_it, d_ex, B_global if args.scaling == "strong" else None)
    gathered_ok = bool(int(g_it.to(torch.int64).sum().item()) == int(iters_step_all) and g_it.numel() == B_global)

    ms_per_step = 1e3 * t_all / args.steps
    value = iters_step_all / (t_all / args.steps)

    # ---- sustained leg: the very same step back to back for >= --sustain-seconds (steady-state clocks; a 25 ms timed region is invisible to
    # a 5 s utilisation sampler) ----
    sustained = None
    if args.sustain_seconds > 0 and not args.profile_lean:
        n_chunk = max(8, int(0.25 / max(ms_per_step * 1e-3, 1e-5)))          # ~0.25 s of steps between two host synchronisations
        done, t_s0 = 0, time.perf_counter()
        D.barrier()
        with SclkSampler(local_rank) as clk:
            while time.perf_counter() - t_s0 < args.sustain_seconds:
                for _ in range(n_chunk):
                    d_lam.zero_()
                    run_solve()
                    h_it.copy_(d_it, non_blocking=True)
                    h_ex.copy_(d_ex, non_blocking=True)
                torch.cuda.synchronize()
                done += n_chunk
        t_sus = D.max_over_ranks(time.perf_counter() - t_s0, dev)
        sustained = {"steps": done, "seconds": t_sus, "ms_per_step": 1e3 * t_sus / done, "value": iters_step_all / (t_sus / done), "sclk_mhz": clk.summary(),
                     "what": "the timed region's step (lambda <- 0; mpcg_pcg_solve; D2H of iters / exit flags) repeated back to back, host wall clock, max over ranks"}
    bytes_iter = synth.algorithmic_bytes(N, precond=args.precond)["pcg_iter"]     # fp32-storage model
    # useful flops of one PCG iteration of one trajectory: two block-tridiagonal products + 2 inner products + 3 axpys
    nblk = (3 * N - 2) + ((3 * N - 2) if args.precond == "ss" else N)
    flops_iter = 2 * 196 * nblk + 10 * 14 * N

    out = {
        "metric": "pcg_iterations_per_sec", "value": value, "unit": "iter/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms_per_step,
        "higher_is_better": True, "scaling": args.scaling, "vs_baseline": None, "dtype": "f32" if args.storage == "f32" else "f32 arithmetic, f16 matrix storage",
        "data": "synthetic",
        "config": {"workload": f"IIWA-14 (n=14) N={N} knots, {args.precond} preconditioner, batch {B} trajectories on this GPU / {B_global} in total "
                               f"(BASELINE config 4: 1024 trajectories; S+Pinv = {2 * B * 3 * 196 * N * 4 / 1e6:.0f} MB in HBM per GPU), lambda0=0, "
                               f"max_iter={max_iter}, exit_tol={args.exit_tol:g}",
                   "knot_points": N, "state_size": 14, "batch_per_gpu": B, "global_batch": B_global,
                   "precond": args.precond, "pcg_max_iter": max_iter, "pcg_exit_tol": args.exit_tol,
                   "parallelism": f"batch-sharded x{world} ({args.scaling} scaling, no data-path collective)", "kernel": kdesc},
        "ms_per_linsolve": ms_per_step / B,
        "linsolves_per_sec": B_global / (ms_per_step * 1e-3),
        "mean_pcg_iters": float(it_host.mean()), "max_iter_exit_rate": float(ex_host.mean()),
        "resident_trajectories_per_gpu": sol.checkPcgOccupancy(),
        "results_gather": {"collective": "all_gather of (iters, exit) per trajectory", "backend": D.backend_name(),
                           "trajectories": int(g_it.numel()), "consistent_with_allreduce_sum": gathered_ok},
        "per_rank_kernel_ms": kern_ms_ranks,
        "timed_region": "K x [lambda <- 0; mpcg_pcg_solve; D2H of (iters u32, exit u8) per trajectory], barrier + synchronize on both sides, max over ranks",
        "self_launched": os.environ.get("MPCG_BENCH_SELF_LAUNCHED") == "1",
    }

    # ---- the kernel `value` is measured on ----
    dom = {"kernel": fam, "kernel_ms": kern_ms, "kernel_ms_max_over_ranks": kern_ms_all, "units_per_launch": iters_step_local,
           "unit_of_work": "one PCG iteration of one trajectory"}
    if fam == "pcg_traj_kernel" and kdesc["reg_rows"] == 0:
        # nothing resident: the HBM model applies as written
        ach = iters_step_local * bytes_iter / (kern_ms * 1e-3) / 1e9
        out["roofline"] = {"bound": "hbm", **dom, "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS,
                           "traffic": None, "bytes_per_unit": bytes_iter}
    else:
        tf = iters_step_local * flops_iter / (kern_ms * 1e-3) / 1e12
        model_gbs = iters_step_local * bytes_iter / (kern_ms * 1e-3) / 1e9
        rr = {"bound": "valu", **dom, "achieved": tf, "peak": FP32_VALU_PEAK_TF, "unit": "TFLOP/s", "frac": tf / FP32_VALU_PEAK_TF,
              "flops_per_unit": flops_iter,
              "why_not_hbm": "S and Pinv stay in registers for the whole solve: HBM sees one read of the lower block triangle per "
                             "solve, not one per iteration; the streaming-model rate below exceeds the HBM peak and is NOT a roofline fraction",
              "hbm_streaming_model": {"bytes_per_unit": bytes_iter, "rate_gbs": model_gbs, "speedup_vs_hbm_streaming_ceiling": model_gbs / HBM_PEAK_GBS},
              "hbm_algorithmic_bytes_per_launch": B * (2 * (2 * N - 1) * 784 + 3 * 14 * N * 4 + 5),
              "traffic": None}
        tr, src = load_traffic(f"{fam}|N{N}_B{B}_{args.precond}_it{max_iter}_tol{args.exit_tol:g}")
        if tr:
            rr["traffic"] = tr["hbm_traffic_bytes_per_launch"]
            rr["traffic_source"] = src
            rr["traffic_gbs"] = tr["hbm_traffic_bytes_per_launch"] / (kern_ms * 1e-3) / 1e9
            rr["traffic_frac_of_hbm_peak"] = rr["traffic_gbs"] / HBM_PEAK_GBS
            rr["traffic_is"] = "bytes per launch from the builder's rocprofv3 --pmc passes of this command (not measured in this run); the rate uses this run's kernel time"
            if "valu_active_frac" in tr:
                rr["valu_active_frac"] = tr["valu_active_frac"]      # SQ_ACTIVE_INST_VALU / (4 SIMDs x SQ_BUSY_CYCLES), same passes
        out["roofline_resident"] = rr

    extras = not args.no_extras and args.storage == "f32"
    if extras:
        # ---- HBM roofline 1: stand-alone block-tridiagonal SpMV, S streamed from HBM (>> L3) ----
        Bs = max(B, args.spmv_batch)
        reps = (Bs + B - 1) // B
        S_big = d_S.repeat(reps, 1)[:Bs].contiguous() if reps > 1 else d_S
        x = torch.randn(Bs, 14 * N, device=dev)
        y = torch.empty_like(x)
        def spmv20():
            for _ in range(20):
                sol.bt_spmv(S_big, x, y)
        ms = timed(spmv20, 3, warm=1) / 20          # 20 back-to-back launches per timing, like the rocprofv3 average
        b_sp = synth.algorithmic_bytes(N)["spmv"] * Bs
        sp = {"bound": "hbm", "kernel": "bt_spmv_kernel", "achieved": b_sp / (ms * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
              "frac": b_sp / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, "kernel_ms": ms, "traffic": None,
              "algorithmic_bytes_per_launch": b_sp, "bytes_per_unit": synth.algorithmic_bytes(N)["spmv"], "units_per_launch": Bs,
              "unit_of_work": "one block-tridiagonal SpMV of one trajectory (SURVEY §8d)", "working_set_mb": Bs * 3 * 196 * N * 4 / 1e6,
              "frac_of_measured_copy_ceiling_6.29TBs": b_sp / (ms * 1e-3) / 1e9 / 6290.0,
              "trajectory_spmv_per_sec": Bs / (ms * 1e-3)}
        # what THIS box's HBM delivers to the plainest stream there is: a device-to-device copy of the same 1.2 GB (read + write bytes);
        # the SpMV fraction moves with it from box to box (0.65 .. 0.77 of the 8 TB/s peak seen across the pool)
        try:
            dst = torch.empty_like(S_big)
            ms_cp = timed(lambda: dst.copy_(S_big), 5, warm=2)
            cp_gbs = 2.0 * S_big.numel() * 4 / (ms_cp * 1e-3) / 1e9
            sp["d2d_copy_gbs_this_run"] = cp_gbs
            sp["frac_of_d2d_copy_this_run"] = sp["achieved"] / cp_gbs
            del dst
        except Exception as e_:
            sp["d2d_copy_gbs_this_run"] = None
        # ... and to a pure read of the same array (the library's probe kernel: 16-byte nontemporal loads, nothing else)
        try:
            sink = torch.zeros(1, device=dev)
            def rd20():
                for _ in range(20):
                    sol.probe_hbm_read(S_big, sink)
            ms_rd = timed(rd20, 3, warm=1) / 20
            sp["read_ceiling_gbs_this_run"] = S_big.numel() * 4 / (ms_rd * 1e-3) / 1e9
            sp["frac_of_read_ceiling_this_run"] = sp["achieved"] / sp["read_ceiling_gbs_this_run"]
        except Exception as e_:
            sp["read_ceiling_gbs_this_run"] = None
        tr, src = load_traffic(f"bt_spmv_kernel|N{N}_B{Bs}")
        if tr:
            sp["traffic"] = tr["hbm_traffic_bytes_per_launch"]
            sp["traffic_source"] = src
            sp["traffic_is"] = "bytes per launch from the builder's rocprofv3 --pmc passes of this command (not measured in this run)"
        if "roofline" not in out:
            out["roofline"] = sp
            # the kernel `value` is measured on, inside the object the driver keeps: nested AND as flat scalars
            rr = out.get("roofline_resident")
            if rr:
                sp["headline_kernel_roof"] = {k_: rr[k_] for k_ in ("bound", "kernel", "kernel_ms", "achieved", "peak", "unit", "frac", "flops_per_unit",
                                                                      "units_per_launch", "traffic", "traffic_frac_of_hbm_peak", "valu_active_frac") if k_ in rr}
                for k_, v_ in sp["headline_kernel_roof"].items():
                    sp["headline_" + k_] = v_
                sp["note"] = ("top level = the HBM-bound kernel of the path (stand-alone block-tridiagonal SpMV, north_star's >= 60 % target), NOT in the timed "
                              "region; headline_* = the register-resident PCG kernel `value` is measured on: bound by fp32 VALU issue, its HBM traffic is one read "
                              "of the lower block triangle per solve")
        else:
            out["roofline_spmv"] = sp
        if args.spmv_mfma:
            sol.set_option("spmv_mfma", 1)           # config 5's MFMA block-GEMV experiment, same launch shape
            ms_mfma = timed(spmv20, 3, warm=1) / 20
            sol.set_option("spmv_mfma", 0)
            out["spmv_mfma_experiment"] = {"kernel": "bt_spmv_mfma_kernel", "ms": ms_mfma, "achieved": b_sp / (ms_mfma * 1e-3) / 1e9,
                                           "unit": "GB/s", "frac": b_sp / (ms_mfma * 1e-3) / 1e9 / HBM_PEAK_GBS,
                                           "mfma_flops_issued_per_launch": 2 * 16 * 16 * 4 * 12 * Bs * N,
                                           "useful_flops_per_launch": 2 * (3 * N - 2) * 196 * Bs}
        del S_big, x, y

    lean = args.profile_lean
    if extras and rank == 0 and not lean:
        # ---- parity of the workload that was just timed: sampled trajectories against the CPU oracle ----
        sys.path.insert(0, os.path.join(ROOT, "oracle"))
        import oracle as orc
        orc.build()
        lam_h = d_lam.cpu().numpy()
        idx = sorted({0, B // 3, (2 * B) // 3, B - 1})
        checks = []
        for b_ in idx:
            c = fp32_check(orc, d_S[b_].cpu().numpy(), d_P[b_].cpu().numpy(), d_g[b_].cpu().numpy(), lam_h[b_], int(it_host[b_]), N, args.precond)
            c["trajectory"] = int(b_)
            checks.append(c)
        out["parity_sample"] = {"against": "oracle/ (CPU, float64 iterate after the same number of iterations; tolerance max(1e-3, 4 x CPU float32 band))",
                                "checked": len(checks), "all_ok": bool(all(c["ok"] for c in checks)), "samples": checks}

    if extras and not lean:
        # ---- the reference's operating regime: lambda warm-started from the previous SQP / MPC step
        # (include/mpcsim.cuh:186,267,337), so solves leave the loop at different iterations.  Emulated by starting
        # from a perturbed converged solution; the hardware workgroup scheduler rebalances.
        lam_star = torch.zeros(B, 14 * N, device=dev)
        sol.solve(d_S, d_P, d_g, lam_star, pcg_config(pcg_exit_tol=1e-9, pcg_max_iter=4000), args.precond)
        gen = torch.Generator(device=dev).manual_seed(1234 + rank)
        scale = lam_star.abs().amax(dim=1, keepdim=True)
        amp = torch.logspace(-4, -1, B, device=dev)[torch.randperm(B, device=dev, generator=gen)].unsqueeze(1)
        lam_w = lam_star + amp * scale * torch.randn(B, 14 * N, device=dev, generator=gen)

        def warm_solve():
            d_lam.copy_(lam_w)
            run_solve()
        ms_w = timed(warm_solve, 3, warm=1) - timed(lambda: d_lam.copy_(lam_w), 3, warm=1)
        itw = d_it.cpu().numpy().astype(np.int64)
        out["warm_start_run"] = {"lambda0": "converged solution + gaussian noise of relative amplitude 1e-4..1e-1 (log-uniform over the batch)",
                                 "mean_pcg_iters": float(itw.mean()), "min_pcg_iters": int(itw.min()), "max_pcg_iters": int(itw.max()),
                                 "max_iter_exit_rate": float(d_ex.float().mean().item()), "kernel_ms": ms_w,
                                 "pcg_iterations_per_sec": float(itw.sum() / (ms_w * 1e-3)),
                                 "linsolves_per_sec": B / (ms_w * 1e-3)}

    if extras and args.profile_mini and N <= 256:
        # (in-run rocprofv3 child: three launches of each producer kernel at the bench batch, nothing else)
        from mpcgpu_amd import Plant, iiwa
        plant = Plant(device=local_rank)
        xu_h, goals_h, xs_h = iiwa.random_windows(N, B, 2024 + rank)
        f32 = lambda a_: torch.from_numpy(np.ascontiguousarray(a_, np.float32)).to(dev)
        d_xu, d_goal, d_xs = f32(xu_h), f32(goals_h.reshape(B, -1)), f32(xs_h)
        for _ in range(3):
            Gk, Ck, gk, ck = sol.generate_kkt(plant, d_goal, d_xs, d_xu, iiwa.TIMESTEP, iiwa.QD_COST, iiwa.r_cost(N))
            sol.form_schur(Gk, Ck, gk, ck, synth.RHO_INIT, "ss")
            sol.compute_dz(Gk, Ck, gk, d_lam)
        torch.cuda.synchronize()

    if extras and not lean and N <= 256:
        # ---- the same path on REAL IIWA-14 systems, produced on the device like the reference's SQP iteration does
        # (include/pcg/sqp.cuh:190-232): generate_kkt -> form_schur -> PCG, cold and warm-started the way the MPC loop
        # warm-starts (lambda of the previous, slightly different, linear system: include/mpcsim.cuh:186,267,337) ----
        from mpcgpu_amd import Plant, iiwa
        plant = Plant(device=local_rank)
        xu_h, goals_h, xs_h = iiwa.random_windows(N, B, 2024 + rank)
        f32 = lambda a_: torch.from_numpy(np.ascontiguousarray(a_, np.float32)).to(dev)
        d_xu, d_goal, d_xs = f32(xu_h), f32(goals_h.reshape(B, -1)), f32(xs_h)
        rc = iiwa.r_cost(N)
        ms_kkt = timed(lambda: sol.generate_kkt(plant, d_goal, d_xs, d_xu, iiwa.TIMESTEP, iiwa.QD_COST, rc), 3, warm=1)
        Gk, Ck, gk, ck = sol.generate_kkt(plant, d_goal, d_xs, d_xu, iiwa.TIMESTEP, iiwa.QD_COST, rc)
        Gk0 = Gk.clone()
        ms_schur = timed(lambda: (Gk.copy_(Gk0), sol.form_schur(Gk, Ck, gk, ck, synth.RHO_INIT, "ss")), 3, warm=1) - timed(lambda: Gk.copy_(Gk0), 3, warm=1)
        Gk.copy_(Gk0)
        rS, rP, rg = sol.form_schur(Gk, Ck, gk, ck, synth.RHO_INIT, "ss")
        lam_any = torch.randn(B, 14 * N, device=dev)
        ms_dz = timed(lambda: sol.compute_dz(Gk, Ck, gk, lam_any), 5, warm=1)
        # per-kernel rooflines of the producers (VERDICT r03 #1a): algorithmic bytes / flops per knot (PRODUCER_MODEL) over this run's times
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
        prod["generate_kkt"]["bound"] = "fp64 VALU issue: ~6,000 instructions per wavefront of four knots at 4 clocks each = 0.32 ms (its HBM floor is 0.04 ms)"
        prod["form_schur"]["kernels"] = "schur_walk_kernel + schur_seam_kernel (chunk length %d)" % sol.get_option("last_schur_chunk")
        prod["form_schur"]["bound"] = ("hbm: its output pattern alone (8,192 row streams of the bd layout, no arithmetic) takes 0.20 ms = 3.7 TB/s on this chip, "
                                       "~0.28 ms with the input reads (profiles/r04_walk_store_side.txt); VALU floor 0.27 ms (unfused DPP multiplies, bit-exact with the oracle)")
        prod["form_schur"]["frac_of_its_store_pattern_ceiling_3.7TBs"] = prod["form_schur"]["achieved"] / 3700.0
        # previous SQP iterate = this one plus a small change of the trajectory -> its multipliers are the warm start
        d_xu_prev = d_xu + 2e-3 * torch.randn_like(d_xu)
        d_xu_prev[:, :14] = d_xu[:, :14]
        Gp, Cp, gp, cp = sol.generate_kkt(plant, d_goal, d_xs, d_xu_prev, iiwa.TIMESTEP, iiwa.QD_COST, rc)
        pS, _, pg = sol.form_schur(Gp, Cp, gp, cp, synth.RHO_INIT, "none")
        lam_prev = sol.block_solve(pS, pg)
        gn = rg.double().norm(dim=1)
        resid = lambda l_: ((rg - sol.bt_spmv(rS, l_)).double().norm(dim=1) / gn)
        res = {}
        for name, lam0 in (("cold", torch.zeros(B, 14 * N, device=dev)), ("warm", lam_prev)):
            l_ = lam0.clone()
            r0 = resid(l_)

            def go():
                l_.copy_(lam0)
                sol.solve(rS, rP, rg, l_, cfg, "ss", iters=d_it, exits=d_ex)
            ms_copy = timed(lambda: l_.copy_(lam0), 3, warm=1)
            ms_ = timed(go, 3, warm=1) - ms_copy               # (l_ now holds the solution of the last run)
            iti = d_it.cpu().numpy().astype(np.int64)
            r1 = resid(l_)
            res[name] = {"mean_pcg_iters": float(iti.mean()), "min_pcg_iters": int(iti.min()), "max_pcg_iters": int(iti.max()),
                         "max_iter_exit_rate": float(d_ex.float().mean().item()), "kernel_ms": ms_,
                         "pcg_iterations_per_sec": float(iti.sum() / (ms_ * 1e-3)), "linsolves_per_sec": B / (ms_ * 1e-3),
                         "true_rel_residual_before_median": float(r0.median().item()), "true_rel_residual_after_median": float(r1.median().item()),
                         "true_rel_residual_after_p90": float(torch.quantile(r1, 0.9).item()),
                         "true_rel_residual_after_max": float(r1.max().item()),
                         "trajectories_whose_true_residual_grew": int((r1 > r0 * (1 + 1e-6)).sum().item())}
        # the dispatch-order hint predicts from the PREVIOUS call's iteration counts, and these timed repetitions replay the same solve, so the
        # prediction is exact here; a real MPC loop's is approximate: both ends are reported
        sol.set_option("sched_hint", 0)
        l_nh = lam_prev.clone()
        ms_nh = timed(lambda: (l_nh.copy_(lam_prev), sol.solve(rS, rP, rg, l_nh, cfg, "ss", iters=d_it, exits=d_ex)), 3, warm=1) - timed(lambda: l_nh.copy_(lam_prev), 3, warm=1)
        sol.set_option("sched_hint", 1)
        res["warm"]["sched_hint"] = "on, with an exact prediction (the timed repetitions replay one solve)"
        res["warm"]["kernel_ms_sched_hint_off"] = ms_nh
        res["warm"]["linsolves_per_sec_sched_hint_off"] = B / (ms_nh * 1e-3)
        out["iiwa_run"] = {"inputs": f"{B} windows of the reference trajectory examples/trajfiles/0_0_traj.csv (first 400 rows), random offset, goals 0..8 steps ahead, "
                                     "state / iterate noise <= 0.05; KKT blocks by mpcg_generate_kkt (IIWA-14 dynamics on the device), Schur by mpcg_form_schur, rho = 1e-3",
                           "pcg": {"max_iter": max_iter, "exit_tol": args.exit_tol, "precond": "ss"}, "generate_kkt_ms": ms_kkt, "form_schur_ms": ms_schur, "compute_dz_ms": ms_dz,
                           "cold_start": res["cold"], "warm_start_from_previous_sqp_iterate": res["warm"],
                           "note": "cond(-S) of these systems is 1e7-2e7 at N=128 (synthetic generator: ~1e5): from lambda0 = 0 most solves hit the reference's iteration "
                                   "cap, which presupposes the MPC loop's warm starts (tests/make_iiwa_golden.py prints the study)"}
        # one whole linear-system step of an SQP iteration (include/pcg/sqp.cuh:190-259: KKT blocks, Schur system, PCG from the previous
        # multipliers, dz) as ONE hipGraph replay: what a caller that keeps the loop on the device pays per iteration and batch
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
            sqp_step = {"what": "generate_kkt -> form_schur (ss) -> PCG warm-started from the previous iterate's multipliers -> compute_dz, one hipGraph replay",
                        "ms_per_batch": ms_step, "batch": B, "sqp_linear_steps_per_sec": B / (ms_step * 1e-3), "us_per_trajectory_step": ms_step * 1e3 / B,
                        "mean_pcg_iters": float(it_step.mean()), "dz_finite": bool(torch.isfinite(dz_g).all().item())}
            del gr
            # (tried this round: the batch cut into 2 / 4 independent parts on as many streams inside one graph, so that one part's PCG tail is filled
            #  with another part's producers — 1.20 / 1.24 ms against 1.15 ms for the single chain: the parts' kernels do not overlap usefully)
        except Exception as e_:                                  # (reported, never fatal for the headline measurement)
            sqp_step = {"error": repr(e_)}
        out["iiwa_run"]["sqp_linear_step_graph"] = sqp_step
        out["roofline_producers"] = prod
        w_ = res["warm"]
        out["config"].update({"iiwa_warm_mean_pcg_iters": w_["mean_pcg_iters"], "iiwa_warm_max_iter_exit_rate": w_["max_iter_exit_rate"],
                              "iiwa_warm_linsolves_per_sec": w_["linsolves_per_sec"], "iiwa_warm_pcg_iterations_per_sec": w_["pcg_iterations_per_sec"],
                              "iiwa_warm_true_residual_median": w_["true_rel_residual_after_median"], "iiwa_warm_true_residual_p90": w_["true_rel_residual_after_p90"],
                              "iiwa_warm_true_residual_max": w_["true_rel_residual_after_max"],
                              "iiwa_warm_trajectories_whose_residual_grew": w_["trajectories_whose_true_residual_grew"],
                              "iiwa_warm_linsolves_per_sec_sched_hint_off": w_["linsolves_per_sec_sched_hint_off"],
                              "iiwa_generate_kkt_ms": ms_kkt, "iiwa_form_schur_ms": ms_schur, "iiwa_warm_pcg_ms": w_["kernel_ms"], "iiwa_compute_dz_ms": ms_dz,
                              "iiwa_sqp_linear_step_ms": sqp_step.get("ms_per_batch"),
                              "iiwa_regime": "real IIWA-14 systems made on the device (mpcg_generate_kkt -> mpcg_form_schur), lambda warm-started from the previous SQP iterate: "
                                             "the regime the reference's iteration caps presuppose; the headline `value` is the cold-start synthetic batch"})
        del Gk, Ck, Gp, Cp, rS, rP, pS

    if extras and rank == 0 and world == 1 and not lean:
        # the short-horizon regime in throughput mode (N = 32, the reference's real-time horizon; row-per-lane kernel)
        sh = {}
        for pc_ in ("ss", "jacobi"):
            Ns, Bs = 32, 2048
            ss_ = PcgSolver(Ns, max_batch=Bs, device=local_rank)
            S0, P0, g0 = build_inputs(ss_, Ns, 64, seed0, pc_, dev, chunk=64)
            Sl, Pl, gl = (t.repeat(Bs // 64, 1).contiguous() for t in (S0, P0, g0))
            cs = pcg_config(pcg_exit_tol=0.0, pcg_max_iter=synth.pcg_max_iter(Ns))
            ls = torch.zeros(Bs, 14 * Ns, device=dev)
            is_ = torch.zeros(Bs, dtype=torch.int32, device=dev)
            xs_ = torch.zeros(Bs, dtype=torch.uint8, device=dev)

            def run_s():
                ls.zero_()
                ss_.solve(Sl, Pl, gl, ls, cs, pc_, iters=is_, exits=xs_)
            ms_s = timed(run_s, 5, warm=1)
            sh[pc_] = {"knots": Ns, "batch": Bs, "pcg_iters_per_solve": synth.pcg_max_iter(Ns), "kernel_ms": ms_s,
                       "pcg_iterations_per_sec": int(is_.sum().item()) / (ms_s * 1e-3), "kernel_family": ss_.get_option("last_kernel_family"),
                       "kernel_waves": ss_.get_option("last_kernel_waves")}
            del Sl, Pl, gl, ls, ss_
        out["short_horizon"] = sh
        out["config2_latency"] = latency_config2(dev)
        out["batch1_sqp_step_latency"] = batch1_sqp_step_latency(dev, args.exit_tol)
        # the headline horizon as ONE trajectory (the reference's own mode of use): ms per SQP-linsolve
        sol1 = PcgSolver(N, max_batch=1, device=local_rank)
        l1 = torch.zeros(1, 14 * N, device=dev)
        i1 = torch.zeros(1, dtype=torch.int32, device=dev)
        x1 = torch.zeros(1, dtype=torch.uint8, device=dev)

        def one():
            l1.zero_()
            sol1.solve(d_S[:1], d_P[:1], d_g[:1], l1, cfg, args.precond, iters=i1, exits=x1)
        ms1 = timed(one, 25, warm=5)
        out["single_trajectory_latency"] = {"workload": f"N={N}, {args.precond}, ONE trajectory, max_iter={max_iter}",
                                            "pcg_iters": int(i1.item()), "ms_per_linsolve": ms1,
                                            "us_per_pcg_iter": ms1 * 1e3 / max(int(i1.item()), 1),
                                            "kernel_family": sol1.get_option("last_kernel_family"), "kernel_waves": sol1.get_option("last_kernel_waves")}

    if extras and rank == 0 and world == 1 and not args.profile_mini:
        # horizons one CU cannot hold (BASELINE config 5's N = 512, and N = 256): the clustered lane-pair kernel, fixed
        # iteration counts = the reference's caps (settings.cuh:123-139); 256 resident systems tiled to the batch
        lh = {}
        for Nl in ((512,) if lean else (256, 512)):          # (--profile-lean: only the streaming leg below)
            sl = PcgSolver(Nl, max_batch=B, device=local_rank)
            S0, P0, g0 = build_inputs(sl, Nl, 32, seed0, "ss", dev, chunk=32)
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
                lh[f"N{Nl}"] = {"pcg_iters_per_solve": synth.pcg_max_iter(Nl), "batch": B, "kernel_ms": ms_b, "pcg_iterations_per_sec": its / (ms_b * 1e-3),
                                "ms_one_trajectory": ms_1, "us_per_pcg_iter_one_trajectory": ms_1 * 1e3 / synth.pcg_max_iter(Nl),
                                "kernel_family": sl.get_option("last_kernel_family"), "members_per_trajectory": sl.get_option("last_kernel_cluster"),
                                "cluster_fixups": sl.get_option("cluster_fixups")}
            if Nl == 512:
                # ---- the PCG solve as an HBM stream: nothing resident (pcg_traj_kernel<16,0,2>), every block re-read every iteration.
                # One workgroup per CU (115 KB of LDS at N=512), so the LIVE matrices are 256 x 2.41 MB = 616 MB >> the 256 MiB Infinity Cache
                # whatever the batch: the HBM model of SURVEY §8d (2,577,344 B per trajectory-iteration) applies as written.  (Round 2 ran
                # this leg at N=128, where the live set of 154-308 MB sat in the Infinity Cache: frac 1.06, not an HBM measurement.)
                ss = PcgSolver(Nl, max_batch=B, device=local_rank)
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
                out["roofline_pcg_streaming"] = {
                    "bound": "hbm" if ach <= HBM_PEAK_GBS else "infinity cache + hbm (frac > 1: NOT an HBM roofline fraction)",
                    "kernel": "pcg_traj_kernel<16,0,2> (no resident rows)", "knot_points": Nl, "batch": B, "kernel_ms": ms_s, "achieved": ach,
                    "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS, "bytes_per_unit": b_it, "units_per_launch": its_s,
                    "resident_trajectories": occ, "live_matrix_set_mb": occ * 2 * 3 * 196 * Nl * 4 / 1e6, "infinity_cache_mb": 268.4,
                    "frac_of_measured_copy_ceiling_6.29TBs": ach / 6290.0, "pcg_iterations_per_sec": its_s / (ms_s * 1e-3), "traffic": None}
                tr, src = load_traffic(f"pcg_traj_kernel<16,0,2>|N{Nl}_B{B}_ss_it{synth.pcg_max_iter(Nl)}_tol0")
                if tr:
                    out["roofline_pcg_streaming"]["traffic"] = tr["hbm_traffic_bytes_per_launch"]
                    out["roofline_pcg_streaming"]["traffic_source"] = src
                del ss
            del Sl, Pl, gl, ll, sl
        if lh:
            out["long_horizon"] = lh
            # ---- linsys_t = double (USE_DOUBLES of the reference): the bench workload's own systems in double, fixed 40 iterations.  N <= 32 would run the
            # register-resident row-per-lane kernel; N = 128 runs the streaming kernel, which reads TWO block columns once the symmetry latch allows
            try:
                Bd = min(B, 1024)
                S64, P64, g64 = torch.nan_to_num(d_S[:Bd]).double(), torch.nan_to_num(d_P[:Bd]).double(), d_g[:Bd].double()
                l64 = torch.zeros(Bd, 14 * N, dtype=torch.float64, device=dev)
                i64 = torch.zeros(Bd, dtype=torch.int32, device=dev); x64 = torch.zeros(Bd, dtype=torch.uint8, device=dev)
                c64 = pcg_config(pcg_exit_tol=0.0, pcg_max_iter=40)
                sol.solve_f64(S64, P64, g64, l64, c64, args.precond, iters=i64, exits=x64)          # (first call: the latch's one blocking check)
                ms64 = timed(lambda: (l64.zero_(), sol.solve_f64(S64, P64, g64, l64, c64, args.precond, iters=i64, exits=x64)), 3, warm=1) - timed(lambda: l64.zero_(), 3, warm=1)
                cols64 = 2 if sol.get_option("symmetry_state") == 1 else 3
                by64 = 2 * cols64 * 196 * N * 8
                out["double_precision"] = {"knot_points": N, "batch": Bd, "pcg_iters_per_solve": 40, "kernel_ms": ms64, "pcg_iterations_per_sec": Bd * 40 / (ms64 * 1e-3),
                                           "kernel_family": sol.get_option("last_kernel_family"), "block_columns_read": cols64, "bytes_per_unit": by64,
                                           "achieved": Bd * 40 * by64 / (ms64 * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": Bd * 40 * by64 / (ms64 * 1e-3) / 1e9 / HBM_PEAK_GBS,
                                           "bound": "hbm (S and Pinv re-read every iteration: a double N=128 trajectory is 1.2 MB, 2.4x the register file)"}
                del S64, P64, g64, l64
            except Exception as e_:
                out["double_precision"] = {"error": repr(e_)}
            # ---- the clustered kernel against the floor of ITS design ("break 0.15 or prove the floor") ----
            # Classic PCG needs two cluster-wide reductions per iteration, each behind a matrix pass, and nothing of the next half can start
            # before the reduced inner product is known.  Phase stamps of one iteration (profiles/r04_lpkc_phases.txt, N = 256, shader clocks
            # at 2.38 GHz): matrix pass 2,090 | last partial published -> the polling wavefront has every granule of the epoch (one L2 hand-off)
            # 635 at best, 1,275 at worst | partials folded, halo knots in LDS 460 | barrier 140 | operand rebuild + alpha / beta chain 360.
            floor_ticks = 2 * (2090 + 635 + 140)
            lh_roof = {"what": "clustered lane-pair kernel vs the floor of classic PCG on its decomposition: 2 x (matrix pass + one L2 hand-off + barrier) per iteration",
                       "floor_us_per_iteration": floor_ticks / 2380.0, "floor_terms_shader_clocks": {"matrix_pass": 2090, "l2_handoff_min": 635, "barrier": 140},
                       "above_the_floor_shader_clocks_per_half": {"fold_partials_and_halo_into_lds": 460, "operand_rebuild_and_scalar_chain": 360, "handoff_jitter_up_to": 640},
                       "source": "profiles/r04_lpkc_phases.txt (tools/_prof/lpkc_phases.py, -DMPCG_PROF build, this round)",
                       "not_built": "a single-reduction (Chronopoulos-Gear) PCG as an opt-in variant: one all-reduce + one neighbour halo hand-off per iteration instead of two "
                                    "all-reduces — costed at +12 % (halo not overlapped) to +30 % (overlapped); its fixed-K iterates drift 5-10x more than the classic recurrence "
                                    "(outside the parity band of DESIGN.md, numpy float32 experiment of round 3), so it could only ever be opt-in"}
            for key_, v_ in lh.items():
                G_ = v_["members_per_trajectory"]
                ncu_ = sol.get_option("num_cus")
                resident = 8 * ((ncu_ // 8) // G_) if ncu_ >= 8 and ncu_ // 8 >= G_ else ncu_ // G_
                us_it = min(resident, B) / (v_["pcg_iterations_per_sec"] * 1e-6)          # one cluster's time per iteration with the chip full of clusters
                lh_roof[key_] = {"pcg_iterations_per_sec": v_["pcg_iterations_per_sec"], "resident_clusters": resident, "us_per_iteration_full_batch": us_it,
                                 "us_per_iteration_one_trajectory": v_["us_per_pcg_iter_one_trajectory"],
                                 "frac_of_floor_full_batch": lh_roof["floor_us_per_iteration"] / us_it,
                                 "frac_of_floor_one_trajectory": lh_roof["floor_us_per_iteration"] / v_["us_per_pcg_iter_one_trajectory"]}
            out["roofline_long_horizon"] = lh_roof

    if extras and rank == 0 and world == 1 and not lean and args.scaling == "weak":
        # ---- what scaling to expect (measured here on one GPU, the multi-GPU curve itself is the driver's to measure) ----
        # weak (default): every rank runs this very workload, no data-path collective: N x value(1) minus launch jitter.
        # strong (BASELINE config 4 as written: 1024 trajectories over 8 GPUs = 128 each): a GPU holds one workgroup per trajectory, so 128
        # trajectories occupy 128 of 256 CUs ONCE — the step takes the batch-128 time below, not 1/8 of the batch-1024 time.
        Bq = max(1, B // 8)
        ms_q = timed(lambda: (d_lam[:Bq].zero_(), sol.solve(d_S[:Bq], d_P[:Bq], d_g[:Bq], d_lam[:Bq], cfg, args.precond, iters=d_it[:Bq], exits=d_ex[:Bq])), 7, warm=2)
        ms_f = timed(lambda: (d_lam.zero_(), run_solve()), 5, warm=1)
        out["scaling_expectation"] = {
            "weak": {"ceiling": "N x value(N=1): ranks are independent (one all-gather of 5 B per trajectory after the timed step)", "efficiency_expected": 1.0},
            "strong": {"global_batch": B, "batch_per_gpu_at_8": Bq, "ms_step_full_batch_1gpu": ms_f, "ms_step_at_batch_per_gpu_at_8": ms_q,
                       "speedup_ceiling_at_8_gpus": ms_f / ms_q, "efficiency_ceiling_at_8_gpus": ms_f / ms_q / 8.0,
                       "why": f"{Bq} trajectories = {Bq} workgroups on {sol.get_option('num_cus')} CUs: the step costs one trajectory's latency however few CUs are busy; "
                              "the chip is filled only from one trajectory per CU upwards"}}
        out["config"]["strong_scaling_speedup_ceiling_at_8_gpus"] = ms_f / ms_q

    if extras and rank == 0 and not lean:
        # the other selectable solver on the same resident systems: batched block-tridiagonal direct solve
        # (GPU counterpart of the reference's QDLDL path, i.e. of what cpu_baseline times on the host)
        lam_d = torch.empty(B, 14 * N, device=dev)
        ms_d = timed(lambda: sol.block_solve(d_S, d_g, lam_d), 5, warm=1)
        nb = min(4, B)
        Sd = d_S[:nb].cpu().numpy()
        gd = d_g[:nb].cpu().numpy()
        ld = lam_d[:nb].cpu().numpy().astype(np.float64)
        res = [float(np.linalg.norm(gd[b] - synth.bd_to_dense(np.nan_to_num(Sd[b]), N) @ ld[b]) / np.linalg.norm(gd[b])) for b in range(nb)]
        out["block_solve"] = {"kernel": "bt_block_solve_kernel (mpcg_block_solve)", "ms_per_batch": ms_d, "batch": B,
                              "linsolves_per_sec": B / (ms_d * 1e-3), "us_per_linsolve_throughput": ms_d * 1e3 / B,
                              "true_rel_residual_sample": res,
                              "note": "fp32 block LU sweep; not the headline metric (which counts PCG iterations)"}

    if rank == 0 and world == 1 and not args.no_cpu_baseline and not lean:
        out["cpu_baseline"] = cpu_baseline(N, S_h, g_h, float(it_host.mean()), args.cpu_seconds)
        out["cpu_baseline"]["gpu_linsolves_per_sec"] = out["linsolves_per_sec"]

    # ---- in-run HBM traffic (rocprofv3 --pmc child passes) and the final shape of the objects the driver keeps.  The driver's record keeps the
    # first ~24 scalar keys of `roofline`, `config` and `cpu_baseline` (nested objects and long strings are cut): what must survive comes first.
    if rank == 0:
        pmc, pmc_note = (None, "not requested")
        if world == 1 and extras and not lean and not args.no_inrun_pmc:
            tail = ["--knots", str(N), "--batch", str(args.batch), "--precond", args.precond, "--exit-tol", repr(args.exit_tol), "--spmv-batch", str(args.spmv_batch)]
            if args.max_iter:
                tail += ["--max-iter", str(args.max_iter)]
            pmc, pmc_note = inrun_pmc(tail)
        sp = out.get("roofline_spmv") or (out.get("roofline") if out.get("roofline", {}).get("kernel") == "bt_spmv_kernel" else None)
        rr = out.get("roofline_resident")
        prod = out.get("roofline_producers", {})
        if pmc:
            for obj, needle in ((sp, "bt_spmv_kernel"), (rr, fam), (prod.get("generate_kkt"), "generate_kkt_kernel"), (prod.get("compute_dz"), "compute_dz_dpp_kernel")):
                e = find_traffic(pmc, needle)
                if obj is not None and e:
                    obj["traffic"] = e["hbm_traffic_bytes_per_launch"]
                    obj["traffic_is"] = pmc_note
                    obj.pop("traffic_source", None)           # (the committed profile's figure has just been replaced by this run's)
                    obj["traffic_over_algorithmic"] = e["hbm_traffic_bytes_per_launch"] / (obj.get("algorithmic_bytes_per_launch") or obj.get("hbm_algorithmic_bytes_per_launch")
                                                                                         or obj["bytes_per_unit"] * obj["units_per_launch"])
            if "form_schur" in prod:
                ew, es = find_traffic(pmc, "schur_walk_kernel"), find_traffic(pmc, "schur_seam_kernel")
                if ew:
                    tb = ew["hbm_traffic_bytes_per_launch"] + (es["hbm_traffic_bytes_per_launch"] if es else 0.0)
                    prod["form_schur"].update({"traffic": tb, "traffic_is": pmc_note, "traffic_over_algorithmic": tb / (prod["form_schur"]["bytes_per_unit"] * prod["form_schur"]["units_per_launch"])})
            out["inrun_pmc_kernels"] = pmc
        if sp is not None and rr is not None and "roofline" in out and out["roofline"] is sp:
            g_ = lambda d_, k_: d_.get(k_) if d_ else None
            ordered = {"bound": sp["bound"], "kernel": sp["kernel"], "achieved": sp["achieved"], "peak": sp["peak"], "unit": sp["unit"], "frac": sp["frac"],
                       "traffic": sp.get("traffic"), "kernel_ms": sp["kernel_ms"],
                       "read_ceiling_gbs_this_run": sp.get("read_ceiling_gbs_this_run"), "frac_of_read_ceiling_this_run": sp.get("frac_of_read_ceiling_this_run"),
                       "headline_kernel": rr["kernel"], "headline_bound": rr["bound"], "headline_frac": rr["frac"], "headline_achieved": rr["achieved"],
                       "headline_peak": rr["peak"], "headline_unit": rr["unit"], "headline_kernel_ms": rr["kernel_ms"],
                       "headline_valu_active_frac": rr.get("valu_active_frac"), "headline_traffic": rr.get("traffic"),
                       "form_schur_frac": g_(prod.get("form_schur"), "frac"), "compute_dz_frac": g_(prod.get("compute_dz"), "frac"),
                       "generate_kkt_frac_of_fp64_valu_peak": g_(prod.get("generate_kkt"), "frac_of_valu_peak"),
                       "traffic_is": sp.get("traffic_is") or pmc_note, "d2d_copy_gbs_this_run": sp.get("d2d_copy_gbs_this_run"),
                       "form_schur_traffic_over_algorithmic": g_(prod.get("form_schur"), "traffic_over_algorithmic")}
            ordered["note"] = ("top level = the HBM-bound kernel of the path (stand-alone block-tridiagonal SpMV, north_star's >= 60 % target; NOT in the timed region); "
                               "headline_* = the register-resident PCG kernel `value` is measured on (fp32 VALU bound; HBM sees one read of the lower block triangle per solve); "
                               "the producers' full objects: roofline_producers")
            for k_, v_ in sp.items():
                ordered.setdefault(k_, v_)
            out["roofline"] = ordered
        cfg_o = out["config"]
        first = {"workload": cfg_o["workload"], "knot_points": N, "batch_per_gpu": B, "global_batch": B_global, "precond": args.precond, "pcg_max_iter": max_iter,
                 "pcg_exit_tol": args.exit_tol, "parallelism": cfg_o["parallelism"], "kernel_family": fam,
                 "sustained_value": sustained["value"] if sustained else None, "sustained_seconds": sustained["seconds"] if sustained else None,
                 "sustained_sclk_mhz_median": (sustained["sclk_mhz"] or {}).get("median") if sustained else None}
        for k_ in ("iiwa_sqp_linear_step_ms", "iiwa_generate_kkt_ms", "iiwa_form_schur_ms", "iiwa_warm_pcg_ms", "iiwa_compute_dz_ms", "iiwa_warm_mean_pcg_iters",
                   "iiwa_warm_linsolves_per_sec", "iiwa_warm_linsolves_per_sec_sched_hint_off", "iiwa_warm_true_residual_median"):
            if k_ in cfg_o:
                first[k_] = cfg_o[k_]
        b1 = out.get("batch1_sqp_step_latency", {})
        for n_ in (32, 128):
            if "us_per_step" in b1.get(f"N{n_}", {}):
                first[f"batch1_sqp_step_us_N{n_}"] = b1[f"N{n_}"]["us_per_step"]
        if "strong_scaling_speedup_ceiling_at_8_gpus" in cfg_o:
            first["strong_scaling_speedup_ceiling_at_8_gpus"] = cfg_o["strong_scaling_speedup_ceiling_at_8_gpus"]
        for k_, v_ in cfg_o.items():
            first.setdefault(k_, v_)
        out["config"] = first
        if sustained:
            out["sustained"] = sustained

    # the JSON line goes out LAST: RCCL writes a version banner through C stdio when the communicator goes away
    if torch.distributed.is_initialized():
        torch.distributed.destroy_process_group()
    try:
        import ctypes
        ctypes.CDLL(None).fflush(None)
    except OSError:
        pass
    if rank == 0:
        print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
