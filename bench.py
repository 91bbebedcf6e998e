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

Prints ONE JSON line on rank 0: value = PCG iterations / second over all GPUs, plus `roofline`
(HBM, algorithmic bytes of the dominant kernel / its HIP-event duration) and, at N=1,
`cpu_baseline` (the reference's QDLDL CPU path, restated in oracle/, timed on this host).
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

HBM_PEAK_GBS = 8000.0   # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec


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
        "host_cpus": os.cpu_count(),
    }


P_HOST = None


def latency_config2(dev, reps=200):
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
            "pcg_waves": sol.get_option("pcg_waves"), "pcg_reg_rows": sol.get_option("pcg_reg_rows")}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--knots", type=int, default=128)
    ap.add_argument("--batch", type=int, default=1024, help="trajectories PER GPU (weak scaling)")
    ap.add_argument("--precond", default="ss", choices=["ss", "jacobi"])
    ap.add_argument("--exit-tol", type=float, default=1e-4)
    ap.add_argument("--max-iter", type=int, default=0, help="0 = reference table (settings.cuh:123-139)")
    ap.add_argument("--pcg-waves", type=int, default=0)
    ap.add_argument("--nt", type=int, default=-1)
    ap.add_argument("--reg-rows", type=int, default=-1)
    ap.add_argument("--lds-rows", type=int, default=-2)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    ap.add_argument("--spmv", action="store_true", help="also time the stand-alone block-tridiagonal SpMV")
    ap.add_argument("--warm", action="store_true", help="also run the workload from warm starts (mixed iteration counts)")
    ap.add_argument("--latency", action="store_true", help="also time BASELINE config 2 (N=32, one trajectory)")
    ap.add_argument("--storage", default="f32", choices=["f32", "f16"],
                    help="matrix storage of S/Pinv; f16 = BASELINE config 5's reduced-precision experiment (arithmetic stays fp32)")
    args = ap.parse_args()

    rank, local_rank, world = D.init()
    if world != args.gpus:
        if rank == 0:
            print(f"warning: --gpus {args.gpus} but WORLD_SIZE={world}; using WORLD_SIZE", file=sys.stderr)
    assert torch.cuda.is_available(), "bench.py needs a HIP device (the product has no CPU path)"
    if os.environ.get("MPCG_FORCE_DEVICE"):          # dry-run of the N>1 flow on a 1-GPU box (with MPCG_DIST_BACKEND=gloo)
        local_rank = int(os.environ["MPCG_FORCE_DEVICE"])
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)

    N, B = args.knots, args.batch
    max_iter = args.max_iter or synth.pcg_max_iter(N)
    cfg = pcg_config(pcg_exit_tol=args.exit_tol, pcg_max_iter=max_iter)

    global P_HOST
    sol = PcgSolver(N, max_batch=B, device=local_rank)
    d_S, d_P, d_g = build_inputs(sol, N, B, rank * B, args.precond, dev)
    ns = min(32, B)
    S_h, P_h, g_h = (t[:ns].cpu().numpy() for t in (d_S, d_P, d_g))     # host copies of the CPU-baseline sample
    P_HOST = P_h
    d_lam = torch.zeros(B, 14 * N, device=dev)
    d_it = torch.zeros(B, dtype=torch.int32, device=dev)
    d_ex = torch.zeros(B, dtype=torch.uint8, device=dev)
    if args.pcg_waves:
        sol.set_option("pcg_waves", args.pcg_waves)
    if args.nt >= 0:
        sol.set_option("nt_loads", args.nt)
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

    def step():
        d_lam.zero_()                       # every step is the same cold-start solve
        run_solve()

    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()

    # --- timed region: exactly K steps, barrier + synchronize on both sides, max over ranks ---
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]
    D.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(args.steps):
        d_lam.zero_()
        ev[i][0].record()                   # HIP events on the stream the kernel is launched on
        run_solve()
        ev[i][1].record()
    torch.cuda.synchronize()
    D.barrier()
    t_local = time.perf_counter() - t0
    t_all = D.max_over_ranks(t_local, dev)

    it_host = d_it.cpu().numpy().astype(np.int64)
    ex_host = d_ex.cpu().numpy()
    iters_step_local = int(it_host.sum())
    iters_step_all = D.sum_over_ranks(iters_step_local, dev)
    kern_ms = float(np.mean([a.elapsed_time(b) for a, b in ev]))
    kern_ms_all = D.max_over_ranks(kern_ms, dev)

    ms_per_step = 1e3 * t_all / args.steps
    value = iters_step_all / (t_all / args.steps)
    bytes_iter = synth.algorithmic_bytes(N, precond=args.precond)["pcg_iter"]     # fp32-storage model
    achieved = iters_step_local * bytes_iter / (kern_ms * 1e-3) / 1e9     # GB/s, this rank's kernel

    out = {
        "metric": "pcg_iterations_per_sec", "value": value, "unit": "iter/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms_per_step,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32" if args.storage == "f32" else "f32 arithmetic, f16 matrix storage",
        "data": "synthetic",
        "config": {"workload": f"IIWA-14 (n=14) N={N} knots, {args.precond} preconditioner, batch {B} trajectories/GPU "
                               f"(BASELINE config 4's batch is 1024; S+Pinv = {2 * B * 3 * 196 * N * 4 / 1e6:.0f} MB in HBM), lambda0=0, "
                               f"max_iter={max_iter}, exit_tol={args.exit_tol:g}",
                   "knot_points": N, "state_size": 14, "batch_per_gpu": B, "global_batch": B * world,
                   "precond": args.precond, "pcg_max_iter": max_iter, "pcg_exit_tol": args.exit_tol,
                   "parallelism": f"batch-sharded x{world}", "pcg_waves": sol.get_option("pcg_waves"),
                   "pcg_reg_rows": sol.get_option("pcg_reg_rows"), "pcg_lds_rows": sol.get_option("pcg_lds_rows"),
                   "nt_loads": sol.get_option("nt_loads")},
        "ms_per_linsolve": ms_per_step / B,
        "linsolves_per_sec": B * world / (ms_per_step * 1e-3),
        "mean_pcg_iters": float(it_host.mean()), "max_iter_exit_rate": float(ex_host.mean()),
        "resident_trajectories_per_gpu": sol.checkPcgOccupancy(),
        "roofline": {"bound": "hbm", "kernel": "pcg_traj_kernel", "achieved": achieved, "peak": HBM_PEAK_GBS,
                     "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": None,
                     "kernel_ms": kern_ms, "kernel_ms_max_over_ranks": kern_ms_all,
                     "algorithmic_bytes_per_launch": iters_step_local * bytes_iter,
                     "bytes_per_unit": bytes_iter, "units_per_launch": iters_step_local,
                     "unit_of_work": "one PCG iteration of one trajectory"},
    }

    # HBM traffic of this exact workload from the committed PMC passes (bench cannot run rocprofv3 on itself)
    try:
        pre = "pcg" if args.storage == "f32" else "pcg16"
        key = (f"N{N}_B{B}_{args.precond}_it{max_iter}_tol{args.exit_tol:g}_w{sol.get_option(pre + '_waves')}"
               f"_rr{sol.get_option(pre + '_reg_rows')}_rl{sol.get_option(pre + '_lds_rows')}_nt{sol.get_option('nt_loads')}"
               + ("" if args.storage == "f32" else "_f16"))
        tr = json.load(open(os.path.join(ROOT, "profiles", "traffic.json"))).get(key)
        if tr:
            out["roofline"]["traffic"] = tr["hbm_traffic_bytes_per_launch"]
            out["roofline"]["traffic_source"] = tr["source"]
            # the L2-miss bytes over the live kernel time: what the memory side actually sustains
            out["roofline"]["traffic_gbs"] = tr["hbm_traffic_bytes_per_launch"] / (out["roofline"]["kernel_ms"] * 1e-3) / 1e9
            out["roofline"]["traffic_frac"] = out["roofline"]["traffic_gbs"] / HBM_PEAK_GBS
    except (OSError, ValueError):
        pass

    if args.warm and args.storage == "f32":
        # the reference's operating regime: lambda is warm-started from the previous SQP / MPC step
        # (include/mpcsim.cuh:186,267,337), so solves leave the loop at different iterations.  Emulated by
        # starting from a perturbed converged solution; the hardware workgroup scheduler rebalances.
        lam_star = torch.zeros(B, 14 * N, device=dev)
        sol.solve(d_S, d_P, d_g, lam_star, pcg_config(pcg_exit_tol=1e-9, pcg_max_iter=4000), args.precond)
        gen = torch.Generator(device=dev).manual_seed(1234 + rank)
        scale = lam_star.abs().amax(dim=1, keepdim=True)
        amp = torch.logspace(-4, -1, B, device=dev)[torch.randperm(B, device=dev, generator=gen)].unsqueeze(1)
        lam_w = lam_star + amp * scale * torch.randn(B, 14 * N, device=dev, generator=gen)
        ts = []
        for i in range(4):
            d_lam.copy_(lam_w)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            run_solve()
            e1.record()
            torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1))
        itw = d_it.cpu().numpy().astype(np.int64)
        ms_w = float(np.median(ts[1:]))
        out["warm_start_run"] = {"lambda0": "converged solution + gaussian noise of relative amplitude 1e-4..1e-1 (log-uniform over the batch)",
                                 "mean_pcg_iters": float(itw.mean()), "min_pcg_iters": int(itw.min()), "max_pcg_iters": int(itw.max()),
                                 "max_iter_exit_rate": float(d_ex.float().mean().item()), "kernel_ms": ms_w,
                                 "pcg_iterations_per_sec": float(itw.sum() / (ms_w * 1e-3)),
                                 "linsolves_per_sec": B / (ms_w * 1e-3)}

    if args.spmv and args.storage == "f32":
        # the same solve with NOTHING resident (every block row re-read every iteration): the variant that sits
        # on the HBM roofline, reported next to the default so that frac > 1 above can be read for what it is
        saved = {k: sol.get_option(k) for k in ("pcg_waves", "pcg_reg_rows", "pcg_lds_rows")}
        sol.set_option("pcg_waves", 16); sol.set_option("pcg_reg_rows", 0); sol.set_option("pcg_lds_rows", 0)
        ts = []
        for i in range(4):
            d_lam.zero_()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            run_solve()
            e1.record()
            torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1))
        ms_s = float(np.median(ts[1:]))
        its_s = int(d_it.sum().item())
        out["streaming_variant"] = {"kernel": "pcg_traj_kernel<16,0,2> (no resident rows)", "kernel_ms": ms_s,
                                    "achieved": its_s * bytes_iter / (ms_s * 1e-3) / 1e9, "unit": "GB/s",
                                    "frac": its_s * bytes_iter / (ms_s * 1e-3) / 1e9 / HBM_PEAK_GBS,
                                    "pcg_iterations_per_sec": its_s / (ms_s * 1e-3)}
        for k, v in saved.items():
            sol.set_option(k, v)

    if args.spmv:
        x = torch.randn(B, 14 * N, device=dev)
        y = torch.empty_like(x)
        for _ in range(3):
            sol.bt_spmv(d_S, x, y)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        reps = 20
        torch.cuda.synchronize()
        e0.record()
        for _ in range(reps):
            sol.bt_spmv(d_S, x, y)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / reps
        b = synth.algorithmic_bytes(N)["spmv"] * B
        sol.set_option("spmv_mfma", 1)           # config 5's MFMA block-GEMV experiment, same launch shape
        for _ in range(3):
            sol.bt_spmv(d_S, x, y)
        torch.cuda.synchronize()
        e0.record()
        for _ in range(reps):
            sol.bt_spmv(d_S, x, y)
        e1.record()
        torch.cuda.synchronize()
        sol.set_option("spmv_mfma", 0)
        ms_mfma = e0.elapsed_time(e1) / reps
        out["spmv_mfma_experiment"] = {"kernel": "bt_spmv_mfma_kernel", "ms": ms_mfma, "achieved": b / (ms_mfma * 1e-3) / 1e9,
                                       "unit": "GB/s", "frac": b / (ms_mfma * 1e-3) / 1e9 / HBM_PEAK_GBS,
                                       "mfma_flops_issued_per_launch": 2 * 16 * 16 * 4 * 12 * B * N,
                                       "useful_flops_per_launch": 2 * (3 * N - 2) * 196 * B}
        out["spmv"] = {"kernel": "bt_spmv_kernel", "ms": ms, "achieved": b / (ms * 1e-3) / 1e9, "unit": "GB/s",
                       "frac": b / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, "algorithmic_bytes_per_launch": b,
                       "trajectory_spmv_per_sec": B / (ms * 1e-3)}

    if rank == 0 and world == 1 and args.latency:
        out["config2_latency"] = latency_config2(dev)
        # the headline horizon as ONE trajectory (the reference's own mode of use): ms per SQP-linsolve
        sol1 = PcgSolver(N, max_batch=1, device=local_rank)
        l1 = torch.zeros(1, 14 * N, device=dev)
        i1 = torch.zeros(1, dtype=torch.int32, device=dev)
        x1 = torch.zeros(1, dtype=torch.uint8, device=dev)
        ts = []
        for i in range(30):
            l1.zero_()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            sol1.solve(d_S[:1], d_P[:1], d_g[:1], l1, cfg, args.precond, iters=i1, exits=x1)
            e1.record()
            torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1))
        out["single_trajectory_latency"] = {"workload": f"N={N}, {args.precond}, ONE trajectory, max_iter={max_iter}",
                                            "pcg_iters": int(i1.item()), "ms_per_linsolve": float(np.median(ts[5:])),
                                            "us_per_pcg_iter": float(np.median(ts[5:])) * 1e3 / max(int(i1.item()), 1),
                                            "pcg_waves": sol1.get_option("pcg_waves"), "pcg_reg_rows": sol1.get_option("pcg_reg_rows")}

    if rank == 0 and args.storage == "f32":
        # the other selectable solver on the same resident systems: batched block-tridiagonal direct solve
        # (GPU counterpart of the reference's QDLDL path, i.e. of what cpu_baseline times on the host)
        lam_d = torch.empty(B, 14 * N, device=dev)
        ts = []
        for _ in range(6):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            sol.block_solve(d_S, d_g, lam_d)
            e1.record()
            torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1))
        ms_d = float(np.median(ts[1:]))
        nb = min(4, B)
        Sd = d_S[:nb].cpu().numpy()
        gd = d_g[:nb].cpu().numpy()
        ld = lam_d[:nb].cpu().numpy().astype(np.float64)
        res = [float(np.linalg.norm(gd[b] - synth.bd_to_dense(np.nan_to_num(Sd[b]), N) @ ld[b]) / np.linalg.norm(gd[b])) for b in range(nb)]
        out["block_solve"] = {"kernel": "bt_block_solve_kernel (mpcg_block_solve)", "ms_per_batch": ms_d, "batch": B,
                              "linsolves_per_sec": B / (ms_d * 1e-3), "us_per_linsolve_throughput": ms_d * 1e3 / B,
                              "true_rel_residual_sample": res,
                              "note": "fp32 block LU sweep, four trajectories per wavefront; not the headline metric (which counts PCG iterations)"}

    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(N, S_h, g_h, float(it_host.mean()), args.cpu_seconds)
        out["cpu_baseline"]["gpu_linsolves_per_sec"] = out["linsolves_per_sec"]

    if rank == 0:
        print(json.dumps(out))
    if torch.distributed.is_initialized():
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
