#!/usr/bin/env python3
"""bench.py — headline benchmark of the PCG hot path (BASELINE.json metric).

  python bench.py [--gpus N] [--steps K] [--warmup W]
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
         --master-port P bench.py --gpus N --steps K --warmup W

One "step" = one batched SQP-linsolve as the reference times it (include/pcg/sqp.cuh:224-241: launch + the two D2H copies of
(pcg_iters, pcg_exit)): `batch` independent IIWA-14 trajectories (N=128 knots, symmetric-stair preconditioner, lambda0 = 0,
pcg_max_iter = 167, pcg_exit_tol = 1e-4) solved by ONE launch of the PCG kernel on every GPU.  Trajectories are independent, so
ranks just get their own `batch` (weak scaling) and the only collectives are the barrier, the max-over-ranks of the time and the
gather of the per-trajectory results (RCCL).  Inputs are synthetic (mpcgpu_amd.synth), built and resident in HBM before the timed region.

Output (rank 0):
  * the LAST stdout line = ONE compact JSON object (< 6 KB, strict JSON, scalars first): metric / value / steps / ms_per_step / config /
    roofline / cpu_baseline — what the driver parses (`compact_line`, tests/test_bench_line.py holds its size);
  * everything else — every side leg's full object — goes to `gpurun_out/bench_full.json` (`--full-json PATH`), never to stdout.
The side legs live in tools/bench_legs.py; `--no-extras` runs the timed region only.
"""
from __future__ import annotations

import argparse
import json
import math
import os
import sys
import time
import types

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))

from mpcgpu_amd import PcgSolver, pcg_config, synth  # noqa: E402
from mpcgpu_amd import dist as D  # noqa: E402
import bench_legs as L  # noqa: E402
from bench_legs import build_inputs, load_traffic, timed  # noqa: E402,F401  (tests and tools use bench.build_inputs / bench.timed)

HBM_PEAK_GBS = 8000.0      # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
FP32_VALU_PEAK_TF = 157.3  # same guide: peak fp32 vector = 256 CUs x 4 SIMDs x 64 flop/clk x 2.4 GHz
LINE_LIMIT = 6000          # bytes; the driver's record of round 4 lost a 23 KB line


# --------------------------------------------------------------------------------------------------------------------------------
# the compact line
# --------------------------------------------------------------------------------------------------------------------------------

def _num(v, sig=6):
    """A JSON-safe scalar: floats rounded to `sig` significant digits, non-finite -> None, numpy scalars -> python."""
    if isinstance(v, (bool, np.bool_)):
        return bool(v)
    if isinstance(v, (int, np.integer)):
        return int(v)
    if isinstance(v, (float, np.floating)):
        v = float(v)
        if not math.isfinite(v):
            return None
        if v == 0.0:
            return 0.0
        return float(f"{v:.{sig}g}")
    return v


def sanitize(o, sig=None):
    """Strict-JSON form of a nested record (no NaN / Infinity, numpy types converted)."""
    if isinstance(o, dict):
        return {str(k): sanitize(v, sig) for k, v in o.items()}
    if isinstance(o, (list, tuple)):
        return [sanitize(v, sig) for v in o]
    if isinstance(o, np.ndarray):
        return sanitize(o.tolist(), sig)
    if isinstance(o, (float, np.floating)):
        f = float(o)
        return (f if sig is None else _num(f, sig)) if math.isfinite(f) else None
    if isinstance(o, (np.integer,)):
        return int(o)
    if isinstance(o, (np.bool_,)):
        return bool(o)
    return o


def _get(d, *path):
    for k in path:
        if not isinstance(d, dict) or k not in d:
            return None
        d = d[k]
    return d


def compact_line(full: dict) -> dict:
    """The object the driver parses, picked out of the full record.  Scalars only inside `config` / `roofline` / `cpu_baseline`
    (one small nested object: cpu_baseline.all_cores), strings short, every float at 6 significant digits."""
    top = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data")
    line = {k: _num(full.get(k)) for k in top}
    c, sus, ii, b1, lh = full.get("config", {}), full.get("sustained") or {}, full.get("iiwa_run") or {}, full.get("batch1_sqp_step_latency") or {}, full.get("long_horizon") or {}
    warm, walk = ii.get("warm_start_from_previous_sqp_iterate") or {}, ii.get("warm_start_consecutive_control_steps") or {}
    cfg = {
        "workload": c.get("workload"), "knot_points": c.get("knot_points"), "state_size": c.get("state_size"), "batch_per_gpu": c.get("batch_per_gpu"),
        "global_batch": c.get("global_batch"), "precond": c.get("precond"), "pcg_max_iter": c.get("pcg_max_iter"), "pcg_exit_tol": c.get("pcg_exit_tol"),
        "parallelism": c.get("parallelism"), "kernel_family": c.get("kernel_family"),
        "mean_pcg_iters": full.get("mean_pcg_iters"), "max_iter_exit_rate": full.get("max_iter_exit_rate"), "ms_per_linsolve": full.get("ms_per_linsolve"),
        "linsolves_per_sec": full.get("linsolves_per_sec"),
        "sustained_value": sus.get("value"), "sustained_seconds": sus.get("seconds"), "sustained_sclk_mhz_median": _get(sus, "sclk_mhz", "median"),
        "iiwa_sqp_linear_step_ms": _get(ii, "sqp_linear_step_graph", "ms_per_batch"), "iiwa_generate_kkt_ms": ii.get("generate_kkt_ms"),
        "iiwa_generate_kkt_f32_ms": ii.get("generate_kkt_f32_ms"), "iiwa_sqp_linear_step_kkt_f32_ms": _get(ii, "sqp_linear_step_graph", "ms_per_batch_kkt_f32"),
        "iiwa_form_schur_ms": ii.get("form_schur_ms"), "iiwa_warm_pcg_ms": warm.get("kernel_ms"), "iiwa_compute_dz_ms": ii.get("compute_dz_ms"),
        "iiwa_warm_mean_pcg_iters": warm.get("mean_pcg_iters"),
        "iiwa_warm_linsolves_per_sec_hint_exact": warm.get("linsolves_per_sec"),
        "iiwa_warm_linsolves_per_sec_hint_realistic": walk.get("linsolves_per_sec_hint_realistic"),
        "iiwa_warm_linsolves_per_sec_hint_off": walk.get("linsolves_per_sec_hint_off") or warm.get("linsolves_per_sec_sched_hint_off"),
        "iiwa_warm_true_residual_median": warm.get("true_rel_residual_after_median"), "iiwa_warm_true_residual_max": warm.get("true_rel_residual_after_max"),
        "batch1_sqp_step_us_N32": _get(b1, "N32", "us_per_step"), "batch1_sqp_step_us_N128": _get(b1, "N128", "us_per_step"),
        "N256_pcg_iterations_per_sec": _get(lh, "N256", "pcg_iterations_per_sec"), "N512_pcg_iterations_per_sec": _get(lh, "N512", "pcg_iterations_per_sec"),
        "N256_single_reduction_ceiling_it_per_sec": (_get(full, "roofline_long_horizon", "single_reduction_ceiling", "N256_batch1024_M_it_per_s", "emulated") or 0) * 1e6 or None,
        "N32_batch2048_pcg_iterations_per_sec": _get(full, "short_horizon", "ss", "pcg_iterations_per_sec"),
        "f64_N128_pcg_iterations_per_sec": _get(full, "double_precision", "pcg_iterations_per_sec"),
        "f64_N128_frac_of_fp64_valu_peak": _get(full, "double_precision", "frac_of_fp64_valu_peak"),
        "f64_N128_it_per_sec_at_167_iterations": _get(full, "double_precision", "at_iteration_cap", "pcg_iterations_per_sec"),
        "f64_N128_row_per_lane_clusters_it_per_sec": _get(full, "double_precision", "row_per_lane_clusters", "pcg_iterations_per_sec"),
        "f64_N64_pcg_iterations_per_sec": _get(full, "double_precision", "N64", "pcg_iterations_per_sec"),
        "f64_N128_streaming_it_per_sec": _get(full, "double_precision", "streaming_kernel", "pcg_iterations_per_sec"),
        "f64_N128_streaming_frac_of_hbm": _get(full, "double_precision", "streaming_kernel", "frac"),
        "f64_form_schur_ms": _get(full, "roofline_producers_f64", "form_schur_f64", "kernel_ms"),
        "f64_compute_dz_ms": _get(full, "roofline_producers_f64", "compute_dz_f64", "kernel_ms"),
        "parity_sample_all_ok": _get(full, "parity_sample", "all_ok"),
        "strong_scaling_speedup_ceiling_at_8_gpus": _get(full, "scaling_expectation", "strong", "speedup_ceiling_at_8_gpus"),
    }
    line["config"] = {k: (_num(v) if not isinstance(v, str) else v[:118]) for k, v in cfg.items() if v is not None}
    sp, rr, prod = full.get("roofline_spmv") or {}, full.get("roofline_resident") or {}, full.get("roofline_producers") or {}
    if full.get("roofline_is") == "headline_streaming":       # (explicit all-streaming PCG run: the timed kernel itself is the HBM kernel)
        sp = full.get("roofline") or {}
    roof = {
        "bound": sp.get("bound"), "kernel": sp.get("kernel"), "achieved": sp.get("achieved"), "peak": sp.get("peak"), "unit": sp.get("unit"), "frac": sp.get("frac"),
        "traffic": sp.get("traffic"), "kernel_ms": sp.get("kernel_ms"), "bytes_per_unit": sp.get("bytes_per_unit"), "units_per_launch": sp.get("units_per_launch"),
        "traffic_over_algorithmic": sp.get("traffic_over_algorithmic"),
        "read_ceiling_gbs_this_run": sp.get("read_ceiling_gbs_this_run"), "frac_of_read_ceiling_this_run": sp.get("frac_of_read_ceiling_this_run"),
        "in_timed_region": False if sp.get("kernel") == "bt_spmv_kernel" else None,
        "headline_kernel": rr.get("kernel"), "headline_bound": rr.get("bound"), "headline_frac": rr.get("frac"), "headline_achieved": rr.get("achieved"),
        "headline_peak": rr.get("peak"), "headline_unit": rr.get("unit"), "headline_kernel_ms": rr.get("kernel_ms"),
        "headline_valu_active_frac": rr.get("valu_active_frac"), "headline_traffic": rr.get("traffic"),
        "headline_hbm_algorithmic_bytes": rr.get("hbm_algorithmic_bytes_per_launch"),
        "form_schur_frac": _get(prod, "form_schur", "frac"), "form_schur_traffic_over_algorithmic": _get(prod, "form_schur", "traffic_over_algorithmic"),
        "compute_dz_frac": _get(prod, "compute_dz", "frac"), "generate_kkt_frac_of_fp64_valu_peak": _get(prod, "generate_kkt", "frac_of_valu_peak"),
        "generate_kkt_f32_frac_of_fp32_valu_peak": _get(prod, "generate_kkt", "frac_of_fp32_valu_peak_kkt_f32"),
        "pcg_streaming_N512_frac": _get(full, "roofline_pcg_streaming", "frac"),
        "traffic_source": sp.get("traffic_source"),
    }
    line["roofline"] = {k: (_num(v) if not isinstance(v, str) else v[:118]) for k, v in roof.items() if v is not None}
    if "traffic" not in line["roofline"]:
        line["roofline"]["traffic"] = None
    cb = full.get("cpu_baseline")
    if cb:
        keys = ("value", "unit", "cores", "kind", "ms_per_linsolve", "sample", "cpu_model", "host_cpus", "config1_N32_us_per_linsolve", "equiv_pcg_iters_per_sec",
                "cpu_pcg_port_iters_per_sec", "rel_residual_float_ldl", "gpu_linsolves_per_sec")
        o = {k: (_num(cb[k]) if not isinstance(cb[k], str) else cb[k][:118]) for k in keys if cb.get(k) is not None}
        ac = cb.get("all_cores")
        if ac:
            o["all_cores"] = {k: _num(ac[k]) for k in ("value", "unit", "cores") if k in ac}
        line["cpu_baseline"] = o
    rg = full.get("results_gather")
    if rg:
        line["results_gather"] = {k: rg[k] for k in ("backend", "trajectories", "consistent_with_allreduce_sum") if k in rg}
    if full.get("per_rank_kernel_ms") is not None:
        line["per_rank_kernel_ms"] = [_num(v) for v in full["per_rank_kernel_ms"]][:16]
    line["self_launched"] = full.get("self_launched")
    line["full_record"] = full.get("full_record")
    return sanitize(line)


def emit(full: dict, path: str | None):
    """Write the full record to `path` (best effort) and print the compact line as the LAST stdout line."""
    if path:
        try:
            os.makedirs(os.path.dirname(os.path.abspath(path)), exist_ok=True)
            with open(path, "w") as f:
                json.dump(sanitize(full), f, indent=1, allow_nan=False)
            full["full_record"] = os.path.relpath(path, ROOT) if os.path.abspath(path).startswith(ROOT) else path
        except OSError as e:
            full["full_record"] = f"not written: {e!r}"
    line = json.dumps(compact_line(full), allow_nan=False, separators=(",", ":"))
    assert len(line) < LINE_LIMIT, f"bench line grew to {len(line)} bytes"
    sys.stdout.flush()
    print(line, flush=True)
    return line


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


def run_leg(full, key, fn, *a, **kw):
    """A side leg never costs the headline line: its exception becomes {"error": ...} in the full record."""
    t0 = time.perf_counter()
    try:
        r = fn(*a, **kw)
    except Exception as e:                                       # noqa: BLE001
        r = {"error": repr(e)}
    if isinstance(r, dict):
        r.setdefault("leg_seconds", round(time.perf_counter() - t0, 2))
    if key:
        full[key] = r
    return r


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)     # (0.25 s of timed steps: the GPU's steady state)
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
    ap.add_argument("--no-extras", action="store_true", help="only the timed region (no side legs)")
    ap.add_argument("--profile-lean", action="store_true",
                    help="for rocprofv3 passes: only the timed region + the two HBM roofline kernels, so that per-kernel averages are not "
                         "diluted by the batch-1 latency launches and the parity sample")
    ap.add_argument("--profile-mini", action="store_true",
                    help="what the PMC passes execute: the timed region, the SpMV leg and the producer kernels (one launch shape each), nothing else")
    ap.add_argument("--inrun-pmc", action="store_true",
                    help="OPT-IN: spawn two rocprofv3 --pmc child passes of this script for roofline.traffic (default: profiles/traffic.json, the committed passes)")
    ap.add_argument("--sustain-seconds", type=float, default=3.0, help="length of the sustained leg (back-to-back timed steps; 0 = skip)")
    ap.add_argument("--spmv-mfma", action="store_true", help="also time BASELINE config 5's MFMA block-GEMV experiment")
    ap.add_argument("--spmv-batch", type=int, default=4096, help="trajectories streamed by the SpMV roofline run (S = batch x 301 KB)")
    ap.add_argument("--storage", default="f32", choices=["f32", "f16"],
                    help="matrix storage of S/Pinv; f16 = BASELINE config 5's reduced-precision experiment (arithmetic stays fp32)")
    ap.add_argument("--full-json", default=os.path.join(ROOT, "gpurun_out", "bench_full.json"), help="where the full record goes ('' = nowhere)")
    ap.add_argument("--plumbing-only", action="store_true",
                    help="launcher / collective dry run (CPU test of the N > 1 flow, tests/test_bench_launcher.py): every rank runs the whole "
                         "distributed flow of this file with made-up iteration counts and NO solve; prints value = null")
    args = ap.parse_args()
    if args.profile_mini:
        args.profile_lean = True
        args.no_cpu_baseline = True

    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        # bare `python bench.py --gpus N`: this process becomes the launcher of N ranks (one per GPU, RCCL), it does not degrade to one rank
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

    sol = PcgSolver(N, max_batch=max(B, args.spmv_batch if not args.no_extras else B), device=local_rank)
    d_S, d_P, d_g = L.build_inputs(sol, N, B, seed0, args.precond, dev, chunk=B if args.profile_lean else 128)      # (PMC passes: one launch shape per kernel)
    ns = min(32, B)
    S_h, P_h, g_h = (t[:ns].cpu().numpy() for t in (d_S, d_P, d_g))     # host copies of the CPU-baseline sample
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

    # like every leg (L.timed): the clocks first, then the caller's W warm-up steps, then exactly K timed steps
    t_ramp = time.perf_counter() + 2.5 * L.RAMP_S
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

    fam = {0: "pcg_traj_kernel", 3: "pcg_generic_kernel", 5: "pcg_rpl_kernel", 6: "pcg_lpk_kernel", 7: "pcg_lpkc_kernel", 11: "pcg_lqb_kernel"}[sol.get_option("last_kernel_family")]
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

    cx = types.SimpleNamespace(sol=sol, dev=dev, N=N, B=B, args=args, cfg=cfg, max_iter=max_iter, rank=rank, world=world, seed0=seed0,
                               d_S=d_S, d_P=d_P, d_g=d_g, d_lam=d_lam, d_it=d_it, d_ex=d_ex, run_solve=run_solve, it_host=it_host,
                               S_h=S_h, P_h=P_h, g_h=g_h)

    # ---- sustained leg: the very same step back to back for >= --sustain-seconds (steady-state clocks; a 25 ms timed region is invisible
    # to a 5 s utilisation sampler) ----
    sustained = None
    if args.sustain_seconds > 0 and not args.profile_lean:
        n_chunk = max(8, int(0.25 / max(ms_per_step * 1e-3, 1e-5)))          # ~0.25 s of steps between two host synchronisations
        done, t_s0 = 0, time.perf_counter()
        D.barrier()
        with L.SclkSampler(local_rank) as clk:
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
        "config": {"workload": f"IIWA-14 N={N} fp32 {args.precond}-PCG, batch {B}/GPU (BASELINE config 4), lambda0=0",
                   "workload_detail": f"IIWA-14 (n=14) N={N} knots, {args.precond} preconditioner, batch {B} trajectories on this GPU / {B_global} in total "
                                      f"(S+Pinv = {2 * B * 3 * 196 * N * 4 / 1e6:.0f} MB in HBM per GPU), lambda0=0, max_iter={max_iter}, exit_tol={args.exit_tol:g}",
                   "knot_points": N, "state_size": 14, "batch_per_gpu": B, "global_batch": B_global,
                   "precond": args.precond, "pcg_max_iter": max_iter, "pcg_exit_tol": args.exit_tol,
                   "parallelism": f"batch-sharded x{world} ({args.scaling} scaling, no data-path collective)", "kernel_family": fam, "kernel": kdesc},
        "ms_per_linsolve": ms_per_step / B,
        "linsolves_per_sec": B_global / (ms_per_step * 1e-3),
        "mean_pcg_iters": float(it_host.mean()), "max_iter_exit_rate": float(ex_host.mean()),
        "resident_trajectories_per_gpu": sol.checkPcgOccupancy(),
        "results_gather": {"collective": "all_gather of (iters, exit) per trajectory", "backend": D.backend_name(),
                           "trajectories": int(g_it.numel()), "consistent_with_allreduce_sum": gathered_ok},
        "per_rank_kernel_ms": kern_ms_ranks,
        "timed_region": "K x [lambda <- 0; mpcg_pcg_solve; D2H of (iters u32, exit u8) per trajectory], barrier + synchronize on both sides, max over ranks "
                        "(the reference's own timed region: include/pcg/sqp.cuh:224-241)",
        "self_launched": os.environ.get("MPCG_BENCH_SELF_LAUNCHED") == "1",
        "sustained": sustained,
    }

    # ---- the kernel `value` is measured on ----
    dom = {"kernel": fam, "kernel_ms": kern_ms, "kernel_ms_max_over_ranks": kern_ms_all, "units_per_launch": iters_step_local,
           "unit_of_work": "one PCG iteration of one trajectory"}
    if fam == "pcg_traj_kernel" and kdesc["reg_rows"] == 0:
        # nothing resident: the HBM model applies as written
        ach = iters_step_local * bytes_iter / (kern_ms * 1e-3) / 1e9
        out["roofline"] = {"bound": "hbm", **dom, "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS,
                           "traffic": None, "bytes_per_unit": bytes_iter}
        out["roofline_is"] = "headline_streaming"
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
        tr, src = L.load_traffic(f"{fam}|N{N}_B{B}_{args.precond}_it{max_iter}_tol{args.exit_tol:g}")
        if tr:
            rr["traffic"] = tr["hbm_traffic_bytes_per_launch"]
            rr["traffic_source"] = src
            rr["traffic_gbs"] = tr["hbm_traffic_bytes_per_launch"] / (kern_ms * 1e-3) / 1e9
            rr["traffic_frac_of_hbm_peak"] = rr["traffic_gbs"] / HBM_PEAK_GBS
            if "valu_active_frac" in tr:
                rr["valu_active_frac"] = tr["valu_active_frac"]      # SQ_ACTIVE_INST_VALU / (4 SIMDs x SQ_BUSY_CYCLES), same passes
        out["roofline_resident"] = rr

    extras = not args.no_extras and args.storage == "f32"
    lean, mini = args.profile_lean, args.profile_mini
    solo = rank == 0 and world == 1
    if extras:
        run_leg(out, "roofline_spmv", L.spmv_roofline, cx)
        if "roofline" not in out:
            out["roofline"] = out["roofline_spmv"]                # (the full record keeps the driver's key too)
    if extras and rank == 0 and not lean:
        run_leg(out, "parity_sample", L.parity_sample, cx)
    if extras and not lean:
        run_leg(out, "warm_start_run", L.warm_start_synthetic, cx)
    if extras and lean and N <= 256:
        run_leg(out, None, L.producers_mini, cx)
    if extras and not lean and N <= 256:
        r = run_leg(out, None, L.iiwa_run, cx)
        if isinstance(r, tuple):
            out["iiwa_run"], out["roofline_producers"] = r
        else:
            out["iiwa_run"] = r
        run_leg(out, "roofline_producers_f64", L.producers_f64, cx)
    if extras and solo and not lean:
        run_leg(out, "short_horizon", L.short_horizon, cx)
        run_leg(out, "config2_latency", L.latency_config2, cx)
        run_leg(out, "batch1_sqp_step_latency", L.batch1_sqp_step_latency, cx)
        run_leg(out, "single_trajectory_latency", L.single_trajectory_latency, cx)
    if extras and solo and not mini:
        r = run_leg(out, None, L.long_horizon, cx, lean)
        if isinstance(r, tuple):
            lh, stream = r
            if lh:
                out["long_horizon"] = lh
                run_leg(out, "roofline_long_horizon", L.long_horizon_floor, cx, lh)
            if stream:
                out["roofline_pcg_streaming"] = stream
        else:
            out["long_horizon"] = r
        if not lean:
            run_leg(out, "double_precision", L.double_precision, cx)
    if extras and solo and not lean and args.scaling == "weak":
        run_leg(out, "scaling_expectation", L.scaling_expectation, cx)
    if extras and rank == 0 and not lean:
        run_leg(out, "block_solve", L.block_solve, cx)
    if solo and not args.no_cpu_baseline and not lean:
        cb = run_leg(out, "cpu_baseline", L.cpu_baseline, cx, args.cpu_seconds)
        if "value" in cb:
            cb["gpu_linsolves_per_sec"] = out["linsolves_per_sec"]

    if solo and extras and not lean and args.inrun_pmc:
        tail = ["--knots", str(N), "--batch", str(args.batch), "--precond", args.precond, "--exit-tol", repr(args.exit_tol), "--spmv-batch", str(args.spmv_batch)]
        if args.max_iter:
            tail += ["--max-iter", str(args.max_iter)]
        pmc, pmc_note = L.inrun_pmc(tail)
        out["inrun_pmc_note"] = pmc_note
        if pmc:
            prod = out.get("roofline_producers") or {}
            for obj, needle in ((out.get("roofline_spmv"), "bt_spmv_kernel"), (out.get("roofline_resident"), fam), (prod.get("generate_kkt"), "generate_kkt_kernel"),
                                (prod.get("compute_dz"), "compute_dz_dpp_kernel")):
                e = L.find_traffic(pmc, needle)
                if isinstance(obj, dict) and "error" not in obj and e:
                    obj["traffic"] = e["hbm_traffic_bytes_per_launch"]
                    obj["traffic_source"] = pmc_note
                    obj["traffic_over_algorithmic"] = e["hbm_traffic_bytes_per_launch"] / (obj.get("algorithmic_bytes_per_launch") or obj.get("hbm_algorithmic_bytes_per_launch")
                                                                                         or obj["bytes_per_unit"] * obj["units_per_launch"])
            ew, es = L.find_traffic(pmc, "schur_walk_kernel"), L.find_traffic(pmc, "schur_seam_kernel")
            if ew and "form_schur" in prod:
                tb = ew["hbm_traffic_bytes_per_launch"] + (es["hbm_traffic_bytes_per_launch"] if es else 0.0)
                prod["form_schur"].update({"traffic": tb, "traffic_source": pmc_note, "traffic_over_algorithmic": tb / (prod["form_schur"]["bytes_per_unit"] * prod["form_schur"]["units_per_launch"])})
            out["inrun_pmc_kernels"] = pmc

    # the JSON line goes out LAST: RCCL writes a version banner through C stdio when the communicator goes away
    if torch.distributed.is_initialized():
        torch.distributed.destroy_process_group()
    try:
        import ctypes
        ctypes.CDLL(None).fflush(None)
    except OSError:
        pass
    if rank == 0:
        emit(out, None if mini else (args.full_json or None))


if __name__ == "__main__":
    main()
