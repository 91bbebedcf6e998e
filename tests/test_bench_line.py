"""The ONE line bench.py prints for the driver (the 23 KB line of round 4 was not parsed): built from a worst-case full
record it stays under bench.LINE_LIMIT bytes (the driver keeps an 8 KB tail), is strict JSON (no NaN / Infinity), carries the contract's
keys with `roofline` and `cpu_baseline`, and the multi-rank line is equally small."""
import json
import math
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

LONG = "x" * 400
PI = math.pi * 1e6


def _full(n_gpus=1):
    f = lambda: PI
    kern = {"kernel_ms": f(), "bytes_per_unit": 8036, "achieved": f(), "unit": "GB/s", "frac": f(), "traffic": f(), "traffic_over_algorithmic": f(), "frac_of_valu_peak": f(),
            "bound": LONG}
    full = {
        "metric": "pcg_iterations_per_sec", "value": 134598642.5401271, "unit": "iter/s", "n_gpus": n_gpus, "steps": 20, "warmup": 5, "ms_per_step": 1.2705031549558043,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": LONG, "workload_detail": LONG, "knot_points": 128, "state_size": 14, "batch_per_gpu": 1024, "global_batch": 1024 * n_gpus, "precond": "ss",
                   "pcg_max_iter": 167, "pcg_exit_tol": 1e-4, "parallelism": LONG, "kernel_family": "pcg_lpk_kernel", "kernel": {"family": "pcg_lpk_kernel", "waves": 8}},
        "ms_per_linsolve": f(), "linsolves_per_sec": f(), "mean_pcg_iters": 167.0, "max_iter_exit_rate": 1.0, "resident_trajectories_per_gpu": 256,
        "results_gather": {"collective": LONG, "backend": "nccl (RCCL)", "trajectories": 1024 * n_gpus, "consistent_with_allreduce_sum": True},
        "per_rank_kernel_ms": [f()] * n_gpus, "timed_region": LONG, "self_launched": n_gpus > 1,
        "sustained": {"steps": 2626, "seconds": f(), "ms_per_step": f(), "value": f(), "sclk_mhz": {"min": 2369, "median": 2392, "max": 2392}, "what": LONG},
        "roofline_resident": {"bound": "valu", "kernel": "pcg_lpk_kernel", "kernel_ms": f(), "achieved": f(), "peak": 157.3, "unit": "TFLOP/s", "frac": f(), "traffic": f(),
                              "valu_active_frac": f(), "hbm_algorithmic_bytes_per_launch": 431513600, "why_not_hbm": LONG, "traffic_source": LONG},
        "roofline_spmv": {"bound": "hbm", "kernel": "bt_spmv_kernel", "achieved": f(), "peak": 8000.0, "unit": "GB/s", "frac": f(), "kernel_ms": f(), "traffic": f(),
                          "algorithmic_bytes_per_launch": 1285423104, "bytes_per_unit": 313824, "units_per_launch": 4096, "unit_of_work": LONG, "traffic_over_algorithmic": f(),
                          "read_ceiling_gbs_this_run": f(), "frac_of_read_ceiling_this_run": f(), "traffic_source": LONG},
        "parity_sample": {"all_ok": True, "samples": [{"rel_err_vs_f64_same_iters": float("nan")}] * 4},
        "warm_start_run": {"mean_pcg_iters": f()},
        "iiwa_run": {"inputs": LONG, "generate_kkt_ms": f(), "form_schur_ms": f(), "compute_dz_ms": f(),
                     "warm_start_from_previous_sqp_iterate": {"mean_pcg_iters": f(), "kernel_ms": f(), "linsolves_per_sec": f(), "linsolves_per_sec_sched_hint_off": f(),
                                                              "true_rel_residual_after_median": f(), "true_rel_residual_after_max": float("inf")},
                     "warm_start_consecutive_control_steps": {"linsolves_per_sec_hint_realistic": f(), "linsolves_per_sec_hint_off": f()},
                     "sqp_linear_step_graph": {"ms_per_batch": f()}},
        "roofline_producers": {"form_schur": dict(kern), "compute_dz": dict(kern), "generate_kkt": dict(kern)},
        "roofline_producers_f64": {"form_schur_f64": dict(kern), "compute_dz_f64": dict(kern)},
        "batch1_sqp_step_latency": {"N32": {"us_per_step": f()}, "N64": {"us_per_step": f()}, "N128": {"us_per_step": f()}, "what": LONG},
        "long_horizon": {"N256": {"pcg_iterations_per_sec": f()}, "N512": {"pcg_iterations_per_sec": f()}},
        "roofline_long_horizon": {"single_reduction_ceiling": {"N256_batch1024_M_it_per_s": {"classic": 38.22, "emulated": 41.48, "target": 44.0}}},
        "roofline_pcg_streaming": {"frac": f()}, "double_precision": {"pcg_iterations_per_sec": f(), "streaming_kernel": {"pcg_iterations_per_sec": f(), "frac": f()}},
        "scaling_expectation": {"strong": {"speedup_ceiling_at_8_gpus": f()}},
        "inrun_pmc_kernels": {LONG + str(i): {"kernel": LONG, "grid": 1} for i in range(12)},
        "cpu_baseline": {"value": f(), "unit": "linsolves/s", "cores": 1, "kind": "port", "ms_per_linsolve": f(), "sample": LONG, "cpu_model": LONG, "host_cpus": 256,
                         "all_cores": {"value": f(), "unit": "linsolves/s", "cores": 256, "equiv_pcg_iters_per_sec": f()}, "config1_N32_us_per_linsolve": f(),
                         "equiv_pcg_iters_per_sec": f(), "cpu_pcg_port_iters_per_sec": f(), "rel_residual_float_ldl": f(), "residual_note": LONG, "gpu_linsolves_per_sec": f()},
        "full_record": "gpurun_out/bench_full.json",
    }
    return full


def _strict(s):
    def bad(c):
        raise ValueError("non-finite constant " + c)
    return json.loads(s, parse_constant=bad)


def test_compact_line_is_small_strict_json_with_the_contract_keys():
    line = json.dumps(bench.compact_line(_full()), allow_nan=False, separators=(",", ":"))
    assert len(line) < bench.LINE_LIMIT < 8192, len(line)
    o = _strict(line)
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config", "roofline",
              "cpu_baseline"):
        assert k in o, k
    assert o["steps"] == 20 and o["warmup"] == 5 and o["value"] == 134599000.0 and o["vs_baseline"] is None
    assert o["config"]["workload"] and len(o["config"]["workload"]) < 120 and o["config"]["knot_points"] == 128 and o["config"]["kernel_family"] == "pcg_lpk_kernel"
    r = o["roofline"]
    for k in ("bound", "kernel", "achieved", "peak", "unit", "frac", "traffic", "kernel_ms", "bytes_per_unit", "units_per_launch", "headline_kernel", "headline_frac",
              "form_schur_frac", "compute_dz_frac", "generate_kkt_frac_of_fp64_valu_peak"):
        assert k in r, k
    assert r["bound"] == "hbm" and r["kernel"] == "bt_spmv_kernel" and r["in_timed_region"] is False
    c = o["cpu_baseline"]
    assert c["kind"] == "port" and c["cores"] == 1 and c["all_cores"]["cores"] == 256 and len(c["sample"]) < 120
    # nothing nested beyond one level inside the three objects, no long strings anywhere
    for obj in (o["config"], o["roofline"]):
        assert all(not isinstance(v, (dict, list)) for v in obj.values())
    assert all(len(v) < 120 for obj in (o["config"], o["roofline"], o["cpu_baseline"]) for v in obj.values() if isinstance(v, str))
    # the non-finite residual of the canned record became null, not Infinity
    assert "iiwa_warm_true_residual_max" not in o["config"] or o["config"]["iiwa_warm_true_residual_max"] is None


def test_multi_rank_line_is_equally_small():
    line = json.dumps(bench.compact_line(_full(8)), allow_nan=False, separators=(",", ":"))
    assert len(line) < bench.LINE_LIMIT
    o = _strict(line)
    assert o["n_gpus"] == 8 and len(o["per_rank_kernel_ms"]) == 8 and o["results_gather"]["trajectories"] == 8192 and o["self_launched"] is True


def test_full_record_is_written_as_strict_json(tmp_path, capsys):
    p = tmp_path / "sub" / "bench_full.json"
    line = bench.emit(_full(), str(p))
    out = capsys.readouterr().out.strip().splitlines()
    assert out[-1] == line and len(out) == 1                   # the line is the ONLY thing on stdout
    full = _strict(p.read_text())
    assert full["parity_sample"]["samples"][0]["rel_err_vs_f64_same_iters"] is None
    assert "inrun_pmc_kernels" in full and "inrun_pmc_kernels" not in _strict(line)


def test_no_extras_record_still_makes_a_line():
    f = {k: v for k, v in _full().items() if k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
                                                    "dtype", "data", "config", "results_gather", "per_rank_kernel_ms", "self_launched", "roofline_resident")}
    o = _strict(json.dumps(bench.compact_line(f), allow_nan=False))
    assert o["roofline"]["headline_kernel"] == "pcg_lpk_kernel" and o["roofline"]["traffic"] is None and "cpu_baseline" not in o


def test_committed_traffic_file_has_every_key_the_bench_looks_up():
    """profiles/traffic.json (tools/make_traffic.py from the round's PMC passes) is where bench.py takes the HBM-side traffic of its roofline objects from.
    tools/make_traffic.py matches kernels by NAME: when round 6 templated generate_kkt_kernel on its arithmetic type the name changed and its entry silently
    disappeared for two profile rounds.  The keys of the default workload (bench_legs.load_traffic call sites) must all be there, with plausible byte counts."""
    import json
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    t = json.load(open(os.path.join(root, "profiles", "traffic.json")))
    assert os.path.exists(os.path.join(root, t["source"].split(" ")[0])), t["source"]
    want = ["pcg_lqb_kernel|N128_B1024_ss_it167_tol0.0001", "bt_spmv_kernel|N128_B4096", "generate_kkt|N128_B1024", "form_schur|N128_B1024", "compute_dz|N128_B1024",
            "pcg_traj_kernel<16,0,2>|N512_B1024_ss_it67_tol0"]
    for k in want:
        assert k in t["kernels"], (k, sorted(t["kernels"]))
        e = t["kernels"][k]
        assert e["hbm_traffic_bytes_per_launch"] > 1e6 and e["fetch_bytes_corrected"] >= 0 and e["write_bytes"] >= 0 and e["launches_averaged"] >= 1, (k, e)
