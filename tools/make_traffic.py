#!/usr/bin/env python3
"""profiles/<tag>_pmc.json (tools/rocprof_summary.py pmc) -> profiles/traffic.json: HBM-side bytes per launch of the kernels
bench.py reports a roofline for, keyed as bench.load_traffic() looks them up.  FETCH_SIZE is doubled per
MI355X_MICROARCH.md §HBM (gfx950 counts a 128-byte read request as 64 bytes for wide coalesced streams).
   python tools/make_traffic.py r02b [N B precond max_iter tol spmv_batch]"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1]
N, B, pc, mi, tol, spB = (sys.argv[2:] + ["128", "1024", "ss", "167", "0.0001", "4096"][len(sys.argv) - 2:])[:6]
pmc = json.load(open(os.path.join(ROOT, "profiles", f"{tag}_pmc.json")))["kernels"]
out = {"source": f"profiles/{tag}_pmc.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of `python bench.py --steps 3 --warmup 1 --profile-lean`)",
       "fetch_correction": "fetch_bytes_corrected = 2 x FETCH_SIZE (gfx950 tallies 128-byte read requests at 64 bytes)", "kernels": {}}
for name, k in pmc.items():
    if "hbm_traffic_bytes_per_launch" not in k:
        continue
    if "pcg_lqb_kernel<128, true>" in name:          # the headline kernel since round 6 (the lane-pair kernel before: the same key with its name)
        key = f"pcg_lqb_kernel|N{N}_B{B}_{pc}_it{mi}_tol{float(tol):g}"
    elif "pcg_lpk_kernel" in name:
        key = f"pcg_lpk_kernel|N{N}_B{B}_{pc}_it{mi}_tol{float(tol):g}"
    elif "bt_spmv_kernel" in name:
        key = f"bt_spmv_kernel|N{N}_B{spB}"
    elif "pcg_traj_kernel<16, 0, 2" in name:
        key = f"pcg_traj_kernel<16,0,2>|N512_B{B}_ss_it67_tol0"          # bench.py's streaming leg: N = 512, batch B, 67 fixed iterations
    elif "generate_kkt_kernel<true, double>" in name or "generate_kkt_kernel<true>" in name:       # (the default build: float64 inside)
        key = f"generate_kkt|N{N}_B{B}"
    elif "compute_dz_dpp_kernel" in name:
        key = f"compute_dz|N{N}_B{B}"
    elif "schur_walk_kernel" in name:
        key = f"form_schur|N{N}_B{B}"
    else:
        continue
    if key.startswith("form_schur"):      # the formation is two kernels: the walking pass + the seam kernel
        seam = next((v for n_, v in pmc.items() if "schur_seam_kernel" in n_ and "hbm_traffic_bytes_per_launch" in v), None)
        if seam:
            k = dict(k, hbm_traffic_bytes_per_launch=k["hbm_traffic_bytes_per_launch"] + seam["hbm_traffic_bytes_per_launch"],
                     fetch_bytes_corrected=k["fetch_bytes_corrected"] + seam["fetch_bytes_corrected"], write_bytes=k["write_bytes"] + seam["write_bytes"])
    out["kernels"][key] = {"kernel": name, "hbm_traffic_bytes_per_launch": k["hbm_traffic_bytes_per_launch"],
                           "fetch_bytes_corrected": k["fetch_bytes_corrected"], "write_bytes": k["write_bytes"],
                           "launches_averaged": k["FETCH_SIZE"]["launches"]}
    # VALU pipe activity of the same kernel from the SQ pass.  Counters are per shader engine (8 CUs x 4 SIMDs = 32 SIMDs);
    # SQ_ACTIVE_INST_VALU counts quad-cycles, SQ_BUSY_CYCLES cycles (MI355X_MICROARCH.md): active = 4 x ACTIVE / (32 x BUSY)
    if "SQ_ACTIVE_INST_VALU" in k and "SQ_BUSY_CYCLES" in k and k["SQ_BUSY_CYCLES"]["avg"]:
        out["kernels"][key]["valu_active_frac"] = k["SQ_ACTIVE_INST_VALU"]["avg"] / (8.0 * k["SQ_BUSY_CYCLES"]["avg"])
json.dump(out, open(os.path.join(ROOT, "profiles", "traffic.json"), "w"), indent=1)
print(json.dumps(out, indent=1))
