#!/usr/bin/env python3
"""Condense rocprofv3 result databases (rocpd sqlite, ROCm 7.2 default output) into the small text /
JSON summaries that are committed under profiles/.

  python tools/rocprof_summary.py stats  gpurun_out/prof_r01/r01_results.db  > profiles/r01_kernel_stats.txt
  python tools/rocprof_summary.py pmc    gpurun_out/pmc_fetch/f_results.db gpurun_out/pmc_write/w_results.db \
                                         > profiles/r01_pmc.json
"""
import json
import sqlite3
import sys


def short(name):
    name = name.replace("void ", "")
    return name if len(name) < 90 else name[:87] + "..."


def stats(db):
    c = sqlite3.connect(db)
    print(f"# rocprofv3 --kernel-trace --stats  ({db})")
    print(f"{'kernel':90s} {'calls':>6s} {'total_us':>12s} {'avg_us':>12s} {'min_us':>12s} {'max_us':>12s} {'pct':>6s}")
    rows = list(c.execute(
        "select name, count(*), sum(end-start)/1e3, avg(end-start)/1e3, min(end-start)/1e3, max(end-start)/1e3 "
        "from kernels group by name order by 3 desc"))
    tot = sum(r[2] for r in rows)
    for nm, n, t, a, mn, mx in rows:
        print(f"{short(nm):90s} {n:6d} {t:12.2f} {a:12.2f} {mn:12.2f} {mx:12.2f} {100 * t / tot:6.2f}")
    # the same kernels by launch shape (grid size): a bench run launches one kernel on several batch sizes, and bench.py's
    # roofline objects each describe ONE of them; median = robust against the few small side launches of persistent-grid kernels
    print("# by launch shape: kernel | grid_x | calls | avg_us | median_us")
    try:
        shapes = {}
        for nm, grid, dur in c.execute("select k.name, d.grid_size_x, (k.end - k.start) / 1e3 from kernels k join rocpd_kernel_dispatch d "
                                       "on d.dispatch_id = k.dispatch_id where k.name like '%mpcg%'"):
            shapes.setdefault((nm, grid), []).append(dur)
        for (nm, grid), ds in sorted(shapes.items(), key=lambda kv: -sum(kv[1])):
            ds.sort()
            print(f"#   {short(nm):80s} {grid:9d} {len(ds):6d} {sum(ds) / len(ds):12.2f} {ds[len(ds) // 2]:12.2f}")
    except sqlite3.Error as e:
        print("#   (shape join failed:", e, ")")
    print("# per-dispatch resources")
    try:
        for r in c.execute("select distinct k.name, s.arch_vgpr_count, s.accum_vgpr_count, s.sgpr_count, d.group_segment_size, "
                           "d.workgroup_size_x, d.grid_size_x from kernels k join rocpd_kernel_dispatch d on d.dispatch_id = k.dispatch_id "
                           "join rocpd_info_kernel_symbol s on s.id = d.kernel_id where k.name like '%mpcg%'"):
            print(f"#   {short(r[0])}: vgpr={r[1]} agpr={r[2]} sgpr={r[3]} lds={r[4]}B wg={r[5]} grid={r[6]}")
    except sqlite3.Error as e:
        print("#   (resource join failed:", e, ")")


def pmc(dbs):
    out = {"note": "per-launch averages; FETCH_SIZE/WRITE_SIZE are in KiB as reported; on gfx950 FETCH_SIZE counts 128-B "
                   "read requests as 64 B for wide coalesced streams (MI355X_MICROARCH.md §HBM) -> fetch_bytes_corrected = 2x",
           "kernels": {}}
    for db in dbs:
        c = sqlite3.connect(db)
        for nm, cnt, n, avg, mn, mx, dur in c.execute(
                "select name, counter_name, count(*), avg(counter_value), min(counter_value), max(counter_value), avg(duration) "
                "from pmc_events where name like '%mpcg%' group by name, counter_name"):
            k = out["kernels"].setdefault(short(nm), {})
            k[cnt] = {"launches": n, "avg": avg, "min": mn, "max": mx, "avg_duration_us_profiled": dur / 1e3}
    for k in out["kernels"].values():
        if "FETCH_SIZE" in k:
            k["fetch_bytes_raw"] = k["FETCH_SIZE"]["avg"] * 1024
            k["fetch_bytes_corrected"] = 2 * k["fetch_bytes_raw"]
        if "WRITE_SIZE" in k:
            k["write_bytes"] = k["WRITE_SIZE"]["avg"] * 1024
        if "fetch_bytes_corrected" in k and "write_bytes" in k:
            k["hbm_traffic_bytes_per_launch"] = k["fetch_bytes_corrected"] + k["write_bytes"]
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    if sys.argv[1] == "stats":
        stats(sys.argv[2])
    else:
        pmc(sys.argv[2:])
