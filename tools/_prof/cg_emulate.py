#!/usr/bin/env python3
"""What could a single-reduction (Chronopoulos-Gear) PCG gain on the clustered lane-pair kernel?  (VERDICT r04 #5.)

In that recurrence the hand-off after the Pinv pass needs no inner product: the S pass's operand is u = Pinv r itself, so that hand-off
carries the neighbours' halo only.  -DMPCG_CG_EMULATE builds exactly that hand-off into the shipping kernel (no partials polled, nothing
folded after the Pinv pass; the numerics are wrong, the timing is what the variant's NOT-overlapped form would have at best) — the matrix
passes, the barriers, the halo granules and the second (full) hand-off are untouched.

    python tools/_prof/cg_emulate.py build        (here: hipcc cross-compiles)  -> tools/_prof/libmpcg_hip_cgemu.so
    python tools/_prof/cg_emulate.py              (on the GPU box)              -> classic vs emulated, N = 256 / 512, batch 1024 / 1
"""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
EMU = os.path.join(ROOT, "tools", "_prof", "libmpcg_hip_cgemu.so")

if len(sys.argv) > 1 and sys.argv[1] == "build":
    from mpcgpu_amd import build as B
    subprocess.check_call([B.HIPCC, *B.FLAGS, "-DMPCG_CG_EMULATE", *B.sources(), "-o", EMU])
    print("built", EMU)
    sys.exit(0)

if len(sys.argv) > 1 and sys.argv[1] == "child":
    import numpy as np, torch
    from mpcgpu_amd import _lib as _L
    if os.environ.get("AB_LIB"):
        _L.LIB_PATH = os.environ["AB_LIB"]
    from mpcgpu_amd import PcgSolver, pcg_config, synth
    import bench
    dev = torch.device("cuda")
    out = {}
    for N in (256, 512):
        k = synth.make_kkt(N, 32, 1)
        S0, P0, g0 = synth.form_schur(k)
        for B_ in (1024, 1):
            rep = (B_ + 31) // 32
            S, P, g = (torch.from_numpy(np.tile(a, (rep, 1))[:B_]).to(dev) for a in (S0, P0, g0))
            K = synth.pcg_max_iter(N)
            cfg = pcg_config(pcg_exit_tol=0.0, pcg_max_iter=K)
            sol = PcgSolver(N, max_batch=B_)
            lam = torch.zeros(B_, 14 * N, device=dev)
            ms = bench.timed(lambda: (lam.zero_(), sol.solve(S, P, g, lam, cfg, "ss")), 9, warm=2)
            assert sol.get_option("last_kernel_family") == 7
            out[f"N{N}_B{B_}"] = {"ms": ms, "M_it_per_s": B_ * K / ms / 1e3, "us_per_iteration": ms * 1e3 / K if B_ == 1 else None}
    print("RESULT " + json.dumps(out))
    sys.exit(0)

res = {}
for name, lib in (("classic", ""), ("halo_only_handoff_after_the_pinv_pass", EMU)):
    env = dict(os.environ, AB_LIB=lib)
    r = subprocess.run([sys.executable, os.path.abspath(__file__), "child"], env=env, capture_output=True, text=True, timeout=600)
    line = [ln for ln in r.stdout.splitlines() if ln.startswith("RESULT ")]
    if not line:
        print(r.stdout[-2000:], r.stderr[-2000:])
        sys.exit(1)
    res[name] = json.loads(line[-1][7:])
for key in res["classic"]:
    a, b = res["classic"][key], res["halo_only_handoff_after_the_pinv_pass"][key]
    print(f"{key}: classic {a['M_it_per_s']:.2f} M it/s ({a['ms']:.3f} ms)  |  emulated single-reduction hand-off {b['M_it_per_s']:.2f} M it/s ({b['ms']:.3f} ms)  |  gain {b['M_it_per_s'] / a['M_it_per_s'] - 1:+.1%}")
print(json.dumps(res))
