#!/usr/bin/env python3
"""Randomised differential run of the PCG kernel families (run on the GPU box):  fuzz_families.py [cases] [seed]
The cases themselves: tests/fuzz_cases.py (a seeded 200-case slice of it runs inside the `-m gpu` suite, tests/test_gpu_fuzz.py)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import fuzz_cases
r = fuzz_cases.run(int(sys.argv[1]) if len(sys.argv) > 1 else 120, int(sys.argv[2]) if len(sys.argv) > 2 else 7)
print(f"{r['cases']} random cases in {r['seconds']:.0f} s: kernel families {r['families']}, worst error / tolerance {r['worst_error_over_tolerance']:.3f} "
      f"({r['marginal_trajectories']} trajectories between 1x and 2x), trajectories skipped because float32 CG itself breaks down (over-iterated, exit_tol 0) "
      f"{r['breakdown_skipped']}; fp16 storage of the same cases: families {r['f16_families']}, {r['f16_bitwise_cases']} cases bit-identical to the fp32 solve of the "
      f"rounded matrices by the same kernel, the others' worst error / tolerance {r['f16_worst_error_over_tolerance']:.3f} ({r['f16_near_breakdown_skipped']} trajectories skipped: the ROUNDED system "
      f"is indefinite and its float64 CG passes a near-breakdown); mismatches {r['mismatches']}")
sys.exit(1 if r["mismatches"] else 0)
