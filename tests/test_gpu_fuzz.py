"""A seeded slice of the differential fuzz (tests/fuzz_cases.py; tools/fuzz_families.py runs thousands of cases) INSIDE the `-m gpu` suite
(VERDICT r04 #2a): 200 random (N in 2..400, batch, preconditioner, warm start, iteration cap) calls on the default launch policy — kernel
families 5 / 6 / 7 — each against the float64 oracle iterate after the same number of iterations, then with fp16 matrix storage."""
import pytest

pytestmark = pytest.mark.gpu


def test_seeded_fuzz_slice_of_the_default_kernels(orc):
    import fuzz_cases
    r = fuzz_cases.run(cases=200, seed=20250929, verbose=True)
    assert r["mismatches"] == 0, r
    assert set(r["families"]) == {5, 6, 7, 11} and min(r["families"].values()) >= 20, r["families"]          # every family the policy picks was exercised
    assert r["warm_start_cases"] >= 60 and r["f16_bitwise_cases"] >= 40, r
    # the test-suite tolerance (max(1e-3, 4 x the CPU float32 band)) with no trajectory beyond it — not only below the fuzz's 2x failure line
    assert r["worst_error_over_tolerance"] <= 1.0 and r["marginal_trajectories"] == 0, r
