"""CPU: host-side logic — synthetic problem generator, byte model, config tables, sharding."""
import numpy as np

from mpcgpu_amd import synth
from mpcgpu_amd.dist import shard_range


def test_algorithmic_bytes_match_baseline_md():
    # BASELINE.md §2 table
    assert [synth.algorithmic_bytes(N)["spmv"] for N in (32, 128, 512)] == [77280, 313824, 1260000]
    assert [synth.algorithmic_bytes(N, precond="ss")["pcg_iter"] for N in (32, 128, 512)] == [158144, 641984, 2577344]
    assert [synth.algorithmic_bytes(N, precond="jacobi")["pcg_iter"] for N in (32, 128, 512)] == [109536, 442848, 1776096]


def test_pcg_max_iter_table():
    # include/common/settings.cuh:123-139
    assert [synth.pcg_max_iter(N) for N in (32, 64, 128, 256, 512, 100)] == [173, 167, 167, 118, 67, 200]


def test_generator_is_deterministic_and_batch_independent():
    a = synth.make_kkt(6, 3, 42)
    b = synth.make_kkt(6, 5, 42)
    for f in ("Q", "R", "A", "Bm", "q", "r", "c"):
        np.testing.assert_array_equal(getattr(a, f), getattr(b, f)[:3])
    c = synth.make_kkt(6, 3, 43)
    assert not np.array_equal(a.A, c.A)
    assert (a.c[:, 0] == 0).all()                  # c_0 = x_0 - x_s = 0
    # Q = blkdiag(q_pos q_pos^T, QD*I), R = R_COST*I
    np.testing.assert_allclose(a.Q[0, 2, :7, :7], np.outer(a.q[0, 2, :7], a.q[0, 2, :7]))
    np.testing.assert_array_equal(a.Q[0, 2, 7:, 7:], synth.QD_COST * np.eye(7))


def test_dense_layouts():
    N, n, m = 5, 14, 7
    k = synth.make_kkt(N, 2, 1)
    G, C, g, c = synth.pack_kkt_dense(k, np.float64)
    assert G.shape[1] == (n * n + m * m) * N - m * m          # include/pcg/sqp.cuh:43-46
    assert C.shape[1] == (n * n + n * m) * (N - 1)
    assert g.shape[1] == (n + m) * N - m and c.shape[1] == n * N
    # column-major blocks; C stores -A, -B
    np.testing.assert_array_equal(G[1, (n * n + m * m) * 2:(n * n + m * m) * 2 + n * n].reshape(n, n).T, k.Q[1, 2])
    np.testing.assert_array_equal(C[0, (n * n + n * m):(n * n + n * m) + n * n].reshape(n, n).T, -k.A[0, 1])
    np.testing.assert_array_equal(C[0, n * n:n * n + n * m].reshape(m, n).T, -k.Bm[0, 0])


def test_schur_shapes_and_poison():
    N = 4
    k = synth.make_kkt(N, 2, 9)
    S, P, g = synth.form_schur(k, poison_unused=True)
    assert S.shape == (2, 3 * 196 * N) and g.shape == (2, 14 * N) and S.dtype == np.float32
    S4 = S.reshape(2, N, 3, 196)
    assert np.isnan(S4[:, 0, 0]).all() and np.isnan(S4[:, -1, 2]).all()
    assert np.isfinite(S4[:, 0, 1:]).all() and np.isfinite(S4[:, 1:-1]).all()
    _, Pj, _ = synth.form_schur(k, precond="jacobi")
    P4 = Pj.reshape(2, N, 3, 196)
    assert (P4[:, :, 0] == 0).all() and (P4[:, :, 2] == 0).all()
    np.testing.assert_array_equal(P4[:, :, 1], np.nan_to_num(P.reshape(2, N, 3, 196))[:, :, 1])


def test_shard_range_partitions():
    for total in (0, 1, 7, 1024, 1027):
        for world in (1, 2, 3, 8):
            rs = [shard_range(total, r, world) for r in range(world)]
            assert rs[0][0] == 0 and rs[-1][1] == total
            assert all(rs[i][1] == rs[i + 1][0] for i in range(world - 1))
            sizes = [hi - lo for lo, hi in rs]
            assert max(sizes) - min(sizes) <= 1
