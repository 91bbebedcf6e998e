"""CPU: the IIWA-14 producer side (stage A) — oracle/iiwa_ref.py (numpy float64 restatement of
include/common/kkt.cuh:22-163 + the plant it calls) PINNED on the reference's own trajectory pair (tests/golden/iiwa_traj_0_0_full.npz =
all 666 rows of examples/trajfiles/0_0_traj.csv and 0_0_eepos.traj: outputs of the reference's own kinematics AND forward dynamics), and
checked against the KKT / Schur fixtures tests/make_iiwa_golden.py produced in the build container."""
import os

import numpy as np
import pytest

from conftest import GOLDEN, ROOT
import iiwa_ref as iiwa
from mpcgpu_amd import synth


@pytest.fixture(scope="module")
def M():
    return iiwa.Model()


@pytest.fixture(scope="module")
def traj():
    d = np.load(os.path.join(GOLDEN, "iiwa_traj_0_0_full.npz"))
    return d["xu"], d["eepos"]


def test_product_trajectory_fixture_is_the_head_of_the_reference_file(traj):
    xu, eep = traj
    d = np.load(os.path.join(ROOT, "mpcgpu_amd", "data", "iiwa_traj_0_0.npz"))
    np.testing.assert_array_equal(d["xu"], xu[:400].astype(np.float32))
    np.testing.assert_array_equal(d["eepos"], eep[:400].astype(np.float32))


def test_end_effector_kinematics_reproduce_the_reference_fixture(M, traj):
    """Row t of 0_0_eepos.traj is the end-effector position of row t of 0_0_traj.csv (the reference generated one from the
    other with its GRiD kinematics): the restated forward kinematics must reproduce all 666 rows (csv precision ~1e-6)."""
    xu, eep = traj
    err = max(np.abs(M.ee_pos(xu[t, :7]) - eep[t, :3]).max() for t in range(xu.shape[0]))
    assert err < 2e-5, err
    # and the Jacobian is the derivative of that map
    q = xu[17, :7]
    J = M.ee_jac(q)
    d = 1e-5 * np.arange(1, 8)
    assert np.abs(M.ee_pos(q + d) - M.ee_pos(q) - J @ d).max() < 1e-8


def test_forward_dynamics_reproduce_the_reference_trajectory(M, traj):
    """THE PIN OF THE DYNAMICS ON REFERENCE-HELD DATA.  0_0_traj.csv was integrated by the reference's own plant:
    consecutive rows are one Euler step (include/common/integrator.cuh:56-104, dt = 1/64 — examples/track_iiwa_pcg.cu:19) of its forward
    dynamics (iiwa_eepos_plant.cuh:127-155, GRAVITY = 0) apart, so the integrator defect c_{k+1} = x_{k+1} - f(x_k, u_k) that
    generate_kkt_submatrices stores (include/common/kkt.cuh:117,160) must vanish on the file's own rows, to its print precision.
    The file is five point-to-point segments; the step ACROSS a seam is a jump to the next waypoint and the step OUT of a waypoint row
    leaves q where it was (the row's stored velocity — +-0.01..0.05 rad/s, zero torque — is a start-up perturbation that was not
    integrated: q defect = -dt qd exactly, qd defect <= 6e-5) — both identified, not masked: they are asserted to be exactly that."""
    xu, _ = traj
    dt = iiwa.TIMESTEP
    rows = xu.shape[0]
    good = iiwa.in_segment_transitions(rows)
    assert len(good) == 656
    d = np.array([np.abs(iiwa.euler_defect(M, xu[t, :14], xu[t, 14:], xu[t + 1, :14])).max() for t in range(rows - 1)])
    assert d[good].max() < 5e-6, d[good].max()                  # (measured: 1.4e-6 max — the file holds float32 values —, 5.8e-8 median)
    assert np.median(d[good]) < 1e-7
    for s0 in iiwa.SEGMENT_STARTS:
        # the step out of a waypoint row: q stays (defect = -dt * stored qd), the velocity defect is small
        df = iiwa.euler_defect(M, xu[s0, :14], xu[s0, 14:], xu[s0 + 1, :14])
        assert np.abs(df[:7] + dt * xu[s0, 7:14]).max() < 1e-6 and np.abs(df[7:]).max() < 1e-4, s0
        assert np.abs(xu[s0 + 1, :7] - xu[s0, :7]).max() < 1e-6
        if s0:
            assert d[s0 - 1] > 1e-2                              # the jump to the next waypoint (not dynamics)
    # the waypoint rows carry no torque and (nearly) no velocity: nothing about the dynamics is hidden in the 9 excluded transitions
    assert all(np.abs(xu[s0, 14:]).max() == 0.0 and np.abs(xu[s0, 7:14]).max() <= 0.05 + 1e-9 for s0 in iiwa.SEGMENT_STARTS)


def test_rigid_body_dynamics_identities(M, traj):
    xu, _ = traj
    q, qd, u = xu[40, :7], xu[40, 7:14], xu[40, 14:]
    Mm = M.mass_matrix(q)
    assert np.allclose(Mm, Mm.T) and np.linalg.eigvalsh(Mm).min() > 1e-4          # symmetric positive definite
    qdd, dq, dqd, du = M.forward_dynamics_and_gradient(q, qd, u)
    assert np.abs(M.rnea(q, qd, qdd) - u).max() < 1e-9                              # ID(FD(u)) = u
    # the gradient formula -Minv dID/dq|qdd equals the derivative of the forward dynamics itself
    fd = lambda q_, qd_, u_: M.forward_dynamics_and_gradient(q_, qd_, u_)[0]
    h = 1e-6
    for j in (0, 3, 6):
        e = np.zeros(7)
        e[j] = h
        assert np.abs((fd(q + e, qd, u) - fd(q - e, qd, u)) / (2 * h) - dq[:, j]).max() < 1e-4 * max(1.0, np.abs(dq).max())
        assert np.abs((fd(q, qd + e, u) - fd(q, qd - e, u)) / (2 * h) - dqd[:, j]).max() < 1e-4 * max(1.0, np.abs(dqd).max())
    assert np.abs(du - np.linalg.inv(Mm)).max() < 1e-10
    # kinetic energy form: qd^T ID(q, 0, qd_as_acc) = qd^T M qd
    assert abs(qd @ M.rnea(q, np.zeros(7), qd) - qd @ Mm @ qd) < 1e-10


def test_generate_kkt_matches_fixture_and_reference_structure(M, orc):
    d = np.load(os.path.join(GOLDEN, "iiwa_kkt_N32.npz"))
    N, n, m = 32, 14, 7
    for s in range(3):
        G, C, g, c = iiwa.generate_kkt(M, d[f"s{s}_xu"].astype(np.float64), d[f"s{s}_goals"].astype(np.float64), d[f"s{s}_xs"].astype(np.float64), N)
        for a, name in ((G, "G"), (C, "C"), (g, "g"), (c, "c")):
            want = d[f"s{s}_{name}"]
            assert np.abs(a - want).max() <= 2e-5 * max(1.0, np.abs(want).max()), (s, name)       # (inputs were rounded to float32 in the fixture)
        # structure of include/common/kkt.cuh / iiwa_eepos_plant.cuh:358-368: Q_k = blkdiag(g g^T, QD I), R_k = R_COST I, C = -A, -B
        Q0 = G[:196].reshape(14, 14).T
        assert np.linalg.matrix_rank(Q0[:7, :7], tol=1e-12) <= 1 and np.allclose(Q0[7:, 7:], iiwa.QD_COST * np.eye(7)) and np.allclose(Q0[:7, 7:], 0)
        assert np.allclose(G[196:196 + 49].reshape(7, 7), iiwa.r_cost(N) * np.eye(7))
        A0 = -C[:196].reshape(14, 14).T
        assert np.allclose(A0[:7, :7], np.eye(7)) and np.allclose(A0[:7, 7:], iiwa.TIMESTEP * np.eye(7))    # Euler: q+ = q + dt qd
        B0 = -C[196:196 + 98].reshape(7, 14).T
        assert np.allclose(B0[:7], 0) and np.linalg.eigvalsh(0.5 * (B0[7:] + B0[7:].T)).min() > 0          # dt Minv
        # the Schur system the oracle forms from the fixture's blocks is the stored one, and PCG behaves as recorded
        S, P, gam, _ = orc.form_schur(d[f"s{s}_G"].copy(), d[f"s{s}_C"], d[f"s{s}_g"], d[f"s{s}_c"], N, synth.RHO_INIT, ss=True)
        np.testing.assert_array_equal(np.nan_to_num(S), d[f"s{s}_S"])
        np.testing.assert_array_equal(np.nan_to_num(P), d[f"s{s}_Pinv"])
        r = orc.pcg(d[f"s{s}_S"], d[f"s{s}_Pinv"], d[f"s{s}_gamma"], np.zeros(n * N, np.float32), N, 5000, 1e-4, "ss")
        assert r["iters"] == int(d[f"s{s}_iters_ss_1e4"]) and r["iters"] <= synth.pcg_max_iter(N)          # inside the reference's cap of 173
        assert float(d[f"s{s}_cond"]) > 1e4                                                                # real systems are ill-conditioned


def _tampered(kind):
    import copy
    m = copy.deepcopy(iiwa.Model())
    if kind == "axis":                                       # a sin coefficient with the wrong sign: no longer Rz(q) X(0)
        t = next(i for i, e in enumerate(m.X_trig) if e[2] < iiwa.NJ)
        m.X_trig = list(m.X_trig)
        m.X_trig[t] = (m.X_trig[t][0], -m.X_trig[t][1], m.X_trig[t][2])
    elif kind == "xhom":                                     # link 3 is 1 cm longer in the homogeneous transforms only
        m.Xhom_const = np.array(m.Xhom_const, dtype=np.float64, copy=True)
        m.Xhom_const.reshape(-1)[3 * 16 + 12 + 1] += 0.01
    elif kind == "inertia":
        m.I = np.array(m.I, dtype=np.float64, copy=True)
        m.I[2, 0, 4] += 1e-3
    elif kind == "nonrigid":                                 # symmetric, but the mass block is not m * 1
        m.I = np.array(m.I, dtype=np.float64, copy=True)
        m.I[2, 3, 4] += 1e-3
        m.I[2, 4, 3] += 1e-3
    return m


@pytest.mark.parametrize("kind,rc,word", [("axis", -2, "z axis"), ("xhom", -1, "different chains"), ("inertia", -1, "symmetric"),
                                          ("nonrigid", -2, "rigid-body form")])
def test_plant_create_rejects_tables_the_kernel_cannot_use(kind, rc, word):
    """mpcg_plant_create checks what the device kernel assumes (kkt_plant.hip.h): joints rotate about their own z axis, the homogeneous
    transforms and the spatial transforms describe the same chain (the end effector comes out of the latter), symmetric inertias of the rigid-body form.
    The checks run on the host before any device call: no GPU needed."""
    from mpcgpu_amd import Plant, _lib
    with pytest.raises(_lib.MpcgError) as e:
        Plant(_tampered(kind), device=0)
    assert e.value.code == rc and word in str(e.value), e.value
