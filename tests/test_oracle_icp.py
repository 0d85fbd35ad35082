"""ICP / matching part of the oracle on CPU: closed forms, numpy cross-checks, tie rules."""
import os

import numpy as np

from vtkcloudpoint_amd import synth

HERE = os.path.dirname(os.path.abspath(__file__))


def test_jacobi_on_the_reference_debug_matrix(oracle):
    """The only fixed input in the reference: the symmetric 3x3 of FrmMain.cs:2637-2640 (no expected output
    recorded there); its eigenvalues are 2 and 2 +- sqrt(5) analytically."""
    A = np.array([[1.0, 0, 2], [0, 2, 0], [2, 0, 3]])
    ev, V = oracle.jacobi_sym(A)
    assert np.allclose(np.sort(ev), [2 - np.sqrt(5), 2.0, 2 + np.sqrt(5)], atol=1e-13)
    assert np.allclose(A @ V, V * ev, atol=1e-13) and np.allclose(V.T @ V, np.eye(3), atol=1e-13)
    rng = np.random.default_rng(2)
    for _ in range(50):
        B = rng.standard_normal((4, 4))
        B = B + B.T
        ev, V = oracle.jacobi_sym(B)
        assert np.allclose(np.sort(ev), np.linalg.eigvalsh(B), atol=1e-12)


def test_calculate_rotation(oracle):
    """ICP.cs:274-285: unit quaternion (w,x,y,z) -> rotation matrix."""
    assert np.allclose(oracle.calc_rotation([1, 0, 0, 0]), np.eye(3))
    t = 0.3
    R = oracle.calc_rotation([np.cos(t / 2), 0, 0, np.sin(t / 2)])
    assert np.allclose(R, [[np.cos(t), -np.sin(t), 0], [np.sin(t), np.cos(t), 0], [0, 0, 1]])


def test_find_closest_ties_and_transform(oracle):
    model = np.array([[0.0, 0, 0], [2.0, 0, 0], [0.0, 2, 0], [2.0, 0, 0]])
    data = np.array([[1.0, 0, 0], [1.0, 1.0, 0], [2.0, 0.0, 0.0], [5.0, 5.0, 5.0]])
    assert oracle.find_closest(model, data).tolist() == [0, 0, 1, 1]  # strict <: lowest index wins
    R = synth.rotation_about((0, 0, 1), 90.0)
    P = oracle.trans_point(np.array([[1.0, 0, 0]]), R, [1, 2, 3])
    assert np.allclose(P, [[1, 3, 3]])


def test_horn_matches_kabsch_svd(oracle):
    rng = np.random.default_rng(4)
    for _ in range(20):
        P = rng.standard_normal((200, 3)) * 5
        R = synth.rotation_about(rng.standard_normal(3), float(rng.uniform(-60, 60)))
        t = rng.standard_normal(3)
        Y = P @ R.T + t
        s = np.zeros(16)
        s[0:3] = P.sum(0)
        s[3:6] = Y.sum(0)
        s[6:15] = (P.T @ Y).reshape(9)
        R1, T1 = oracle.horn_from_sums(s, len(P))
        # Kabsch
        H = (P - P.mean(0)).T @ (Y - Y.mean(0))
        U, _, Vt = np.linalg.svd(H)
        D = np.diag([1, 1, np.sign(np.linalg.det(Vt.T @ U.T))])
        Rk = Vt.T @ D @ U.T
        assert np.allclose(R1, Rk, atol=1e-10) and np.allclose(R1, R, atol=1e-10) and np.allclose(T1, t, atol=1e-9)


def test_icp_recovers_known_transform_and_fixture(oracle):
    c = synth.config_icp(nd=5000, nm=100, jitter=0.0)
    for rule in (oracle.STOP_SSE_DELTA, oracle.STOP_RMSE):
        r = oracle.icp(c["model"], c["data"], 1e-4, 100, rule)
        assert np.abs(r["R"] - c["R_true"]).max() < 1e-12 and np.abs(r["T"] - c["T_true"]).max() < 1e-10
        assert r["rmse"] < 1e-10 and r["iters"] <= 4
    g = np.load(os.path.join(HERE, "golden", "icp_5k.npz"))
    cj = synth.config_icp(nd=5000, nm=100, jitter=0.05)
    r = oracle.icp(cj["model"], cj["data"], 1e-4, 100, oracle.STOP_SSE_DELTA)
    assert np.array_equal(r["R"], g["R"]) and np.array_equal(r["T"], g["T"]) and r["iters"] == int(g["iters"])
    assert np.array_equal(oracle.find_closest(cj["model"], cj["data"]), g["nn0"])
    # the stop rule is |SSE - previous SSE| < e (ICP.cs:180): with jitter the RMSE floor is the noise level
    assert 0.08 < r["rmse"] < 0.09


def test_match(oracle):
    truths = np.array([[0.0, 0, 0], [10.0, 0, 0], [0.0, 0, 0]])
    centers = np.array([[0.1, 0, 0], [9.0, 0, 0], [50.0, 50, 50]])
    M = np.eye(4)
    M[0, 3] = 0.5  # shift x by 0.5
    r = oracle.match(centers, truths, M, 2.0)
    assert np.allclose(r["matched_xyz"][:, 0], [0.6, 9.5, 50.5])
    assert r["nearest"].tolist() == [0, 1, 1] and r["is_matched"].tolist() == [1, 1, 0] and r["count"] == 2
