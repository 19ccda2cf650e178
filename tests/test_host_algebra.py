"""The product's host-side m x m algebra (blockcg_amd/csrc/small_matrix.hpp: Cholesky, full-pivot inverse,
triangular inverse, products, row norms) against numpy, on the CPU; the reference does this with Eigen
(inc/fields.hpp:142, inc/block_solvers.hpp:142,153,155)."""
import os
import subprocess

import numpy as np
import pytest

from conftest import ROOT, rel_err


@pytest.fixture(scope="module")
def probe():
    out = os.path.join(ROOT, "examples", "_build")
    os.makedirs(out, exist_ok=True)
    exe = os.path.join(out, "small_matrix_probe")
    r = subprocess.run(["g++", "-std=c++14", "-O2", "-Wall", "-I", os.path.join(ROOT, "blockcg_amd", "csrc"),
                        os.path.join(ROOT, "tests", "cpp", "small_matrix_probe.cpp"), "-o", exe], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    return exe


@pytest.mark.parametrize("m", [1, 3, 8, 16, 32])
def test_small_matrix_against_numpy(probe, m):
    rng = np.random.default_rng(m)
    A = rng.uniform(-1, 1, (m, m)) + 1j * rng.uniform(-1, 1, (m, m))
    raw = subprocess.run([probe, str(m)], input=np.ascontiguousarray(A.T).tobytes(), capture_output=True).stdout
    d = np.frombuffer(raw, dtype=np.float64)
    mm = 2 * m * m
    mats = [d[k * mm:(k + 1) * mm].view(np.complex128).reshape(m, m).T for k in range(4)]
    R, inv, rinv, prod = mats
    rn, ok = d[4 * mm:4 * mm + m], d[4 * mm + m]
    G = A.conj().T @ A + m * np.eye(m)
    assert ok == 1.0
    assert np.allclose(np.tril(R, -1), 0) and np.all(np.diag(R).real > 0)
    assert rel_err(R.conj().T @ R, G) < 1e-13
    assert rel_err(R, np.linalg.cholesky(G).conj().T) < 1e-12
    assert rel_err(inv @ A, np.eye(m)) < 1e-10 * max(1.0, np.linalg.cond(A) / 1e3)
    assert rel_err(rinv @ R, np.eye(m)) < 1e-13
    assert rel_err(prod, A @ A.conj().T) < 1e-14
    assert rel_err(rn, np.linalg.norm(A, axis=1)) < 1e-14


@pytest.mark.parametrize("fixture,m", [("ref1d_v96_m5.npz", 5), ("ref1d_v48_m12.npz", 12), ("ref1d_v128_m3.npz", 3)])
def test_eigen_style_decomposition_members_of_the_dropin_matrix(fixture, m):
    """blockcg::cmatrix's fullPivLu().solve / inverse() and llt().matrixL().adjoint() -- the Eigen members the reference's
    own solver templates call (inc/block_solvers.hpp:31,36,73,142,166; inc/fields.hpp:142) -- against numpy and against what
    Eigen itself returned in the reference-generated fixtures (`inverse_M`, `chol_upper`)."""
    from conftest import load_golden
    out = os.path.join(ROOT, "examples", "_build")
    os.makedirs(out, exist_ok=True)
    exe = os.path.join(out, "eigen_members_probe")
    r = subprocess.run(["g++", "-std=c++11", "-O2", "-Wall", "-I", os.path.join(ROOT, "blockcg_amd", "include"),
                        os.path.join(ROOT, "tests", "cpp", "eigen_members_probe.cpp"), "-o", exe], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    g = load_golden(fixture)
    A, G = g["M"], g["hermitian_dot_YY"]
    rng = np.random.default_rng(m)
    B = rng.uniform(-1, 1, (m, m)) + 1j * rng.uniform(-1, 1, (m, m))
    col = lambda a: np.ascontiguousarray(np.asarray(a, dtype=np.complex128).T).tobytes()  # noqa: E731
    raw = subprocess.run([exe, str(m)], input=col(A) + col(B) + col(G), capture_output=True)
    assert raw.returncode == 0
    d = np.frombuffer(raw.stdout, dtype=np.float64)
    mm = 2 * m * m
    X, Ai, R = (d[k * mm:(k + 1) * mm].view(np.complex128).reshape(m, m).T for k in range(3))
    assert rel_err(A @ X, B) < 1e-12 and rel_err(X, np.linalg.solve(A, B)) < 1e-11
    assert rel_err(Ai, g["inverse_M"]) < 1e-12        # Eigen's fullPivLu().solve(Identity) in the reference's build
    assert rel_err(R, g["chol_upper"]) < 1e-13         # Eigen's llt().matrixL().adjoint()
    assert np.allclose(np.tril(R, -1), 0)
    assert rel_err(d[3 * mm:3 * mm + m], np.linalg.norm(A, axis=1)) < 1e-14
