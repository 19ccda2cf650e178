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
