"""pytest configuration: markers, shared fixtures, tolerances."""
import glob
import os
import sys

import numpy as np
import pytest
import torch  # noqa: F401  -- FIRST, before libblockcg_hip.so is loaded: see "HIP runtime load order" below

# HIP runtime load order.  This image's torch wheel bundles its own libamdhip64.so / libhsa-runtime64.so / librccl.so
# (unversioned SONAMEs) next to the system ROCm ones the product links (libamdhip64.so.7, librccl.so.1).  Two ROCr
# instances cannot both drive the GPU from one process: whichever initialises second sees "No HIP GPUs are available".
# torch loads its bundle with RTLD_GLOBAL, so when torch is imported BEFORE the product library the library's HIP and RCCL
# symbols bind to the bundle and the process has one runtime (this is what bench.py does).  Tests that use torch tensors
# next to the library (halo buffer views, RCCL) therefore need this import order; the product itself never imports torch.

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN_DIR = os.path.join(ROOT, "tests", "golden")

# fp64 tolerances, stated once and used everywhere
TOL_KERNEL = 1e-13      # relative Frobenius error of one primitive vs the reference / oracle
TOL_COEFF = 1e-10       # m x m coefficient matrices of the first iterations
TOL_SOLUTION = 1e-8     # final X vs reference at well-conditioned configurations


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def rel_err(a, b):
    a = np.asarray(a)
    b = np.asarray(b)
    nb = np.linalg.norm(b.ravel())
    return np.linalg.norm((a - b).ravel()) / (nb if nb > 0 else 1.0)


def golden_files(pattern="*.npz"):
    return sorted(glob.glob(os.path.join(GOLDEN_DIR, pattern)))


def load_golden(name):
    return np.load(os.path.join(GOLDEN_DIR, name))


@pytest.fixture(scope="session")
def orc():
    import oracle
    return oracle.Oracle()


# ---- an evaluation of the n-D operator that shares nothing with the repository's n-D formula -----------------------------
# The reference has a 1-D operator only (inc/dirac_op.hpp:14-21).  With the links of every direction but mu set to zero the
# n-D operator of DESIGN.md section 1 acts on each lattice line parallel to axis mu as eta times THAT 1-D operator with the
# line's links, and by linearity the full operator is the sum over mu.  reference_D_line is the reference's loop written in
# numpy for one line (pinned against the reference's own 1-D fixture by tests/test_oracle_golden.py); hop_by_lines gathers
# the lines by explicit coordinates -- no site-index strides, no neighbour tables, no eta expression shared with
# oracle/oracle.hpp or the kernels beyond "the parity of x_0 + ... + x_{mu-1}".
def reference_D_line(U, psi):
    """inc/dirac_op.hpp:17-20 on ONE periodic line: U [V, 3, 3] in the host layout ([x, k, r] = U_x(r, k)), psi [V, m, 3]
    ([x, j, c]).  Returns 0.5 * (U[x] psi[x+1] - U[x-1]^dagger psi[x-1])."""
    V = psi.shape[0]
    out = np.zeros_like(psi)
    for x in range(V):
        Ux = U[x].T                      # 3 x 3 matrix U_x(r, k)
        Ub = U[(x - 1 + V) % V].T
        fwd = psi[(x + 1) % V]           # [m, 3] = (3 x m matrix)^T
        bwd = psi[(x - 1 + V) % V]
        out[x] = 0.5 * (fwd @ Ux.T - bwd @ np.conj(Ub))   # (U psi)^T = psi^T U^T ; (U^dagger psi)^T = psi^T conj(U)
    return out


def hop_by_lines(U, dims, psi):
    """(D psi)(x) = sum_mu eta_mu(x) [reference 1-D D along the line through x parallel to mu], eta_mu = (-1)^(x_0+..+x_{mu-1}).
    U [V, ndim, 3, 3], psi [V, m, 3], sites lexicographic with x_0 fastest (the ABI's host layouts)."""
    import itertools
    nd = len(dims)
    V = int(np.prod(dims))
    coords = np.array(list(itertools.product(*[range(d) for d in reversed(dims)])))[:, ::-1]  # row s = coordinates of site s
    assert coords.shape == (V, nd) and all(np.array_equal(coords[k], [k] + [0] * (nd - 1)) for k in range(min(dims[0], 2)))
    index = {tuple(c): s for s, c in enumerate(coords)}
    out = np.zeros_like(psi)
    for mu in range(nd):
        if dims[mu] == 1:
            continue
        others = [range(d) for nu, d in enumerate(dims) if nu != mu]
        for rest in itertools.product(*others):
            line = []
            for xm in range(dims[mu]):
                c = list(rest)
                c.insert(mu, xm)
                line.append(index[tuple(c)])
            line = np.array(line)
            eta = -1.0 if sum(coords[line[0]][:mu]) % 2 else 1.0
            out[line] += eta * reference_D_line(U[line, mu], psi[line])
    return out
