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
