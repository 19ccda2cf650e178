"""Domain-decomposed SBCGrQ on the GPU box: 2 and 4 ranks share the one GPU and exchange halos through
gloo; every rank checks its sub-lattice against the oracle on the whole lattice
(tests/dist_gpu_worker.py)."""
import os
import subprocess
import sys

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu

CASES = [
    # dims,            grid,          m,  generic
    ([32, 4, 4, 8], [1, 1, 1, 2], 16, False),   # k_hop4, x3 split
    ([32, 4, 4, 4], [2, 1, 1, 1], 16, False),   # k_hop4, x0 split (row-end ghost lanes, U_0 ghost)
    ([32, 4, 4, 4], [2, 2, 1, 1], 16, False),   # 4 ranks, x0 and x1 split
    ([16, 4, 8, 4], [1, 1, 2, 2], 16, False),   # 4 ranks, x2 and x3 split
    ([8, 8, 4, 4], [1, 2, 1, 1], 16, False),    # general stencil kernel (L0 not a multiple of the tile)
    ([16, 8, 4, 4], [1, 1, 2, 1], 8, False),    # m = 8: fast stencil + generic row kernels
    ([6, 4, 6], [1, 1, 2], 3, True),            # generic kernels, 3-D lattice
    ([8, 4, 4, 4], [2, 1, 1, 2], 32, False),    # m = 32
]


@pytest.mark.parametrize("dims,grid,m,generic", CASES, ids=lambda v: "x".join(map(str, v)) if isinstance(v, list) else str(v))
def test_domain_decomposed_solve(dims, grid, m, generic):
    world = 1
    for g in grid:
        world *= g
    env = dict(os.environ, BCG_TEST_DIMS=",".join(map(str, dims)), BCG_TEST_GRID=",".join(map(str, grid)),
               BCG_TEST_M=str(m), BCG_TEST_GENERIC="1" if generic else "0", OMP_NUM_THREADS="1",
               BCG_HOP_BLOCKS="8", BCG_HOP_PATCH="16,2,2")
    port = 29700 + (hash((tuple(dims), tuple(grid), m)) % 200)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}", "--master-addr",
           "127.0.0.1", "--master-port", str(port), os.path.join(ROOT, "tests", "dist_gpu_worker.py")]
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-3000:]
    assert "DIST_GPU_OK" in out.stdout
