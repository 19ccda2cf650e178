"""world_size-2 and -4 gloo runs on the CPU: process grid, halo message plan and exchange routine of the
multi-GPU path (tests/dist_cpu_worker.py).  No GPU, no compute calls into the HIP library."""
import os
import subprocess
import sys

import pytest

from conftest import ROOT


def _run(world, dims, m, port):
    env = dict(os.environ, BCG_TEST_DIMS=",".join(map(str, dims)), BCG_TEST_M=str(m), OMP_NUM_THREADS="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}", "--master-addr",
           "127.0.0.1", "--master-port", str(port), os.path.join(ROOT, "tests", "dist_cpu_worker.py")]
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-3000:]
    assert "DIST_CPU_OK" in out.stdout


@pytest.mark.parametrize("world,dims,m,port", [(2, [4, 4, 2, 6], 3, 29611), (4, [6, 2, 4, 4], 2, 29612), (2, [8, 6], 4, 29613)])
def test_halo_plan_and_exchange_gloo(world, dims, m, port):
    import blockcg_amd
    if not os.path.exists(blockcg_amd.LIB_PATH):
        blockcg_amd.build()
    _run(world, dims, m, port)


def test_grid_and_plan_consistency():
    """Every send in the plan of one rank is matched by the mirrored receive in the plan of its peer."""
    from blockcg_amd.comm import coords_of, grid_for, halo_plan
    for world, dims in [(2, [8, 4, 4, 4]), (4, [8, 4, 4, 4]), (8, [4, 4, 4, 4]), (8, [16, 8, 8, 8])]:
        grid = grid_for(world, len(dims))
        assert int(__import__('numpy').prod(grid)) == world
        plans = {r: halo_plan(dims, grid, coords_of(r, grid), 48)[0] for r in range(world)}
        for r, msgs in plans.items():
            sends_to = {}
            for k, (ps, pr, so, ro, nb) in enumerate(msgs):
                sends_to.setdefault(ps, []).append((k, nb))
            for peer, lst in sends_to.items():
                recvs = [(k, nb) for k, (ps, pr, so, ro, nb) in enumerate(plans[peer]) if pr == r]
                assert [nb for _, nb in lst] == [nb for _, nb in recvs], (world, r, peer)
                # posting order per peer is what RCCL matches on: the k-th send to a peer pairs with its k-th receive
                assert [k for k, _ in lst] == [k for k, _ in recvs]
