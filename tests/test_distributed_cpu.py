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


def test_bench_selects_the_headline_ladder():
    """bench.py --gpus N without shape flags: N = 1 is 64^4 (BASELINE configs[2]); N = 2, 4, 8 keep 64^3 x 128 sites per
    GPU in capacity mode (ring 32: two 15-slice chunks, exchanges overlapped) on grids (1,1,2,1), (1,2,2,1), (2,2,2,1), so that N = 8 is the headline 128^4."""
    import importlib.util
    import os
    from conftest import ROOT
    from blockcg_amd.comm import coords_of, grid_for
    spec = importlib.util.spec_from_file_location("bench", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    assert bench.resolve_shape(1, None, None) == ([64, 64, 64, 64], 0, False)
    want = {2: ([1, 1, 2, 1], [64, 64, 128, 128]), 4: ([1, 2, 2, 1], [64, 128, 128, 128]), 8: ([2, 2, 2, 1], [128] * 4)}
    for n, (grid, gdims) in want.items():
        local, cap, ladder = bench.resolve_shape(n, None, None)
        assert (local, cap, ladder) == ([64, 64, 64, 128], 32, True)
        g = grid_for(n, 4, keep_last=cap > 0)
        assert g == grid and [l * x for l, x in zip(local, g)] == gdims
        assert sorted(tuple(coords_of(r, g)) for r in range(n)) == sorted(set(tuple(coords_of(r, g)) for r in range(n)))
    assert bench.resolve_shape(2, [16, 16, 16, 16], None) == ([16, 16, 16, 16], 0, False)  # explicit shape: no capacity mode


def test_bench_config_4_is_the_declared_wide_block_shape():
    """bench.py --config 4: BASELINE configs[4]'s width and shift count on the largest volume that fits a GPU."""
    import argparse
    import importlib.util
    import os
    from conftest import ROOT
    from blockcg_amd.comm import grid_for
    spec = importlib.util.spec_from_file_location("bench", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    for world, cap in ((1, 0), (8, 32)):
        a = bench.apply_config(argparse.Namespace(config=4, m=16, shifts=4, local_dims=None, capacity=None), world)
        assert (a.m, a.shifts, a.local_dims, a.capacity) == (32, 8, [64, 64, 64, 32], cap)
        local, capacity, ladder = bench.resolve_shape(world, a.local_dims, a.capacity)
        assert local == [64, 64, 64, 32] and capacity == cap and not ladder
        # 19 fields (X_s, P_s, Q = B, T, tmp) of 64^3 x 32 sites x 1536 B + links: under 288 GB
        assert (19 * 64 ** 3 * 32 * 1536 + 64 ** 3 * 32 * 576) < 260e9
    assert grid_for(8, 4, keep_last=True) == [2, 2, 2, 1]
    b = bench.apply_config(argparse.Namespace(config=None, m=16, shifts=4, local_dims=None, capacity=None), 1)
    assert (b.m, b.shifts, b.local_dims) == (16, 4, None)
    # --config 1: BASELINE configs[1] (32^4, m = 8, 1 shift), one GPU only
    c1 = bench.apply_config(argparse.Namespace(config=1, m=16, shifts=4, local_dims=None, capacity=None), 1)
    assert (c1.m, c1.shifts, c1.local_dims) == (8, 1, [32, 32, 32, 32])
    with pytest.raises(SystemExit):
        bench.apply_config(argparse.Namespace(config=1, m=16, shifts=4, local_dims=None, capacity=None), 8)
