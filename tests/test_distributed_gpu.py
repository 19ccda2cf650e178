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


def _mock_transport():
    """libblockcg_rccl_mock.so: the native transport (blockcg_amd/csrc/comm_rccl.cpp, unchanged) over a host-staged stand-in
    for the seven RCCL calls it makes (tests/cpp/mock_rccl.hpp), so that its ranks can share the one GPU of a test box."""
    lib = os.path.join(ROOT, "blockcg_amd", "_build", "libblockcg_rccl_mock.so")
    r = subprocess.run(["make", "-C", os.path.join(ROOT, "blockcg_amd", "csrc"), "-s", "mock"], capture_output=True, text=True)
    assert r.returncode == 0 and os.path.exists(lib), r.stderr[-2000:]
    return lib


def _sweep_mock_files():
    """The stand-in transport leaves the last collectives' files of a run in /dev/shm (a rank cannot know when the others
    have read them); every rank has exited by now."""
    import glob
    import shutil
    for d in glob.glob("/dev/shm/bcg_mock_*"):
        shutil.rmtree(d, ignore_errors=True)


def _run_ranks(dims, grid, m, generic, ring=0, blocks="8", patch="16,2,2", overlap=True, native=False, expect_ring_overlap=False,
               half=False, expect_checkerboard=False, half_chunk=0):
    world = 1
    for g in grid:
        world *= g
    env = dict(os.environ, BCG_TEST_DIMS=",".join(map(str, dims)), BCG_TEST_GRID=",".join(map(str, grid)),
               BCG_TEST_M=str(m), BCG_TEST_GENERIC="1" if generic else "0", BCG_TEST_RING=str(ring), OMP_NUM_THREADS="1",
               BCG_HOP_BLOCKS=blocks, BCG_HOP_PATCH=patch, BCG_TEST_OVERLAP="1" if overlap else "0")
    if native:
        env.update(BCG_TEST_TRANSPORT="native", BCG_RCCL_LIB=_mock_transport())
    if expect_ring_overlap:
        env.update(BCG_TEST_EXPECT_RING_OVERLAP="1")
    if half:  # (the oracle's converged whole-lattice solve only on the small lattices: dist_gpu_worker.half_volume_oracle)
        env.update(BCG_TEST_HALF="1", BCG_TEST_HALF_ORACLE_SOLVE="1" if dims[0] * dims[1] * dims[2] * (dims[3] if len(dims) > 3 else 1) <= 4096 else "0")
    if expect_checkerboard:
        env.update(BCG_TEST_EXPECT_CHECKERBOARD="1")
    if half_chunk:
        env.update(BCG_HALF_CHUNK=str(half_chunk), BCG_TEST_EXPECT_HALF_CHUNKED="1")
    port = 29700 + (hash((tuple(dims), tuple(grid), m, ring)) % 200)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}", "--master-addr",
           "127.0.0.1", "--master-port", str(port), os.path.join(ROOT, "tests", "dist_gpu_worker.py")]
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    if native:
        _sweep_mock_files()
    assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-3000:]
    assert "DIST_GPU_OK" in out.stdout


@pytest.mark.parametrize("dims,grid,m,generic", CASES, ids=lambda v: "x".join(map(str, v)) if isinstance(v, list) else str(v))
def test_domain_decomposed_solve(dims, grid, m, generic):
    _run_ranks(dims, grid, m, generic)


HALF_CASES = [
    # dims,            grid,          m,  generic, blocks, checkerboard bundle sweep expected, x3 chunk (0: whole sweeps, blocking exchanges)
    ([8, 4, 4, 8], [1, 1, 1, 2], 16, False, "8", False, 0),     # x3 divided, generic half-volume kernel (row shorter than a tile pair)
    ([8, 8, 4, 4], [2, 2, 1, 1], 8, False, "8", False, 0),      # 4 ranks, x0 (half faces compact in x1; generic kernel, declared) and x1
    ([64, 16, 16, 6], [1, 2, 2, 1], 16, False, "32", True, 0),  # checkerboard bundle sweep (compact row 32), ghost rows in x1 and x2, 4 ranks
    ([64, 16, 8, 12], [1, 2, 1, 1], 16, False, "32", True, 5),  # the ladder's form: x3 whole, swept in chunks of 5, 5, 2 slices, exchanges overlapped
    ([32, 16, 8, 4], [1, 2, 1, 1], 32, False, "32", True, 0),   # m = 32
]


@pytest.mark.parametrize("dims,grid,m,generic,blocks,cb,chunk", HALF_CASES,
                         ids=lambda v: "x".join(map(str, v)) if isinstance(v, list) else str(v))
def test_half_volume_fields_on_a_divided_lattice(dims, grid, m, generic, blocks, cb, chunk):
    """SURVEY 8f-4 on the ladder: half-volume (parity-compact) fields with half ghost faces -- the operator blocks and the
    two-half-solves solve of every rank's sites against the whole-lattice oracle (tests/dist_gpu_worker.py,
    half_volume_checks), generic kernel and checkerboard bundle sweep.  The native transport carries half fields in
    test_bare_bench_command_half_volume_option (x2 and x3 divided) and in the 8-rank thread rehearsal (x1, x2, x3)."""
    _run_ranks(dims, grid, m, generic, blocks=blocks, patch="16,2,2" if m == 16 else "8,2,2", half=True, expect_checkerboard=cb,
               half_chunk=chunk)


RING_CASES = [
    # dims,            grid,          m,  ring slices (x3 undivided; faces of the ring exchanged per chunk of ring-2 slices)
    ([32, 4, 4, 8], [2, 1, 1, 1], 16, 4),    # x0 split
    ([16, 8, 8, 6], [1, 2, 2, 1], 16, 3),    # 4 ranks, x1 and x2 split, one-slice chunks
    ([64, 4, 4, 8], [2, 1, 2, 1], 8, 8),     # m = 8, ring = L3
    ([16, 4, 4, 12], [2, 1, 1, 1], 32, 4),   # m = 32
]


COLUMN_CASES = [
    # local lattice 32x8x8x6 (m = 16) with 32 blocks and 16x2x2 patches: the column sweep with ghost faces in every direction
    # -- as one launch over 2 x 2 column bundles (k_hop4b, "whole") and as the interior + boundary classes of k_hop4c
    ([64, 8, 8, 6], [2, 1, 1, 1], 16, 0),     # x0 divided: edge lanes patched from the ghost face, U_0 ghost
    ([32, 16, 8, 6], [1, 2, 1, 1], 16, 0),    # x1
    ([32, 8, 16, 12], [1, 1, 2, 2], 16, 0),   # x2 and x3, 4 ranks
    ([64, 16, 8, 6], [2, 2, 1, 1], 16, 3),    # x0, x1 and capacity mode (one-slice windows: row form)
    ([64, 8, 8, 24], [2, 1, 1, 1], 16, 12),   # capacity ring 12: ten-slice windows, the bundle sweep with ring addressing and ghosts
]


@pytest.mark.parametrize("overlap", [False, True], ids=["whole", "interior+boundary"])
@pytest.mark.parametrize("dims,grid,m,ring", COLUMN_CASES, ids=lambda v: "x".join(map(str, v)) if isinstance(v, list) else str(v))
def test_domain_decomposed_solve_column_sweep(dims, grid, m, ring, overlap):
    """overlap: the split exchange, i.e. the interior and the boundary tile classes of the column-sweep kernel in two
    launches (capacity mode never splits)."""
    # capacity mode with the split callbacks: the per-chunk exchanges of the ring overlap the stencil work (ring >= 4)
    _run_ranks(dims, grid, m, False, ring, blocks="32", patch="16,2,2", overlap=overlap,
               expect_ring_overlap=bool(overlap and ring >= 4))


@pytest.mark.parametrize("dims,grid,m,ring", RING_CASES, ids=lambda v: "x".join(map(str, v)) if isinstance(v, list) else str(v))
def test_domain_decomposed_solve_capacity_mode(dims, grid, m, ring):
    """gloo transport with the split callbacks: rings of >= 4 slices take the overlapped form (chunks of (ring - 2) / 2 slices,
    exchange of chunk k posted while chunk k + 1's first stencil and chunk k - 1's second stencil run); ring 3 the serial one."""
    _run_ranks(dims, grid, m, False, ring, expect_ring_overlap=ring >= 4)


RING_OVERLAP_NATIVE = [
    # dims,              grid,          m,  ring, blocks
    ([64, 16, 8, 6], [2, 2, 1, 1], 16, 6, "32"),     # 4 ranks, two-slice chunks (3 of them), row form of the stencil
    ([32, 16, 8, 24], [1, 2, 1, 1], 16, 24, "32"),   # ring = L3 = 24: chunks of 11, 11, 2 slices -> bundle sweep windows
    ([32, 16, 8, 24], [1, 1, 2, 1], 16, 12, "32"),   # ring 12: five-slice chunks, last chunk short, the wrap slice restored from its copy
    ([16, 4, 4, 12], [2, 1, 1, 1], 32, 4, "8"),      # m = 32, one-slice chunks
]


@pytest.mark.parametrize("dims,grid,m,ring,blocks", RING_OVERLAP_NATIVE,
                         ids=lambda v: "x".join(map(str, v)) if isinstance(v, list) else str(v))
def test_capacity_mode_overlapped_exchanges_native_transport(dims, grid, m, ring, blocks):
    """The headline's mode with its exchanges overlapped, over the NATIVE transport's split form (second stream + events):
    per chunk pack -> halo_exchange_begin -> [first stencil of the next chunk] -> halo_exchange_end -> second stencil; the
    source's faces of slice 0 are put back from a device copy for the wrap slice instead of a second exchange.  Each rank
    checks operator, Gram matrix and solve against the whole-lattice oracle."""
    _run_ranks(dims, grid, m, False, ring, blocks=blocks, patch="16,2,2", overlap=True, native=True, expect_ring_overlap=True)


def test_rccl_on_library_memory_views():
    """Backend nccl (= RCCL) on the tensor views bench.py builds over library-owned device memory: an
    all-reduce and a batched self send/recv in a world of one (the only RCCL world a one-GPU box allows)."""
    code = r'''
import os, sys, torch, torch.distributed as dist
sys.path.insert(0, os.environ["BCG_ROOT"])
from blockcg_amd.comm import _DevMem, exchange_messages
torch.cuda.set_device(0)
dist.init_process_group("nccl", device_id=torch.device("cuda", 0))
base = torch.arange(4096, dtype=torch.float64, device="cuda")
view = torch.as_tensor(_DevMem(base.data_ptr(), base.numel() * 8, "<f8", 8), device="cuda")
assert view.data_ptr() == base.data_ptr()
dist.all_reduce(view)
torch.cuda.synchronize()
assert torch.equal(view, torch.arange(4096, dtype=torch.float64, device="cuda"))
send = torch.as_tensor(_DevMem(base.data_ptr(), 4096 * 8), device="cuda")
dst = torch.zeros(4096 * 8, dtype=torch.uint8, device="cuda")
recv = torch.as_tensor(_DevMem(dst.data_ptr(), 4096 * 8), device="cuda")
s = torch.cuda.Stream()
with torch.cuda.stream(s):
    exchange_messages(send, recv, [(0, 0, 0, 1024, 2048), (0, 0, 2048, 8192, 4096)], None, direct=True)
s.synchronize()
assert torch.equal(dst[1024:3072], send[0:2048]) and torch.equal(dst[8192:12288], send[2048:6144])
# split form used for the overlap: post, enqueue independent work, then make the stream wait
from blockcg_amd.comm import post_messages
dst.zero_()
with torch.cuda.stream(s):
    works = post_messages(send, recv, [(0, 0, 0, 4096, 1024)], None)
    filler = torch.ones(1 << 20, device="cuda").sum()
    for w in works:
        w.wait()
    got = dst[4096:5120].clone()
s.synchronize()
assert torch.equal(got, send[0:1024]) and float(filler) == float(1 << 20)
dist.destroy_process_group()
print("RCCL_VIEW_OK")
'''
    env = dict(os.environ, BCG_ROOT=ROOT, MASTER_ADDR="127.0.0.1", MASTER_PORT="29651", RANK="0", WORLD_SIZE="1",
               LOCAL_RANK="0")
    out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and "RCCL_VIEW_OK" in out.stdout, out.stdout[-2000:] + out.stderr[-3000:]


def test_bench_two_ranks_on_one_gpu():
    """bench.py's multi-rank path end to end (gloo, both ranks on GPU 0, small local volume)."""
    env = dict(os.environ, BCG_BACKEND="gloo", BCG_DEVICE="0", OMP_NUM_THREADS="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
           "--master-port", "29652", os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1",
           "--local-dims", "16", "16", "16", "16"]
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-3000:]
    import json
    line = [ln for ln in out.stdout.splitlines() if ln.startswith("{")][-1]
    d = json.loads(line)
    assert d["n_gpus"] == 2 and d["steps"] == 3 and d["value"] > 0 and d["config"]["global_dims"] == [16, 16, 16, 32]


def test_bench_ranks_on_one_gpu_native_transport():
    """bench.py's default transport path (BCG_BACKEND=rccl: unique id over the gloo control plane, blockcg_amd.rccl.RcclComm,
    capacity ring, x3 undivided process grid) with 4 ranks on GPU 0 over the host-staged stand-in."""
    env = dict(os.environ, BCG_BACKEND="rccl", BCG_RCCL_LIB=_mock_transport(), BCG_DEVICE="0", OMP_NUM_THREADS="1",
               BCG_HOP_BLOCKS="32", BCG_HOP_PATCH="16,2,2")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=4", "--master-addr", "127.0.0.1",
           "--master-port", "29653", os.path.join(ROOT, "bench.py"), "--gpus", "4", "--steps", "3", "--warmup", "1",
           "--local-dims", "32", "8", "8", "8", "--capacity", "4"]
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    _sweep_mock_files()
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-3000:]
    import json
    d = json.loads([ln for ln in out.stdout.splitlines() if ln.startswith("{")][-1])
    assert d["n_gpus"] == 4 and d["config"]["transport"] == "rccl" and d["capacity_ring_slices"] == 4
    assert d["config"]["process_grid"] == [1, 2, 2, 1] and d["config"]["global_dims"] == [32, 16, 16, 8] and d["value"] > 0


@pytest.mark.parametrize("gpus", [2, 4])
def test_bare_bench_command_starts_its_own_ranks(gpus):
    """The command line the round-end driver types -- `python bench.py --gpus N ...`, no launcher, WORLD_SIZE unset: the
    parent starts the N ranks as fresh children before touching a GPU (tests/test_bench_launcher.py checks the parent on the
    CPU) and relays rank 0's line.  Here with real ranks: native transport over the host-staged stand-in, all on GPU 0,
    capacity ring (the headline's mode)."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(BCG_BACKEND="rccl", BCG_RCCL_LIB=_mock_transport(), BCG_DEVICE="0", OMP_NUM_THREADS="1", BCG_HOP_BLOCKS="32",
               BCG_HOP_PATCH="16,2,2", BCG_BENCH_TIMEOUT="500")
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(gpus), "--steps", "3", "--warmup", "1",
           "--local-dims", "32", "8", "8", "8", "--capacity", "4"]
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    _sweep_mock_files()
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-3000:]
    import json
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    d = json.loads(lines[0])
    grid = {2: [1, 1, 2, 1], 4: [1, 2, 2, 1]}[gpus]
    assert d["n_gpus"] == gpus and d["steps"] == 3 and d["warmup"] == 1 and d["value"] > 0
    assert d["config"]["transport"] == "rccl" and d["capacity_ring_slices"] == 4 and d["config"]["process_grid"] == grid
    assert d["config"]["rccl_communicators"] == 2  # the split exchange runs on a communicator of its own
    assert d["config"]["global_dims"] == [32 * grid[0], 8 * grid[1], 8 * grid[2], 8]
    comm = d["comm_ms_per_iteration"]
    assert comm and comm["allreduce"] > 0 and comm["pack_faces"] > 0 and any(k.startswith("halo_exchange") for k in comm)
    assert d["roofline"] and 0 < d["roofline"]["frac"] <= 1 and d["device_bytes_in_use"] > 0


def test_bare_bench_command_half_volume_option():
    """`python bench.py --gpus 4 --half`: two half-volume solves per step on the ladder's grid for half fields (x0, the
    direction they are compact in, and x3, swept in chunks with overlapped exchanges, undivided: 4 ranks = (1,2,2,1)),
    native transport over the stand-in, checkerboard bundle sweep with ghost rows, chunks of 2 slices."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(BCG_BACKEND="rccl", BCG_RCCL_LIB=_mock_transport(), BCG_DEVICE="0", OMP_NUM_THREADS="1", BCG_HOP_BLOCKS="32",
               BCG_HOP_PATCH="16,2,2", BCG_BENCH_TIMEOUT="500", BCG_HALF_CHUNK="2")
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "4", "--half", "--steps", "4", "--warmup", "1",
           "--local-dims", "64", "8", "8", "6"]
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    _sweep_mock_files()
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-3000:]
    import json
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 4 and d["steps"] == 4 and d["value"] > 0 and d["capacity_ring_slices"] == 0
    assert d["config"]["process_grid"] == [1, 2, 2, 1] and d["config"]["global_dims"] == [64, 16, 16, 6]
    assert "two half-volume solves" in d["config"]["workload"]
    # per operator application: tmp[L3-1], then three chunks of two slices: 4 launches of the first stencil, 3 of the second
    assert d["stencil_kernel_launches"].get("k_hop4b_checkerboard", 0) == 7 * 2 * 4  # x two parities x 4 iterations
    comm = d["comm_ms_per_iteration"]
    assert comm and comm["allreduce"] > 0 and comm["pack_faces"] > 0 and comm["halo_exchange_begin"] > 0
    assert "halo_exchange" not in comm  # every exchange of the timed region in the split form


def test_bare_bench_command_strong_scaling_ladder():
    """`python bench.py --gpus 4 --ladder strong`: the lattice stays what it is at N = 1 and is divided over the ranks
    (process grid (1,1,2,2): x3 divided, so no ring -- the split exchange overlaps the interior tiles instead), `scaling`
    says strong.  Four ranks on GPU 0 over the native transport's stand-in, on a rehearsal-sized lattice."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(BCG_BACKEND="rccl", BCG_RCCL_LIB=_mock_transport(), BCG_DEVICE="0", OMP_NUM_THREADS="1", BCG_HOP_BLOCKS="32",
               BCG_HOP_PATCH="16,2,2", BCG_BENCH_TIMEOUT="500", BCG_BENCH_STRONG_DIMS="32,8,16,16")
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "4", "--ladder", "strong", "--steps", "4", "--warmup", "1"]
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    _sweep_mock_files()
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-3000:]
    import json
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["scaling"] == "strong" and d["n_gpus"] == 4 and d["capacity_ring_slices"] == 0 and d["value"] > 0
    assert d["config"]["global_dims"] == [32, 8, 16, 16] and d["config"]["process_grid"] == [1, 1, 2, 2]
    assert "strong-scaling ladder" in d["config"]["workload"] and len(d["config"]["memory_ladder"]) == 1
    comm = d["comm_ms_per_iteration"]
    assert comm and comm["allreduce"] > 0 and comm["halo_exchange_begin"] > 0  # the split exchange overlaps the interior tiles


@pytest.mark.parametrize("how", ["begin_fails_on_one_rank", "free_memory_too_small"])
def test_bare_bench_command_steps_down_the_memory_ladder(how):
    """`python bench.py --gpus 4 --step-down` with less memory than the first rungs need (BCG_DEBUG_FIELD_BUDGET, the
    library's stand-in for a full device), native transport over the stand-in, ranks sharing GPU 0.
    begin_fails_on_one_rank: ONLY rank 2 is short and the free-memory check is told to trust what it reads, so it is
    bcg_sbcgrq_begin that finds out, on one rank -- the ranks must agree before their first collective (no rank left waiting
    in thinQR's all-reduce), every rank abandons ring 16, then ring 8, and the half-volume form runs in the same processes.
    free_memory_too_small: every rank reads too little free memory: the ring rungs are skipped without allocating."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(BCG_BACKEND="rccl", BCG_RCCL_LIB=_mock_transport(), BCG_DEVICE="0", OMP_NUM_THREADS="1", BCG_HOP_BLOCKS="32",
               BCG_HOP_PATCH="16,2,2", BCG_BENCH_TIMEOUT="500", BCG_HALF_CHUNK="4", BCG_BENCH_RESERVE=str(8 << 20))
    field = 32 * 8 * 8 * 16 * 768  # one full field at m = 16: 25.2 MB; a ring rung holds 10 of them, the half-volume form 11 halves
    if how == "begin_fails_on_one_rank":
        env.update(BCG_DEBUG_FIELD_BUDGET=str(int(7.5 * field)), BCG_DEBUG_BUDGET_ONLY_RANK="2", BCG_BENCH_TRUST_FREE="1")
    else:
        env.update(BCG_DEBUG_FIELD_BUDGET=str(int(12.0 * field)))  # plans incl. links, faces, scratch: 14.3, 13.8 and 9.7 fields
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "4", "--steps", "4", "--warmup", "1",
           "--local-dims", "32", "8", "8", "16", "--capacity", "16", "--step-down"]
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    _sweep_mock_files()
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-3000:]
    import json
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    d = json.loads(lines[0])
    notes = d["config"]["memory_ladder"]
    assert [n["rung"] for n in notes] == ["ring 16", "ring 8", "half-volume"] and notes[-1]["verdict"] == "ran"
    want = "abandoned" if how == "begin_fails_on_one_rank" else "skipped"
    assert all(n["verdict"].startswith(want) for n in notes[:2]), notes
    assert d["config"]["memory_ladder_rung"] == "half-volume" and d["config"]["half_volume_solves"] and d["capacity_ring_slices"] == 0
    assert "NOT asked for by a flag" in d["config"]["workload"] and d["config"]["global_dims"] == [32, 16, 16, 16]
    assert d["n_gpus"] == 4 and d["steps"] == 4 and d["value"] > 0 and d["residual_after_timed_steps"] < 1.0
    assert "memory ladder: running half-volume" in out.stderr
    if how == "begin_fails_on_one_rank":
        assert "another rank of the process grid could not allocate" in out.stderr  # what ranks 0, 1, 3 were told
        assert "rank 2: BCG_ERR_HIP" in out.stderr                                    # ... and rank 2 itself


NATIVE_CASES = [
    # dims,             grid,          m,  ring, overlap (split exchange: second stream + events), blocks
    ([32, 4, 4, 8], [1, 1, 1, 2], 16, 0, True, "8"),     # x3 split over two ranks: + and - neighbour are the same peer
    ([64, 8, 8, 6], [2, 1, 1, 1], 16, 0, False, "32"),   # x0 split, blocking exchange, bundle sweep with ghost halos
    ([32, 16, 8, 6], [1, 2, 2, 1], 16, 0, True, "32"),   # 4 ranks: two split directions, interior + boundary launches
    ([64, 16, 8, 6], [2, 2, 1, 1], 16, 3, False, "32"),  # the headline's mode: capacity ring, x3 undivided, windowed exchanges
    ([8, 4, 4, 4], [2, 1, 1, 2], 32, 0, True, "8"),      # m = 32, general stencil
    ([32, 16, 8, 24], [1, 2, 1, 1], 16, 12, False, "32"),  # ring 12 (ten-slice windows): bundle sweep, ring addressing, x1 ghosts
]


@pytest.mark.parametrize("dims,grid,m,ring,overlap,blocks", NATIVE_CASES,
                         ids=lambda v: "x".join(map(str, v)) if isinstance(v, list) else str(v))
def test_native_transport_multi_rank(dims, grid, m, ring, overlap, blocks):
    """The NATIVE bcg_comm implementation (comm_rccl.cpp: one ncclGroup of sends/receives per exchange, posting-order
    matching of two messages to one peer, begin/end on a second stream with events, all-reduce of the Gram buffer) run by
    2 and 4 processes sharing the GPU, with RCCL's seven calls replaced by a host-staged stand-in that keeps their matching
    rules.  Every rank checks operator, Gram matrix and solve of its sub-lattice against the whole-lattice oracle."""
    _run_ranks(dims, grid, m, False, ring, blocks=blocks, patch="16,2,2", overlap=overlap, native=True)


THREAD_CASES = [
    # global dims,       grid,          m,  ring
    ([32, 16, 8, 32], [2, 2, 2, 1], 16, 32),   # the headline's launch: ring 32 = L3, overlapped chunks of 15, 15 and 2 slices
    ([32, 16, 8, 24], [2, 2, 2, 1], 16, 0),    # whole tmp: interior + boundary launches with three divided directions
    ([64, 8, 8, 12], [1, 2, 2, 2], 16, -1),    # half-volume fields with x3 divided too: ghost rows of the checkerboard sweep in x1, x2, x3
    ([64, 8, 16, 12], [1, 2, 4, 1], 16, -5),  # `bench.py --gpus 8 --half`: its grid (x0, x3 whole), x3 in chunks of 5, 5, 2, exchanges overlapped
]


@pytest.mark.parametrize("dims,grid,m,ring", THREAD_CASES, ids=lambda v: "x".join(map(str, v)) if isinstance(v, list) else str(v))
def test_headline_process_grid_eight_ranks_as_threads(dims, grid, m, ring):
    """BASELINE configs[3]'s process grid -- 8 ranks on (2,2,2,1): three divided directions, each two ranks wide, so every
    rank exchanges two messages with the same peer in all three (posting-order matching) -- with `bench.py --gpus 8`'s
    capacity ring (32 slices, overlapped chunks of C = 15: bundle-sweep windows with ring addressing and ghost faces in x0,
    x1 and x2).  Eight GPU processes exceed what a one-GPU box admits (six), so the ranks are threads of one process over
    the native transport's stand-in in its synchronous mode (tests/dist_threads_worker.py); each rank checks operator, Gram
    matrix and a fixed-work solve against the whole-lattice oracle.  The bare 8-process launch of bench.py itself is
    rehearsed without GPUs in tests/test_bench_launcher.py::test_bare_headline_command_plans_eight_ranks."""
    half = ring < 0  # full-volume checks without a ring, then the half-volume fields' (checkerboard bundle sweep with ghost
    chunk = -ring if ring < -1 else 0  # rows, half faces, two half solves against the oracle's full solve); < -1: x3 chunk
    ring = max(ring, 0)
    env = dict(os.environ, BCG_TEST_DIMS=",".join(map(str, dims)), BCG_TEST_GRID=",".join(map(str, grid)), BCG_TEST_M=str(m),
               BCG_TEST_RING=str(ring), OMP_NUM_THREADS="1", BCG_HOP_BLOCKS="32", BCG_HOP_PATCH="16,2,2", BCG_MOCK_SYNC="1",
               BCG_RCCL_LIB=_mock_transport())
    if half:  # (the half solves are judged by their true residuals here; against the oracle's own converged solve on the
        # small lattices of test_half_volume_fields_on_a_divided_lattice -- minutes of host CPU at these sizes)
        env.update(BCG_TEST_HALF="1", BCG_TEST_EXPECT_CHECKERBOARD="1", BCG_TEST_HALF_ORACLE_SOLVE="0")
    if chunk:
        env.update(BCG_HALF_CHUNK=str(chunk), BCG_TEST_EXPECT_HALF_CHUNKED="1")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "dist_threads_worker.py")], env=env, capture_output=True,
                         text=True, timeout=1200)
    _sweep_mock_files()
    assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-6000:]
    assert "DIST_THREADS_OK 8" in out.stdout
    if ring:
        assert "k_hop4b" in out.stdout  # the 15-slice windows ran the bundle sweep
