"""`python bench.py --gpus N` as the driver types it (no launcher around it, WORLD_SIZE unset): the parent must start the
ranks as fresh children before it has imported torch or the library, relay the line, propagate failures and never leave
ranks behind.  CPU-only here; the same bare command with real ranks on a GPU is tests/test_distributed_gpu.py."""
import os
import subprocess
import sys
import time

import pytest

from conftest import ROOT

BENCH = os.path.join(ROOT, "bench.py")


def _clean_env(**extra):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(OMP_NUM_THREADS="1", **extra)
    return env


def test_parent_imports_neither_torch_nor_the_library():
    code = (
        "import sys, os\n"
        f"sys.path.insert(0, {ROOT!r})\n"
        "import bench\n"
        "bad = [m for m in sys.modules if m == 'torch' or m.startswith('torch.') or m.startswith('blockcg_amd') or m == 'oracle']\n"
        "assert not bad, bad\n"
        "maps = open('/proc/self/maps').read()\n"
        "assert 'libamdhip64' not in maps and 'libblockcg' not in maps and 'libhsa-runtime' not in maps\n"
        "assert bench.descendants(os.getpid()) == []\n"
        "print('PARENT_CLEAN')\n")
    out = subprocess.run([sys.executable, "-c", code], env=_clean_env(), capture_output=True, text=True, timeout=120)
    assert out.returncode == 0 and "PARENT_CLEAN" in out.stdout, out.stdout + out.stderr


@pytest.mark.skipif(os.path.exists("/dev/kfd"), reason="needs a machine WITHOUT a GPU: the ranks must fail")
def test_bare_multi_gpu_command_propagates_rank_failure():
    """No GPU in this container: every rank fails at context creation (the product has no CPU fallback) and the bare
    command must come back non-zero, promptly, with the reason on stderr -- not hang, not rc 0."""
    t0 = time.time()
    out = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--steps", "1", "--warmup", "0", "--local-dims", "8", "4", "4", "4",
                          "--no-cpu-baseline"], env=_clean_env(BCG_BACKEND="gloo"), capture_output=True, text=True, timeout=600)
    assert out.returncode != 0
    assert not [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert "no usable HIP device" in out.stderr or "No HIP GPUs" in out.stderr or "BCG_ERR_NO_DEVICE" in out.stderr, out.stderr[-3000:]
    assert time.time() - t0 < 300


def test_bare_multi_gpu_command_kills_hung_ranks():
    """Ranks that never finish: after BCG_BENCH_TIMEOUT the parent terminates the launcher and kills exactly what it started.
    (How many processes exist by then depends on how fast this machine imports torch; whatever was started must be gone.)"""
    env = _clean_env(BCG_BENCH_TEST_HANG="1", BCG_BENCH_TIMEOUT="20")
    proc = subprocess.Popen([sys.executable, BENCH, "--gpus", "2", "--steps", "1", "--warmup", "0", "--local-dims", "8", "4", "4", "4"],
                            env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
    sys.path.insert(0, ROOT)
    import bench
    started = set()
    t0 = time.time()
    while time.time() - t0 < 18 and proc.poll() is None:  # sample the process tree until just before the timeout fires
        started.update(bench.descendants(proc.pid))
        time.sleep(0.5)
    assert len(started) >= 1, started  # at least the launcher; normally the launcher and both ranks
    out, err = proc.communicate(timeout=180)
    assert proc.returncode != 0 and "did not finish" in err, err[-2000:]
    time.sleep(0.5)
    alive = [p for p in started if os.path.exists(f"/proc/{p}") and "Z" not in open(f"/proc/{p}/stat").read().split(")")[-1].split()[0]]
    assert not alive, alive


def _bench_module():
    import importlib.util
    spec = importlib.util.spec_from_file_location("bench", BENCH)
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    return bench


def test_roofline_picks_the_bound_that_binds_the_dominant_kernel(tmp_path):
    """bench.roofline_of: the kernel class with the most time, priced against the roof its arithmetic intensity puts it
    under (flops per byte against the ridge of the two peaks); PMC traffic is quoted only for the shape it was measured at;
    a fraction above 1 is flagged (`accounting_suspect`) and never keeps the line from printing."""
    import json
    import pytest
    bench = _bench_module()
    prof = {
        "hop": {"ms": 100.0, "count": 10, "bytes": 10 * 35.4e9, "flops": 10 * 0.15e12},
        "phaseC_multi4": {"ms": 175.0, "count": 3, "bytes": 3 * 257.7e9, "flops": 3 * 3.09e12},
        "stencil_form_k_hop4b": {"ms": 0.0, "count": 10, "bytes": 0.0},
        "halo_exchange": {"ms": 5.0, "count": 10, "bytes": 0.0},
    }
    tj = tmp_path / "traffic.json"
    tj.write_text(json.dumps({"phaseC_multi4": {"bytes_per_launch": 257.7e9}, "hop": {"bytes_per_launch": 46.8e9},
                              "_shape": {"local_dims": [64] * 4, "m": 16, "n_shifts": 4, "capacity": 0}}))
    r = bench.roofline_of(prof, [64] * 4, 16, 4, 0, 1, traffic_path=str(tj))
    assert r["kernel"] == "phaseC_multi4" and r["bound"] == "mfma" and r["unit"] == "TFLOP/s" and r["peak"] == bench.FP64_PEAK_TFLOPS
    assert r["achieved"] == pytest.approx(3.09e12 / (175.0 / 3 * 1e-3) / 1e12) and r["frac"] == pytest.approx(r["achieved"] / 78.6)
    assert r["hbm_frac"] == pytest.approx(257.7e9 / (175.0 / 3 * 1e-3) / 1e9 / 8000.0) and r["hbm_frac"] < r["frac"]
    assert r["intensity_flop_per_byte"] == pytest.approx(3.09e12 / 257.7e9) and r["intensity_flop_per_byte"] > r["ridge_flop_per_byte"]
    assert r["ridge_flop_per_byte"] == pytest.approx(78.6e12 / 8e12) and r["accounting_suspect"] is False
    # the choice follows the intensity, not the larger fraction: a launch under the ridge is an HBM kernel even when its
    # flop fraction comes out larger (mis-priced flops would otherwise flatter the line)
    low = dict(prof, phaseC_multi4={"ms": 175.0, "count": 3, "bytes": 3 * 257.7e9, "flops": 3 * 2.4e12})
    r2 = bench.roofline_of(low, [64] * 4, 16, 4, 0, 1, traffic_path=str(tj))
    assert r2["intensity_flop_per_byte"] < r2["ridge_flop_per_byte"] and r2["bound"] == "hbm" and r2["unit"] == "GB/s"
    assert r["traffic"] == 257.7e9 and r["stencil_traffic_ratio"]["hop"] == pytest.approx(46.8 / 35.4)
    assert set(r["per_kernel_frac"]) == {"hop", "phaseC_multi4"}  # entries without bytes are not kernels
    # another shape: the counters are not quoted
    assert bench.roofline_of(prof, [64, 64, 64, 128], 16, 4, 32, 1, traffic_path=str(tj))["traffic"] is None
    # a memory-bound dominant kernel is priced against HBM
    prof["phaseC_multi4"] = {"ms": 58.0, "count": 1, "bytes": 257.7e9, "flops": 3.09e12}
    r = bench.roofline_of(prof, [64] * 4, 16, 4, 0, 1, traffic_path=str(tj))
    assert r["kernel"] == "hop" and r["bound"] == "hbm" and r["unit"] == "GB/s" and r["frac"] == pytest.approx(35.4e9 / 10e-3 / 1e9 / 8000.0)
    # more bytes than the peak could move in the measured time (a cache-resident shape, or an accounting error): flagged
    prof["hop"]["bytes"] = 10 * 90e9
    r = bench.roofline_of(prof, [64] * 4, 16, 4, 0, 1, traffic_path=str(tj))
    assert r["accounting_suspect"] is True and r["per_kernel_frac"]["hop"] > 1.0
    assert bench.roofline_of({}, [64] * 4, 16, 4, 0, 1) is None


def test_half_volume_ladder_plans_eight_ranks():
    """`python bench.py --gpus 8 --half --plan-only`: the declared half-volume ladder -- 128^4 as two half solves per GPU
    share of 128 x 64 x 32 x 128 sites on a (1,2,4,1) grid (x0, which half fields are compact in, and x3, swept in chunks
    with overlapped exchanges, stay whole), no ring, shift updates grouped over four iterations, half the bytes per face."""
    import json
    out = subprocess.run([sys.executable, BENCH, "--gpus", "8", "--half", "--plan-only"], env=_clean_env(), capture_output=True,
                         text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-3000:]
    d = json.loads([ln for ln in out.stdout.splitlines() if ln.startswith("{")][-1])
    assert d["plan_only"] and d["n_gpus"] == 8 and d["config"]["process_grid"] == [1, 2, 4, 1]
    assert d["config"]["global_dims"] == [128] * 4 and d["config"]["half_volume_solves"]
    assert d["capacity_ring_slices"] == 0 and d["shift_group_depth"] == 4
    assert 185e9 < d["device_bytes_planned"] < 200e9  # 13 half fields of 12.9 GB + full links + faces: 284 GB in capacity mode
    for r in d["ranks"]:
        c = r["coords"]
        assert c[0] == 0 and c[3] == 0 and r["rank"] == c[1] + 2 * c[2] and len(r["messages"]) == 4
        f1, f2 = 128 * 32 * 128, 128 * 64 * 128  # sites of a face of direction 1, 2
        assert [mm["bytes"] for mm in r["messages"]] == [f1 * 384, f1 * 384, f2 * 384, f2 * 384]  # 768 B per site, halved
    for a in d["ranks"]:
        for k, mm in enumerate(a["messages"]):
            b = d["ranks"][mm["send_to"]]
            assert b["messages"][k]["recv_from"] == a["rank"] and b["messages"][k]["bytes"] == mm["bytes"]


def test_bare_headline_command_plans_eight_ranks():
    """`python bench.py --gpus 8 --plan-only`, typed bare as the round-end driver types the real command: the parent starts 8
    fresh rank processes (torch.distributed.run, gloo control plane), each derives its share with the library's host-side
    entry points (comm.grid_for / coords_of, bcg_halo_plan, bcg_sbcgrq_plan_bytes) and rank 0 prints the plan.  This is the
    BASELINE headline's launch geometry -- 128^4 on a (2,2,2,1) grid, 64^3 x 128 per GPU, capacity ring 32 with overlapped
    15-slice chunks -- checked without a GPU: eight GPU processes do not fit a one-GPU test box (its process guard allows
    six), so the computing rehearsal of this grid runs its ranks as threads (tests/test_distributed_gpu.py)."""
    import json
    out = subprocess.run([sys.executable, BENCH, "--gpus", "8", "--plan-only"], env=_clean_env(), capture_output=True, text=True,
                         timeout=600)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["plan_only"] and d["n_gpus"] == 8 and d["config"]["process_grid"] == [2, 2, 2, 1]
    assert d["config"]["global_dims"] == [128] * 4 and d["config"]["headline_ladder"] and d["config"]["transport"] == "rccl"
    assert d["capacity_ring_slices"] == 32 and d["ring_overlapped"] and d["ring_chunk_slices"] == 15
    assert d["ring_chunks"] == [15] * 8 + [8] and d["shift_group_depth"] == 2
    # fits the device with room for the runtime and RCCL's two communicators: 288 GiB - 16 GiB
    assert d["device_bytes_planned"] < (288 - 16) * 2 ** 30
    face = 64 * 64 * 128 * 3 * 16 * 16  # sites of a face x bytes per site at m = 16
    assert len(d["ranks"]) == 8
    for r in d["ranks"]:
        c = r["coords"]
        assert r["rank"] == c[0] + 2 * c[1] + 4 * c[2] and c[3] == 0 and r["ghost_sites"] == 6 * 64 * 64 * 128
        msgs = r["messages"]
        assert len(msgs) == 6 and all(mm["bytes"] == face for mm in msgs)
        for mu in range(3):  # a direction split over two ranks: the plus and the minus neighbour are the same peer, twice
            peer = r["rank"] ^ (1 << mu)
            lo, hi = msgs[2 * mu], msgs[2 * mu + 1]
            assert lo["send_to"] == lo["recv_from"] == hi["send_to"] == hi["recv_from"] == peer
            # low face -> the peer's plus ghost, high face -> its minus ghost: crossed offsets inside the direction's range
            assert lo["send_offset"] == hi["recv_offset"] == 2 * mu * face and hi["send_offset"] == lo["recv_offset"] == (2 * mu + 1) * face
    # what rank a sends to b in message k is what b expects from a in ITS message k (posting-order matching)
    for a in d["ranks"]:
        for k, mm in enumerate(a["messages"]):
            b = d["ranks"][mm["send_to"]]
            assert b["messages"][k]["recv_from"] == a["rank"] and b["messages"][k]["bytes"] == mm["bytes"]
    # the shorter ladder rungs the driver also runs
    for n, grid, gd in ((2, [1, 1, 2, 1], [64, 64, 128, 128]), (4, [1, 2, 2, 1], [64, 128, 128, 128])):
        out = subprocess.run([sys.executable, BENCH, "--gpus", str(n), "--plan-only"], env=_clean_env(), capture_output=True,
                             text=True, timeout=600)
        assert out.returncode == 0, out.stderr[-3000:]
        d = json.loads([ln for ln in out.stdout.splitlines() if ln.startswith("{")][0])
        assert d["config"]["process_grid"] == grid and d["config"]["global_dims"] == gd and d["capacity_ring_slices"] == 32
