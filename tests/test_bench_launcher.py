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


def test_roofline_prices_the_stencil_family_and_the_pmc_rate(tmp_path):
    """The two stencil applications of an iteration are two profile classes (hop, hop_shifted_gram) of one kernel template:
    together they are the class with the most time, and `roofline` prices them together (launch-weighted), with every
    member and the dominant single class beside it.  pmc_bytes_per_iteration: counters x launch counts, for the shape the
    counters were taken at only and only when every class has an entry."""
    import json
    import pytest
    bench = _bench_module()
    prof = {
        "hop": {"ms": 4 * 8.3, "count": 4, "bytes": 4 * 35.43e9, "flops": 4 * 0.155e12},
        "hop_shifted_gram": {"ms": 4 * 10.8, "count": 4, "bytes": 4 * 48.32e9, "flops": 4 * 0.258e12},
        "phaseC_multi4": {"ms": 57.1, "count": 1, "bytes": 257.7e9, "flops": 3.09e12},
        "phaseC": {"ms": 33.2, "count": 3, "bytes": 3 * 64.4e9, "flops": 3 * 0.3e12},
        "phaseB": {"ms": 29.8, "count": 4, "bytes": 4 * 38.65e9, "flops": 4 * 0.31e12},
        "stencil_form_k_hop4b": {"ms": 0.0, "count": 8, "bytes": 0.0},
    }
    tj = tmp_path / "traffic.json"
    tj.write_text(json.dumps({"hop": {"bytes_per_launch": 46.1e9}, "hop_shifted_gram": {"bytes_per_launch": 60.7e9},
                              "phaseC_multi4": {"bytes_per_launch": 257.7e9}, "phaseC": {"bytes_per_launch": 64.5e9},
                              "phaseB": {"bytes_per_launch": 38.7e9},
                              "_shape": {"local_dims": [64] * 4, "m": 16, "n_shifts": 4, "capacity": 0}}))
    r = bench.roofline_of(prof, [64] * 4, 16, 4, 0, 1, traffic_path=str(tj))
    assert r["kernel"] == "stencil family: hop + hop_shifted_gram" and r["bound"] == "hbm" and r["launches"] == 8
    assert r["ms_in_timed_region"] == pytest.approx(4 * 19.1) and r["avg_launch_ms"] == pytest.approx(19.1 / 2)
    assert r["achieved"] == pytest.approx((35.43e9 + 48.32e9) / 19.1e-3 / 1e9) and r["frac"] == pytest.approx(r["achieved"] / 8000.0)
    assert r["traffic"] == pytest.approx((46.1e9 + 60.7e9) / 2) and r["algorithmic_bytes_per_launch"] == pytest.approx((35.43e9 + 48.32e9) / 2)
    assert set(r["members"]) == {"hop", "hop_shifted_gram"} and r["members"]["hop"]["frac"] == pytest.approx(35.43e9 / 8.3e-3 / 1e9 / 8000)
    assert r["dominant_single_class"]["kernel"] == "phaseC_multi4" and r["dominant_single_class"]["bound"] == "mfma"
    assert bench.family_of("hop_shifted_gram_ring") == "stencil" and bench.family_of("phaseC") == "phaseC"
    # the PMC rate of the whole iteration: K = 4 iterations in this profile
    pmc = bench.pmc_bytes_per_iteration(prof, 4, [64] * 4, 16, 4, 0, 1, traffic_path=str(tj))
    assert pmc == pytest.approx((4 * 46.1e9 + 4 * 60.7e9 + 257.7e9 + 3 * 64.5e9 + 4 * 38.7e9) / 4)
    assert bench.pmc_bytes_per_iteration(prof, 4, [64, 64, 64, 128], 16, 4, 32, 1, traffic_path=str(tj)) is None  # another shape
    assert bench.pmc_bytes_per_iteration(prof, 4, [64] * 4, 16, 4, 0, 2, traffic_path=str(tj)) is None  # several ranks
    prof["axpby"] = {"ms": 1.0, "count": 1, "bytes": 1e9, "flops": 0.0}  # a class the counters do not cover
    assert bench.pmc_bytes_per_iteration(prof, 4, [64] * 4, 16, 4, 0, 1, traffic_path=str(tj)) is None


def test_memory_ladder_rungs_and_plans():
    """bench.memory_ladder / rung_plan (host arithmetic over bcg_sbcgrq_plan_bytes): the headline shape falls back through
    ring 32 -> 16 -> 8 -> the half-volume form of the same lattice; an explicit shape has one rung unless --step-down; the
    strong-scaling ladder has one."""
    sys.path.insert(0, ROOT)
    from blockcg_amd import _lib
    bench = _bench_module()
    lib = _lib.load()
    rungs = bench.memory_ladder(8, [64, 64, 64, 128], 32, False, 16, True)
    assert [r["label"] for r in rungs] == ["ring 32", "ring 16", "ring 8", "half-volume"]
    assert [r["grid"] for r in rungs] == [[2, 2, 2, 1]] * 3 + [[1, 2, 4, 1]]
    assert rungs[3]["local_dims"] == [128, 64, 32, 128] and rungs[3]["half"] and rungs[3]["capacity"] == 0
    plans = [bench.rung_plan(lib, r, 8, 16, 4) for r in rungs]
    assert all(p["gdims"] == [128] * 4 for p in plans)  # every rung is the SAME lattice
    assert plans[0]["planned"] > plans[1]["planned"] > plans[2]["planned"] > plans[3]["planned"]
    assert 288e9 < plans[0]["planned"] < 290e9 and 203e9 < plans[3]["planned"] < 207e9  # (14 half fields: + the spare P_0 of the deferred X_0)
    assert (plans[0]["chunk"], plans[1]["chunk"], plans[2]["chunk"]) == (15, 7, 3) and plans[3]["depth"] == 4 and plans[0]["depth"] == 2
    for n in (2, 4):
        assert [r["label"] for r in bench.memory_ladder(n, [64, 64, 64, 128], 32, False, 16, True)][-1] == "half-volume"
    assert len(bench.memory_ladder(1, [64] * 4, 0, False, 16, True)) == 1
    assert len(bench.memory_ladder(4, [32, 8, 8, 16], 16, False, 16, False)) == 1
    assert [r["label"] for r in bench.memory_ladder(4, [32, 8, 8, 16], 16, False, 16, False, step_down=True)] == ["ring 16", "ring 8", "half-volume"]
    assert len(bench.memory_ladder(8, [64, 32, 32, 32], 0, False, 16, True, strong=True)) == 1
    assert len(bench.memory_ladder(8, [128, 64, 32, 128], 0, True, 16, True)) == 1  # --half asked for: nothing below it


@pytest.mark.parametrize("short_gb,rung,ring", [(2, "ring 16", 16), (5, "half-volume", 0)])
def test_bare_headline_command_steps_down_the_memory_ladder(short_gb, rung, ring):
    """`python bench.py --gpus 8 --plan-only` told (BCG_DEBUG_FIELD_BUDGET, the library's stand-in for a full device) that
    the GPUs have `short_gb` GB less than ring 32 needs: the eight ranks agree on the next rung that fits, print it, and
    plan THAT launch (grid, messages, bytes)."""
    import json
    sys.path.insert(0, ROOT)
    from blockcg_amd import _lib
    bench = _bench_module()
    p32 = bench.rung_plan(_lib.load(), bench.memory_ladder(8, [64, 64, 64, 128], 32, False, 16, True)[0], 8, 16, 4)["planned"]
    free = p32 + bench.RUNTIME_RESERVE - short_gb * 10 ** 9
    out = subprocess.run([sys.executable, BENCH, "--gpus", "8", "--plan-only"], env=_clean_env(BCG_DEBUG_FIELD_BUDGET=str(free)),
                         capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-3000:]
    d = json.loads([ln for ln in out.stdout.splitlines() if ln.startswith("{")][-1])
    assert d["config"]["memory_ladder_rung"] == rung and d["capacity_ring_slices"] == ring and d["config"]["global_dims"] == [128] * 4
    notes = d["config"]["memory_ladder"]
    assert notes[0]["rung"] == "ring 32" and "exceed" in notes[0]["verdict"] and notes[-1]["rung"] == rung and notes[-1]["verdict"] == "chosen"
    assert all(n["free_bytes_min_over_ranks"] == free for n in notes)
    assert d["device_bytes_planned"] + bench.RUNTIME_RESERVE <= free
    assert d["config"]["process_grid"] == ([1, 2, 4, 1] if ring == 0 else [2, 2, 2, 1]) and d["config"]["half_volume_solves"] == (ring == 0)
    assert len(d["ranks"]) == 8 and all(len(r["messages"]) == (4 if ring == 0 else 6) for r in d["ranks"])


def test_memory_ladder_exhausted_is_an_error_not_a_hang():
    """80 GB short is still the half-volume form (205 GB planned); 100 GB short fits nothing: every rank exits non-zero with the reason."""
    import json
    sys.path.insert(0, ROOT)
    from blockcg_amd import _lib
    bench = _bench_module()
    p32 = bench.rung_plan(_lib.load(), bench.memory_ladder(2, [64, 64, 64, 128], 32, False, 16, True)[0], 2, 16, 4)["planned"]
    for short_gb, ok in ((80, True), (100, False)):
        free = p32 + bench.RUNTIME_RESERVE - short_gb * 10 ** 9
        out = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--plan-only"], env=_clean_env(BCG_DEBUG_FIELD_BUDGET=str(free)),
                             capture_output=True, text=True, timeout=600)
        if ok:
            assert out.returncode == 0, out.stderr[-3000:]
            d = json.loads([ln for ln in out.stdout.splitlines() if ln.startswith("{")][-1])
            assert d["config"]["memory_ladder_rung"] == "half-volume" and d["config"]["process_grid"] == [1, 1, 2, 1]
        else:
            assert out.returncode != 0 and "no rung of the memory ladder fits" in out.stderr


def test_strong_scaling_ladder_plans():
    """`--ladder strong`: V = 64^4 in total at every N, whole tmp, `scaling` = strong."""
    import json
    for n, grid in ((1, [1, 1, 1, 1]), (4, [1, 1, 2, 2])):
        out = subprocess.run([sys.executable, BENCH, "--gpus", str(n), "--ladder", "strong", "--plan-only"], env=_clean_env(),
                             capture_output=True, text=True, timeout=600)
        assert out.returncode == 0, out.stderr[-3000:]
        d = json.loads([ln for ln in out.stdout.splitlines() if ln.startswith("{")][-1])
        assert d["config"]["global_dims"] == [64] * 4 and d["config"]["process_grid"] == grid and d["scaling"] == "strong"
        assert d["capacity_ring_slices"] == 0 and not d["config"]["headline_ladder"] and len(d["config"]["memory_ladder"]) == 1


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
    assert 200e9 < d["device_bytes_planned"] < 210e9  # 14 half fields of 12.9 GB + full links + faces: 284 GB in capacity mode
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
