#!/usr/bin/env python3
"""bench.py -- SBCGrQ iteration throughput on MI355X (BASELINE.json metric).

A "step" is one SBCGrQ iteration (one pass of the loop body, inc/block_solvers.hpp:132-182) in
fixed-work mode (eps = eps_shifts = 0: every shift stays active, work per iteration is constant).
Workload at N = 1: BASELINE.json configs[2], V = 64^4, m = 16 right-hand sides, 4 shifts, fp64 --
the largest configuration that fits one GPU (128^4 needs 2.4 TB).

N > 1 is the ladder that ends at the BASELINE headline, configs[3] (V = 128^4, m = 16, 4 shifts on 8 GPUs): every
GPU holds 64 x 64 x 64 x 128 sites (capacity mode with a ring of 32 x3-slices: 15-slice chunks whose face exchanges
overlap the stencil work on the neighbouring chunks; 287 GB planned of the 288 GiB = 309 GB) and the process
grid grows over x2, x1, x0 with x3 undivided:
    N = 2: 64 x 64 x 128 x 128 (grid 1,1,2,1)   N = 4: 64 x 128 x 128 x 128 (1,2,2,1)   N = 8: 128^4 (2,2,2,1)
`python bench.py --gpus N` starts its N ranks itself (fresh child processes under torch.distributed.run, before this
process has imported torch or touched a GPU); under an existing launcher (WORLD_SIZE set) it is one of the ranks.
One process per GPU; halo faces (grouped ncclSend/ncclRecv) and the m x m all-reduce go over RCCL/xGMI through
libblockcg_rccl.so (native code, include/blockcg_rccl.h); torch.distributed (gloo) is only the launcher's control plane
(rendezvous of RCCL's unique id, the barrier around the timed region and the max over ranks).
BCG_BACKEND=torch-nccl selects the torch.distributed RCCL transport of blockcg_amd/comm.py instead, BCG_BACKEND=gloo the
host-staged one (several ranks sharing one GPU, rehearsals only).  --local-dims / --capacity override the shape.

The headline ladder steps itself down on memory (`memory_ladder`): 288.8 GB of the 309.2 GB per GPU are planned at ring 32,
and what RCCL's communicators take is only known once they are up.  So after the communicators exist (and have contacted
every peer once: RCCL allocates per-peer buffers at first use) every rank reads its free device memory, the ranks take the
MINIMUM over the gloo control plane and the first rung whose plan (bcg_sbcgrq_plan_bytes) fits is run:
    ring 32 -> ring 16 -> ring 8 -> the declared half-volume form (--half: 192 GB per GPU, same lattice, grid (1,2,4,1))
A rung whose allocation still fails on any rank (bcg_sbcgrq_begin returns on EVERY rank then: the ranks agree before their
first collective) is abandoned by all ranks together and the next one tried, in the same processes.  The line records the
rung that ran (`config.memory_ladder`, `capacity_ring_slices`, `config.workload`) and stderr every step.
`--ladder strong` is the other curve SURVEY.md section 8d asks for: V = 64^4 in total on 1/2/4/8 GPUs (no ring needed).

Prints ONE JSON line on rank 0.  `value` = lattice-site iterations per second summed over all
GPUs (iterations/s x global volume) with all inputs resident in HBM; iterations/s and the achieved
algorithmic HBM GB/s are reported next to it.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured copy)
# fp64 peak, matrix pipe and VALU alike: 256 CUs x 4 SIMDs x 2.4 GHz x 2048 flop per v_mfma_f64_16x16x4_f64 / 64 cycles per
# issue (tools/microbench/mfma_f64_rate.hip: 65.0 cycles, 77.4 TFLOP/s measured; profiles/r03_group_depth.txt)
FP64_PEAK_TFLOPS = 78.6
SHIFTS = [0.0, 1e-6, 1e-4, 1e-2, 1e-5, 1e-3, 1e-1, 1.0]  # SURVEY.md section 8d; first S, sorted
MASS = 1e-3


def descendants(pid):
    """PIDs of every live descendant of `pid` (from /proc), children before parents' siblings -- the exact processes this
    script started, for the timeout path."""
    kids = {}
    for d in os.listdir("/proc"):
        if not d.isdigit():
            continue
        try:
            with open(f"/proc/{d}/stat") as f:
                st = f.read()
            ppid = int(st[st.rindex(")") + 2:].split()[1])
        except (OSError, ValueError):
            continue
        kids.setdefault(ppid, []).append(int(d))
    out, todo = [], [pid]
    while todo:
        p = todo.pop()
        for k in kids.get(p, []):
            out.append(k)
            todo.append(k)
    return out


def cpu_baseline(m, S, shifts, mass, budget_iters=32):
    """Time the reference's own SBCGrQ (oracle/_ref, built from /root/reference in the build container)
    or, if that is absent, this repository's CPU restatement, on one host core, at V = 16^4: a bounded sample of the
    same workload (32 fixed iterations, about 13 s)."""
    import numpy as np
    import oracle
    dims = [16, 16, 16, 16]
    V = 16 ** 4
    O = oracle.Oracle()
    cores = 1
    if oracle.ref_available() and m in oracle.REF_SUPPORTED_M:
        R = oracle.Reference(four_d=True)
        U = O.fill_gauge(dims, 1)
        B = O.fill_field(m, V, 2)
        R.make_dirac_nd(dims, mass, U)
        t_setup = R.sbcgrq(B, shifts, 0.0, 0.0, max_iterations=0)["seconds"]
        t_total = R.sbcgrq(B, shifts, 0.0, 0.0, max_iterations=budget_iters)["seconds"]
        kind = "reference"
        what = "unmodified reference SBCGrQ + field primitives (oracle/_ref/libref4d.so) over the 4-D operator"
    else:
        dt, t_setup = O.bench_sbcgrq(m, dims, mass, shifts, budget_iters, seed=1)
        t_total = dt + t_setup
        kind = "port"
        what = "CPU restatement oracle/oracle.hpp"
    dt = max(t_total - t_setup, 1e-9)
    out = {"value": V * budget_iters / dt, "unit": "site-iter/s", "cores": cores, "kind": kind,
           "sample": f"{what}; V=16^4, m={m}, {S} shifts, {budget_iters} fixed iterations (eps=0), "
                     f"{dt:.2f} s on 1 of {os.cpu_count()} host cores",
           "iterations_per_sec_at_sample": budget_iters / dt}
    # Labelled extra: the CPU restatement with its site loops split over host threads.  The reference
    # itself is single-threaded, so `value` above stays the one-core figure.
    try:
        nthr = max(1, min(16, len(os.sched_getaffinity(0))))
        O.set_threads(nthr)
        dt_mt, _ = O.bench_sbcgrq(m, dims, mass, shifts, budget_iters, seed=1)
        O.set_threads(1)
        out["all_cores"] = {"value": V * budget_iters / max(dt_mt, 1e-9), "unit": "site-iter/s", "cores": nthr, "kind": "port",
                            "sample": f"oracle/oracle.hpp with OpenMP over sites, {nthr} threads, same sample, {dt_mt:.2f} s"}
    except Exception as e:
        out["all_cores"] = {"error": repr(e)}
    return out


def launch_ranks(args, argv):
    """`python bench.py --gpus N` with no launcher around it (WORLD_SIZE unset): start the N ranks as fresh child processes
    -- `python -m torch.distributed.run --nproc-per-node N bench.py <same flags>` -- relay rank 0's JSON line and the exit
    code.  The parent never imports torch or blockcg_amd and makes no GPU call (a process that has initialised the GPU must
    not be replaced, and N ranks must not race to build): it compiles the libraries once with make, picks a free port,
    and kills the whole process group on a timeout (BCG_BENCH_TIMEOUT seconds, default 1500)."""
    import signal
    import socket
    import subprocess
    csrc = os.path.join(ROOT, "blockcg_amd", "csrc")
    targets = [["-j4"]]  # libblockcg_hip.so + libblockcg_rccl.so
    if os.path.basename(os.environ.get("BCG_RCCL_LIB", "")) == "libblockcg_rccl_mock.so":
        targets.append(["mock"])  # rehearsal transport (several ranks on one GPU)
    for t in targets:
        r = subprocess.run(["make", "-C", csrc, "-s"] + t, stdout=sys.stderr)
        if r.returncode != 0:
            sys.exit(f"bench.py: building the libraries failed (make rc {r.returncode})")
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + list(argv)
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"),
               OMP_NUM_THREADS=os.environ.get("OMP_NUM_THREADS", "1"))
    timeout = float(os.environ.get("BCG_BENCH_TIMEOUT", "1500"))
    proc = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, text=True, start_new_session=True)
    try:
        out, _ = proc.communicate(timeout=timeout)
    except subprocess.TimeoutExpired:
        ranks = descendants(proc.pid)       # the launcher's workers (it starts each in a session of its own)
        proc.send_signal(signal.SIGTERM)    # torch.distributed.run forwards it to its workers
        try:
            out, _ = proc.communicate(timeout=15)
        except subprocess.TimeoutExpired:
            out = None
        for pid in ranks + [proc.pid]:      # whatever is left: exactly the processes started here
            try:
                os.kill(pid, signal.SIGKILL)
            except ProcessLookupError:
                pass
        if out is None:
            out, _ = proc.communicate()
        sys.stdout.write(out or "")
        sys.exit(f"bench.py: the {args.gpus} ranks did not finish within {timeout:.0f} s; launcher and ranks killed")
    lines = [ln for ln in (out or "").splitlines() if ln.startswith("{")]
    other = [ln for ln in (out or "").splitlines() if not ln.startswith("{")]
    if other:
        sys.stderr.write("\n".join(other) + "\n")
    if proc.returncode != 0:
        sys.exit(proc.returncode if proc.returncode > 0 else 1)
    if not lines:
        sys.exit("bench.py: the ranks exited cleanly but rank 0 printed no JSON line")
    print(lines[-1], flush=True)


def apply_config(args, world):
    """--config 4: the declared stand-in shape of BASELINE configs[4] (m = 32, 8 shifts; 128^4 at that width needs 8 TB): the
    largest volume that fits 288 GB per GPU, 64^3 x 32 sites (19 fields of 12.9 GB + links = 250 GB), i.e. 128^3 x 32 on 8 GPUs.
    --config 1: BASELINE configs[1] (32^4, m = 8, 1 shift, one GPU).
    --config 2 / 3 are the defaults at N = 1 / N > 1 and change nothing."""
    if args.config == 1:
        if world != 1:
            sys.exit("--config 1 is a one-GPU configuration (32^4, m = 8, 1 shift)")
        args.m, args.shifts = 8, 1
        if args.local_dims is None:
            args.local_dims = [32, 32, 32, 32]
    if args.config == 4:
        args.m, args.shifts = 32, 8
        if args.local_dims is None:
            args.local_dims = [64, 64, 64, 32]
        if args.capacity is None:
            args.capacity = 32 if world > 1 else 0  # ring = L3: no memory saved, but the chunked exchanges overlap
    return args


def resolve_shape(world, local_dims, capacity, half=False):
    """(sites per GPU, capacity ring, on-the-headline-ladder?) for a world size: the defaults documented at the top."""
    ladder = world > 1 and local_dims is None
    if half:
        # --half: the same ladder as two half-volume solves per GPU share (no ring: a half solve's 13 half fields + links are
        # 188 GB at this share), the process grid growing over x2 and x1 (comm.grid_for_half) with x0 -- the direction the half
        # fields are compact in -- and x3 -- swept in chunks whose face exchanges overlap the stencil -- undivided:
        # N = 2 128x64x64x128 (1,1,2,1), N = 4 128x128x64x128 (1,2,2,1), N = 8 128^4 (1,2,4,1)
        if capacity:
            sys.exit("--half: half-volume fields have no ring form (capacity mode)")
        if local_dims is None:
            local_dims = [128, 64, 32, 128] if world > 1 else [64, 64, 64, 64]
        return list(local_dims), 0, ladder
    if local_dims is None:
        local_dims = [64, 64, 64, 128] if world > 1 else [64, 64, 64, 64]
    if capacity is None:
        # ring 32: with the per-chunk exchanges overlapped the ring holds two chunks (C = 15 slices: the bundle form of the
        # stencil needs windows of >= 10); measured on one GPU at this share, serial form: ring 16 139.5 ms, ring 8 142.5 ms
        capacity = 32 if ladder else 0
    return list(local_dims), capacity, ladder



# what a rung's plan must leave free: kernel code objects, event pools, the runtime's own growth (1.5 GiB; BCG_BENCH_RESERVE
# bytes for rehearsals at test sizes, where the plans themselves are megabytes)
RUNTIME_RESERVE = int(os.environ.get("BCG_BENCH_RESERVE", 3 * 2 ** 29))
ASSUMED_FREE = (288 - 16) * 2 ** 30  # --plan-only without a GPU: a 288 GiB device less 16 GiB for the runtime and RCCL


def memory_ladder(world, local_dims, capacity, half, m, default_shape, strong=False, step_down=None):
    """The configurations a run may fall back through, in order: [{"label", "local_dims", "capacity", "half", "grid"}].
    One rung unless stepping down applies: the default headline shape on several GPUs, or --step-down on an explicit one.
    Ring rungs halve the ring while it still divides L3 and stays >= 8 slices (a ring of 8 is the smallest measured); the
    last rung is the declared half-volume form of the SAME global lattice on the half ladder's grid, where extents allow."""
    from blockcg_amd.comm import grid_for, grid_for_half
    ndim = len(local_dims)

    def rung(label, ld, cap, hf):
        if world == 1:
            grid = [1] * ndim
        elif hf:
            grid = grid_for_half(world, ndim)
        else:
            grid = grid_for(world, ndim, keep_last=cap > 0)
        if os.environ.get("BCG_BENCH_GRID") and world > 1 and hf == bool(half):  # rehearsal aid: an explicit process grid
            grid = [int(x) for x in os.environ["BCG_BENCH_GRID"].split(",")]
            assert len(grid) == ndim and int(__import__("math").prod(grid)) == world
        return {"label": label, "local_dims": list(ld), "capacity": int(cap), "half": bool(hf), "grid": grid}

    first = rung("half-volume" if half else (f"ring {capacity}" if capacity else "whole tmp"), local_dims, capacity, half)
    rungs = [first]
    stepping = (default_shape and world > 1 and not strong) if step_down is None else step_down
    if not stepping or half:
        return rungs
    L3 = local_dims[-1]
    r = capacity // 2
    while capacity and r >= 8 and L3 % r == 0:
        rungs.append(rung(f"ring {r}", local_dims, r, False))
        r //= 2
    if ndim == 4 and world > 1:
        gdims = [l * g for l, g in zip(local_dims, first["grid"])]
        hgrid = grid_for_half(world, ndim)
        if all(gd % g == 0 and (gd // g) % 2 == 0 for gd, g in zip(gdims, hgrid)):
            rungs.append(rung("half-volume", [gd // g for gd, g in zip(gdims, hgrid)], 0, True))
    return rungs


def rung_plan(lib, rung, world, m, S):
    """Host arithmetic only (bcg_sbcgrq_plan_bytes): what one rank of this rung allocates, its ring chunking and the depth
    its shift updates are grouped over (pair_shifts_depth in the library)."""
    import ctypes
    ld, cap, half, grid = rung["local_dims"], rung["capacity"], rung["half"], rung["grid"]
    ndim = len(ld)
    gdims = [l * g for l, g in zip(ld, grid)]
    iv = lambda v: (ctypes.c_int * 4)(*(list(v) + [1] * (4 - len(v))))  # noqa: E731
    overlapped = bool(cap >= 4 and world > 1)  # every transport offers the split callbacks
    chunk = ((cap - 2) // 2 if overlapped else cap - 2) if cap else 0
    if m not in (8, 16, 32):
        depth = 1
    elif S < 2:  # a single system groups for the deferred X_0 update's sake alone (m = 8, 16 outside capacity mode)
        depth = 4 if (m in (8, 16) and not cap) else 1
    else:
        depth = 2 if (cap or m == 32) else 4
    planned = ctypes.c_size_t()
    rc = lib.bcg_sbcgrq_plan_bytes(ndim, iv(gdims), iv(grid), m, S, 1, cap, 1 if overlapped else 0, depth, ctypes.byref(planned))
    if rc != 0:
        raise ValueError(f"bcg_sbcgrq_plan_bytes rejected the rung {rung['label']} (rc {rc})")
    if half:  # one half-volume solve at a time: every work field (X_s, P_s, Q, T, tmp, the further residual buffers) holds
        # half the sites; links, face buffers and scratch stay (bcg_sbcgrq_device_bytes_half needs a context: same arithmetic)
        V_local = 1
        for l in ld:
            V_local *= l
        spare = 1 if depth >= 2 and m in (8, 16) else 0  # the spare P_0 of the deferred X_0 update (outside capacity mode)
        planned.value -= (2 * S + 3 + max(0, depth - 2) + spare) * (V_local // 2) * 3 * m * 16
    L3 = ld[-1]
    return {"gdims": gdims, "planned": planned.value, "overlapped": overlapped, "chunk": chunk, "depth": depth,
            "chunks": ([min(chunk, L3 - lo) for lo in range(0, L3, chunk)] if chunk else [])}


def ladder_note(rung, plan, free, verdict):
    return {"rung": rung["label"], "local_dims": rung["local_dims"], "process_grid": rung["grid"], "planned_bytes": plan["planned"],
            "free_bytes_min_over_ranks": free, "runtime_reserve_bytes": RUNTIME_RESERVE, "verdict": verdict}


STENCIL_CLASSES = ("hop", "hop_shifted", "hop_shifted_gram", "hop_boundary", "hop_ring", "hop_shifted_ring", "hop_shifted_gram_ring",
                   "hop_half", "hop_half_shifted", "hop_half_shifted_gram")


def family_of(name):
    """Profile classes that are instantiations of one kernel template are one FAMILY for the roofline: the two stencil
    applications of an iteration (A1 `hop*`, A2 `hop_shifted*`: k_hop4b / k_hop4c) are booked as two classes, and booked
    apart neither is the class with the most time while together they are."""
    return "stencil" if name in STENCIL_CLASSES else name


def roofline_of(prof, local_dims, m, S, capacity, world, traffic_path=None):
    """The `roofline` object of the bench line: the kernel FAMILY with the most time in the timed region (HIP events taken
    by the library on its stream; family_of: the stencil's two applications count together), priced against the roof its
    arithmetic intensity puts it under -- HBM bytes or fp64 flops.
    `bytes` / `flops` = the ALGORITHMIC bytes and fp64 flops of the launches timed under a name (accumulated by the library
    per launch: per-site figures of DESIGN.md section 4 x the sites the launch processes), so split launches (phase C in
    several launches, capacity-mode windows) are priced right.  `members` prices every class of the family on its own and
    `dominant_single_class` is the one class with the most time (what `roofline` was before the families).
    Returns None when nothing was profiled."""
    kernels = {k: v for k, v in prof.items() if not k.startswith("stencil_form_") and v.get("bytes", 0) > 0}
    if not kernels:
        return None
    ridge = FP64_PEAK_TFLOPS * 1e12 / (HBM_PEAK_GBPS * 1e9)
    # HBM bytes per launch from the PMC counters are measured in separate rocprofv3 passes
    # (tools/profile_round.sh -> profiles/hbm_traffic.json); they are quoted only for the shape they were taken at
    tj = None
    traffic_source = None
    tpath = traffic_path or os.path.join(ROOT, "profiles", "hbm_traffic.json")
    if os.path.exists(tpath):
        cand = json.load(open(tpath))
        shape = cand.get("_shape", {})
        if (shape.get("local_dims") == list(local_dims) and shape.get("m") == m and shape.get("n_shifts") == S
                and shape.get("capacity", 0) == capacity and world == 1):
            tj = cand
            traffic_source = f"profiles/hbm_traffic.json ({shape.get('measured', 'separate rocprofv3 --pmc passes')})"

    def priced(label, names):
        """One roofline entry for the launches of the classes `names` together."""
        ms = sum(kernels[k]["ms"] for k in names)
        count = sum(kernels[k]["count"] for k in names)
        nbytes = sum(kernels[k]["bytes"] for k in names)
        flops = sum(kernels[k].get("flops", 0.0) for k in names)
        avg_ms = ms / count
        ach = nbytes / (ms * 1e-3) / 1e9
        tf = flops / (ms * 1e-3) / 1e12
        # which roof binds: the arithmetic intensity (algorithmic flops / algorithmic bytes) against the ridge point of the
        # two peaks -- a property of the work, not of which of the two fractions came out larger
        intensity = flops / nbytes if nbytes > 0 else 0.0
        r = {"bound": "hbm", "kernel": label, "achieved": ach, "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBPS}
        if intensity > ridge:
            r = {"bound": "mfma", "kernel": label, "achieved": tf, "peak": FP64_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": tf / FP64_PEAK_TFLOPS}
        traffic = None
        if tj is not None and all(k in tj for k in names):  # PMC bytes per launch of the family: launch-weighted mean
            traffic = sum(tj[k]["bytes_per_launch"] * kernels[k]["count"] for k in names) / count
        r.update({"intensity_flop_per_byte": intensity, "ridge_flop_per_byte": ridge, "traffic": traffic,
                  "traffic_source": traffic_source if traffic is not None else None, "avg_launch_ms": avg_ms, "launches": count,
                  "algorithmic_bytes_per_launch": nbytes / count, "flops_per_launch": flops / count,
                  "hbm_frac": ach / HBM_PEAK_GBPS, "fp64_frac": tf / FP64_PEAK_TFLOPS, "ms_in_timed_region": ms})
        return r

    families = {}
    for k in kernels:
        families.setdefault(family_of(k), []).append(k)
    fam = max(families, key=lambda f: sum(kernels[k]["ms"] for k in families[f]))
    names = sorted(families[fam])
    label = names[0] if len(names) == 1 else f"{fam} family: " + " + ".join(names)
    roof = priced(label, names)
    if len(names) > 1:
        roof["members"] = {k: {kk: vv for kk, vv in priced(k, [k]).items()
                               if kk in ("bound", "achieved", "unit", "frac", "avg_launch_ms", "launches", "traffic",
                                         "algorithmic_bytes_per_launch", "hbm_frac", "fp64_frac")} for k in names}
    single = max(kernels, key=lambda k: kernels[k]["ms"])
    roof["dominant_single_class"] = {kk: vv for kk, vv in priced(single, [single]).items()
                                     if kk in ("kernel", "bound", "achieved", "peak", "unit", "frac", "avg_launch_ms", "launches",
                                               "traffic", "hbm_frac", "fp64_frac", "ms_in_timed_region")}
    stencil_ratio = None
    if tj is not None:
        sr = {kn: tj[kn]["bytes_per_launch"] / (kernels[kn]["bytes"] / kernels[kn]["count"])
              for kn in ("hop", "hop_shifted_gram") if kn in tj and kn in kernels}
        stencil_ratio = sr or None
    per_kernel = {k: round(v["bytes"] / (v["ms"] * 1e-3) / 1e9 / HBM_PEAK_GBPS, 4) for k, v in kernels.items() if v["ms"] > 0}
    # second roofline: the fp64 flops of the same launches against the chip's fp64 rate (matrix pipe = VALU rate on
    # gfx950: one v_mfma_f64_16x16x4_f64 per 64 cycles and SIMD, measured by tools/microbench/mfma_f64_rate.hip).
    # It binds the grouped phase C (k_phaseC_multi), whose fields are read once per several iterations.
    per_kernel_flop = {k: round(v.get("flops", 0.0) / (v["ms"] * 1e-3) / 1e12 / FP64_PEAK_TFLOPS, 4)
                       for k, v in kernels.items() if v["ms"] > 0}
    roof.update({"stencil_traffic_ratio": stencil_ratio, "per_kernel_frac": per_kernel, "per_kernel_fp64_frac": per_kernel_flop})
    # Algorithmic bytes and flops are lower bounds of the work, so a fraction above 1 means either an accounting error or
    # a shape whose fields fit the 256 MB Infinity Cache (small --local-dims): flagged, never fatal -- the line must print
    # (tests/test_bench_launcher.py keeps the strict check on the accounting itself).
    over = sorted(k for k, f in list(per_kernel.items()) + list(per_kernel_flop.items()) if f > 1.0)
    roof["accounting_suspect"] = bool(over)
    if over:
        sys.stderr.write(f"bench.py: roofline fraction above 1 for {over} (cache-resident shape, or an accounting error)\n")
    return roof


def pmc_bytes_per_iteration(prof, K, local_dims, m, S, capacity, world, traffic_path=None):
    """HBM bytes per iteration from the PMC counters (profiles/hbm_traffic.json: FETCH_SIZE + WRITE_SIZE per launch, taken
    in separate rocprofv3 passes) x this run's launch counts -- only for the shape the counters were taken at and only when
    every profiled kernel class has an entry; None otherwise."""
    tpath = traffic_path or os.path.join(ROOT, "profiles", "hbm_traffic.json")
    if world != 1 or not os.path.exists(tpath):
        return None
    tj = json.load(open(tpath))
    shape = tj.get("_shape", {})
    if not (shape.get("local_dims") == list(local_dims) and shape.get("m") == m and shape.get("n_shifts") == S
            and shape.get("capacity", 0) == capacity):
        return None
    kernels = {k: v for k, v in prof.items() if not k.startswith("stencil_form_") and v.get("bytes", 0) > 0}
    if not kernels or not all(k in tj for k in kernels):
        return None
    return sum(tj[k]["bytes_per_launch"] * v["count"] for k, v in kernels.items()) / K


def plan_only(args, world, default_shape):
    """`--plan-only`: the launch geometry of this run without touching a GPU.  Each rank calls the same host-side entry
    points the real run uses -- memory_ladder / rung_plan (comm.grid_for, bcg_sbcgrq_plan_bytes), coords_of, bcg_halo_plan
    (the face messages: peers, offsets, sizes) -- picks the rung the real run would pick for the free memory it is told
    (BCG_DEBUG_FIELD_BUDGET bytes, the library's stand-in for a full device; else a 288 GiB device less 16 GiB) and rank 0
    gathers everything over the gloo control plane into one JSON line."""
    import ctypes
    from blockcg_amd import _lib
    from blockcg_amd.comm import coords_of
    lib = _lib.load()
    rank = int(os.environ.get("RANK", "0"))
    ndim = len(args.local_dims)
    dist = None
    if world > 1:
        import torch.distributed as dist
        dist.init_process_group(backend="gloo")
    m, S = args.m, args.shifts
    rungs = memory_ladder(world, args.local_dims, args.capacity, args.half, m, default_shape, args.ladder == "strong",
                          True if args.step_down else None)
    free = int(os.environ.get("BCG_DEBUG_FIELD_BUDGET", "0")) or ASSUMED_FREE
    notes, chosen, plan = [], None, None
    for rung in rungs:
        pl = rung_plan(lib, rung, world, m, S)
        fits = pl["planned"] + RUNTIME_RESERVE <= free
        notes.append(ladder_note(rung, pl, free, "chosen" if fits else "plan + reserve exceed the free memory"))
        if fits:
            chosen, plan = rung, pl
            break
    if chosen is None:
        if rank == 0:
            print(json.dumps({"plan_only": True, "n_gpus": world, "error": "no rung of the memory ladder fits", "memory_ladder": notes}), flush=True)
        sys.exit("bench.py: no rung of the memory ladder fits the free device memory")
    grid, gdims = chosen["grid"], plan["gdims"]
    coords = coords_of(rank, grid)
    iv = lambda v: (ctypes.c_int * 4)(*(list(v) + [1] * (4 - len(v))))  # noqa: E731
    ps, pr = (ctypes.c_int * 8)(), (ctypes.c_int * 8)()
    os_, or_, nb = (ctypes.c_size_t * 8)(), (ctypes.c_size_t * 8)(), (ctypes.c_size_t * 8)()
    ghost = ctypes.c_int64()
    cv = (ctypes.c_int * 4)(*(list(coords) + [0] * (4 - len(coords))))
    # (half-volume fields: every face holds half its sites -- the same messages at half the bytes per site)
    n = lib.bcg_halo_plan(ndim, iv(gdims), iv(grid), cv, 3 * m * 16 // (2 if chosen["half"] else 1), ps, pr, os_, or_, nb, ctypes.byref(ghost))
    assert n >= 0
    transport = os.environ.get("BCG_BACKEND", "rccl")
    mine = {"rank": rank, "coords": coords, "ghost_sites": ghost.value,
            "messages": [{"send_to": ps[k], "recv_from": pr[k], "send_offset": os_[k], "recv_offset": or_[k], "bytes": nb[k]} for k in range(n)],
            "device_bytes_planned": plan["planned"]}
    ranks = [mine]
    if dist is not None:
        ranks = [None] * world if rank == 0 else None
        dist.gather_object(mine, ranks, dst=0)
    if rank == 0:
        ld = chosen["local_dims"]
        out = {"plan_only": True, "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
               "scaling": "strong" if args.ladder == "strong" else "weak",
               "config": {"workload": f"SBCGrQ V={'x'.join(map(str, gdims))} ({'x'.join(map(str, ld))} per GPU), m={m}, {S} shifts",
                          "global_dims": gdims, "process_grid": grid, "m": m, "shifts": sorted(SHIFTS[:S]), "transport": transport,
                          "headline_ladder": bool(default_shape and world > 1 and args.ladder != "strong"),
                          "half_volume_solves": bool(chosen["half"]), "memory_ladder": notes, "memory_ladder_rung": chosen["label"]},
               "capacity_ring_slices": chosen["capacity"], "ring_overlapped": plan["overlapped"], "ring_chunk_slices": plan["chunk"],
               "ring_chunks": plan["chunks"],
               "shift_group_depth": plan["depth"], "device_bytes_planned": max(r["device_bytes_planned"] for r in ranks), "ranks": ranks}
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--half", action="store_true",
                    help="declared option (SURVEY 8f-4): the solve as two half-volume solves, one per site parity "
                         "(dirac_op::D couples opposite parities only); a step = one iteration of each, timed as K "
                         "iterations of the even solve + K of the odd one (one solve's fields alive at a time)")
    ap.add_argument("--steps", type=int, default=12)  # a multiple of the depth the shift updates are grouped over (4)
    ap.add_argument("--warmup", type=int, default=4)
    ap.add_argument("--local-dims", type=int, nargs="+", default=None,
                    help="sites per GPU (default: 64 64 64 64 on one GPU, 64 64 64 128 on several)")
    ap.add_argument("--m", type=int, default=16)
    ap.add_argument("--shifts", type=int, default=4)
    ap.add_argument("--config", type=int, default=None, choices=[1, 2, 3, 4],
                    help="a BASELINE.json configuration by number: 1 = 32^4, m=8, 1 shift on one GPU; 2 = 64^4, m=16, 4 shifts on one GPU (the default at N=1); "
                         "3 = the 128^4 ladder (the default at N>1); 4 = the wide-block stress, m=32 and 8 shifts, on the "
                         "largest volume that fits: 64^3 x 32 sites per GPU (19 fields of 12.9 GB + links = 250 GB; the "
                         "BASELINE's 128^4 would need 8 TB), i.e. 128^3 x 32 on 8 GPUs")
    ap.add_argument("--ladder", choices=["weak", "strong"], default="weak",
                    help="weak (default): 64^3 x 128 sites per GPU, ending at 128^4 on 8 GPUs (the BASELINE headline); strong: "
                         "V = 64^4 in total on 1/2/4/8 GPUs (SURVEY.md section 8d), no ring, `scaling` = strong")
    ap.add_argument("--step-down", action="store_true",
                    help="step down the memory ladder (smaller rings, then the half-volume form) on an explicit shape too; the "
                         "default multi-GPU shape always does")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--generic", action="store_true", help="force the generic VALU kernels")
    ap.add_argument("--plan-only", action="store_true",
                    help="no GPU work: every rank derives its share of the run (process grid, coordinates, face messages, "
                         "capacity ring and chunking, planned device bytes, the rung of the memory ladder) with the library's "
                         "host-side functions and rank 0 prints them as one JSON line -- a rehearsal of the launch geometry "
                         "on a machine without the GPUs")
    ap.add_argument("--capacity", type=int, default=None, metavar="R",
                    help="capacity mode: keep the operator's intermediate field as a ring of R x3 slices "
                         "(bcg_capacity_mode); the process grid then leaves x3 undivided "
                         "(default: 0 on one GPU, 32 on several with the default shape)")
    args = ap.parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        return launch_ranks(args, sys.argv[1:])
    world = int(os.environ.get("WORLD_SIZE", "1"))
    apply_config(args, world)
    if os.environ.get("BCG_BENCH_TEST_HANG"):  # tests/test_bench_launcher.py: a rank that never finishes
        time.sleep(3600)
    strong = args.ladder == "strong"
    if strong:
        if args.local_dims is not None or args.capacity or args.half or args.config == 4:
            sys.exit("--ladder strong is the fixed lattice 64^4 over the GPUs: no shape flags")
        from blockcg_amd.comm import grid_for
        sgrid = grid_for(world, 4) if world > 1 else [1, 1, 1, 1]
        total = [64, 64, 64, 64]
        if os.environ.get("BCG_BENCH_STRONG_DIMS"):  # rehearsal aid: a smaller fixed lattice (several ranks on one test GPU)
            total = [int(x) for x in os.environ["BCG_BENCH_STRONG_DIMS"].split(",")]
        args.local_dims, args.capacity = [t // g for t, g in zip(total, sgrid)], 0
    default_shape = (args.local_dims is None and args.capacity is None) or strong
    args.local_dims, args.capacity, headline_ladder = resolve_shape(world, args.local_dims, args.capacity, args.half)
    headline_ladder = headline_ladder and not strong

    if args.plan_only:
        return plan_only(args, world, default_shape)

    import torch
    import blockcg_amd as bc

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    device = int(os.environ.get("BCG_DEVICE", local_rank))  # BCG_DEVICE: rehearse several ranks on one GPU (with gloo)
    if args.gpus > 1 and world != args.gpus:
        sys.exit(f"--gpus {args.gpus} needs torch.distributed.run with --nproc-per-node {args.gpus} (WORLD_SIZE={world})")
    dist = None
    ndim = len(args.local_dims)
    transport = "none"
    if world > 1:
        import torch.distributed as dist
        from blockcg_amd.comm import TorchDistComm, coords_of
        torch.cuda.set_device(device)
        transport = os.environ.get("BCG_BACKEND", "rccl")
        if transport not in ("rccl", "torch-nccl", "gloo"):
            sys.exit(f"BCG_BACKEND={transport}: expected rccl (native, default), torch-nccl or gloo")
        if transport == "torch-nccl":  # torch's "nccl" backend is RCCL on ROCm
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", device))
        else:
            dist.init_process_group(backend="gloo")  # control plane only when transport == "rccl"

    def over_ranks(value, op):
        """min / max of a number over the ranks on the control plane (the value itself on one rank)."""
        if dist is None:
            return value
        t = torch.tensor([float(value)], dtype=torch.float64, device="cuda" if dist.get_backend() == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MIN if op == "min" else dist.ReduceOp.MAX)
        return float(t[0].item())

    def say(msg):
        if rank == 0:
            sys.stderr.write(f"bench.py: {msg}\n")
            sys.stderr.flush()

    # tests: BCG_DEBUG_FIELD_BUDGET on ONE rank only (the allocation failure every other rank must learn of)
    only = os.environ.get("BCG_DEBUG_BUDGET_ONLY_RANK")
    if only is not None and int(only) != rank:
        os.environ.pop("BCG_DEBUG_FIELD_BUDGET", None)
    m, S = args.m, args.shifts
    shifts = sorted(SHIFTS[:S])
    from blockcg_amd import _lib as bc_lib
    lib = bc_lib.load()
    rungs = memory_ladder(world, args.local_dims, args.capacity, args.half, m, default_shape, strong,
                          True if args.step_down else None)

    class Run:  # what one rung holds alive
        ctx = comm = D = st = X = B = None
        grid = gdims = None

    run = Run()

    def teardown():
        """Everything of the current rung's context, in dependency order (fields before the transport before the context)."""
        if run.st is not None:
            run.st.end()
        run.st = run.X = run.B = run.D = None
        if run.comm is not None and transport == "rccl":
            run.comm.close()
        run.comm = None
        if run.ctx is not None:
            run.ctx.close()
        run.ctx = None

    def setup(rung, plan):
        """Context + transport of a rung (rank-collective); a rung on the same grid and lattice keeps them."""
        if run.ctx is not None and run.grid == rung["grid"] and run.gdims == plan["gdims"]:
            return
        teardown()
        run.grid, run.gdims = list(rung["grid"]), list(plan["gdims"])
        if world == 1:
            run.ctx = bc.Context(run.gdims, device=device)
        else:
            coords = coords_of(rank, run.grid)
            if transport == "rccl":
                from blockcg_amd import rccl
                uid = [rccl.get_unique_id() if rank == 0 else None]
                dist.broadcast_object_list(uid, src=0)
                run.ctx = bc.Context(run.gdims, device=device, grid=run.grid, coords=coords)
                run.comm = rccl.RcclComm(run.ctx, uid[0], rank, world)
                run.comm.warm_up()  # RCCL's per-peer buffers exist before the free memory is read
            else:
                run.comm = TorchDistComm(device)
                run.ctx = bc.Context(run.gdims, device=device, grid=run.grid, coords=coords, stream=run.comm.stream_ptr)
                run.comm.attach(run.ctx)
        if args.generic:
            run.ctx.force_generic(True)

    def start_solve(par):
        """Fields and solver state of one solve (bcg_sbcgrq_begin).  An allocation that fails on ANY rank fails the begin on
        EVERY rank (the library's ranks agree before their first collective), and the control plane confirms it: returns
        False on all ranks together, with nothing of the solve left allocated."""
        ok = 1.0
        try:
            if run.D is None:
                run.D = bc.dirac_op(run.ctx, MASS, seed=1)
            run.B = bc.block_fermion_field(run.ctx, m, parity=par).setRandom(seed=2)
            run.X = [bc.block_fermion_field(run.ctx, m, parity=par) for _ in shifts]
        except bc.BlockCGError as e:
            sys.stderr.write(f"bench.py: rank {rank}: {e}\n")
            ok = 0.0
        ok = over_ranks(ok, "min")
        if ok:
            try:
                run.st = bc.SBCGrQState(run.X, run.B, run.D, shifts, 0.0, 0.0, consume_B=True)
            except bc.BlockCGError as e:
                sys.stderr.write(f"bench.py: rank {rank}: {e}\n")
                ok = 0.0
            ok = over_ranks(ok, "min")
        if not ok:
            if run.st is not None:
                run.st.end()
            run.st = run.X = run.B = None
            run.ctx.synchronize()
        return bool(ok)

    # ---- the memory ladder: the first rung whose plan fits the free memory AND whose allocations succeed on every rank ----
    notes, chosen, plan = [], None, None
    budget = int(os.environ.get("BCG_DEBUG_FIELD_BUDGET", "0"))  # the library's stand-in for a full device caps "free" too
    trust_free = os.environ.get("BCG_BENCH_TRUST_FREE", "") == "1"  # tests: let bcg_sbcgrq_begin be the one to find out
    for rung in rungs:
        pl = rung_plan(lib, rung, world, m, S)
        setup(rung, pl)
        run.ctx.capacity_mode(rung["capacity"])
        free = torch.cuda.mem_get_info(device)[0]
        if run.D is not None:  # the links of an earlier rung on this context are part of the plan and already allocated
            free += run.ctx.V * ndim * 144
        if budget and not trust_free:
            free = min(free, budget)
        free = int(over_ranks(free, "min"))
        if len(rungs) > 1 and pl["planned"] + RUNTIME_RESERVE > free:
            notes.append(ladder_note(rung, pl, free, "skipped: plan + reserve exceed the free device memory"))
            say(f"memory ladder: {rung['label']} plans {pl['planned'] / 1e9:.1f} GB + {RUNTIME_RESERVE / 1e9:.1f} GB reserve, "
                f"{free / 1e9:.1f} GB free (minimum over ranks): stepping down")
            continue
        if start_solve(0 if rung["half"] else None):
            notes.append(ladder_note(rung, pl, free, "ran"))
            chosen, plan = rung, pl
            break
        notes.append(ladder_note(rung, pl, free, "abandoned: an allocation failed on at least one rank"))
        say(f"memory ladder: {rung['label']} could not be allocated on every rank: stepping down")
    if chosen is None:
        teardown()
        if rank == 0:
            sys.stderr.write("bench.py: no rung of the memory ladder could be allocated: " + json.dumps(notes) + "\n")
        sys.exit(1)
    if len(rungs) > 1:
        say(f"memory ladder: running {chosen['label']} ({plan['planned'] / 1e9:.1f} GB planned per GPU)")
    ctx, grid, gdims = run.ctx, run.grid, run.gdims
    local_dims, capacity, half = chosen["local_dims"], chosen["capacity"], chosen["half"]

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()
        ctx.synchronize()

    # one solve on all sites, or (half-volume form) one per site parity, each timed over exactly K iterations between barriers
    dt, bytes_in_use, residual = 0.0, 0, None
    for par in ((0, 1) if half else (None,)):
        if run.st is None and not start_solve(par):  # (the odd solve of the half-volume form: same plan as the even one)
            teardown()
            sys.exit(f"bench.py: rank {rank}: the solve of parity {par} could not be allocated")
        st = run.st
        st.iterate(args.warmup)
        mem_free, mem_total = torch.cuda.mem_get_info(device)  # with every field of the solve alive
        bytes_in_use = max(bytes_in_use, mem_total - mem_free)
        ctx.profiling(True)
        if not par:
            ctx.profile_reset()
        barrier()
        t0 = time.perf_counter()
        st.iterate(args.steps)          # EXACTLY K iterations; iterate() synchronizes the stream before returning
        barrier()
        dt += time.perf_counter() - t0
        ctx.profiling(False)
        residual = st.residual if residual is None else max(residual, st.residual)
        st.end()  # (the even solve's fields go before the odd solve's are made)
        run.st = run.X = run.B = st = None
    prof = ctx.profile()
    dt = over_ranks(dt, "max")
    bytes_in_use = int(over_ranks(bytes_in_use, "max"))
    comm = run.comm
    n_comms = comm.communicators if transport == "rccl" else None  # 2: the split exchange has a communicator of its own
    if comm is not None and comm.error is not None:
        raise comm.error
    planned_now = ctx.sbcgrq_device_bytes_half(m, S, consume_B=True) if half else ctx.sbcgrq_device_bytes(m, S, consume_B=True)
    bytes_iter = ctx.bytes_per_iteration(m, S)

    if rank == 0:
        Vg = 1
        for d in gdims:
            Vg *= d
        K = args.steps
        its = K / dt
        bytes_iter_total = bytes_iter * world
        hbm_gbps = bytes_iter_total * its / 1e9
        moved_per_gpu = sum(v.get("bytes", 0.0) for k, v in prof.items() if not k.startswith("stencil_form_")) / K
        # (half-volume form: the PMC traffic file was taken on full-volume launches -- passing a world of 2 keeps it from being quoted)
        roof = roofline_of(prof, list(local_dims), m, S, capacity, 2 if half else world)
        pmc_iter = pmc_bytes_per_iteration(prof, K, list(local_dims), m, S, capacity, 2 if half else world)
        out = {
            "metric": "SBCGrQ lattice-site iterations/sec (iterations/sec x global volume), fp64",
            "value": Vg * its, "unit": "site-iter/s", "n_gpus": world, "steps": K, "warmup": args.warmup,
            "ms_per_step": dt / K * 1e3, "higher_is_better": True, "scaling": "strong" if strong else "weak", "vs_baseline": None,
            "dtype": "f64", "data": "synthetic",
            "config": {"workload": f"SBCGrQ V={'x'.join(map(str, gdims))} ({'x'.join(map(str, local_dims))} per GPU), "
                                   f"m={m}, {S} shifts, mass={MASS}, fixed-work (eps=0)"
                                   + ("; BASELINE configs[4] (m=32, 8 shifts) on the largest volume that fits 288 GB per GPU"
                                      if args.config == 4 else "")
                                   + ("; strong-scaling ladder: V=64^4 in total on 1/2/4/8 GPUs" if strong else "")
                                   + ("; ladder to the 128^4 headline: N=1 64^4, N=2 64x64x128x128, N=4 64x128^3, N=8 128^4 "
                                      f"(64^3x128 per GPU, capacity ring {capacity})" if headline_ladder and not half else "")
                                   + ("; two half-volume solves (one per site parity), a step = one iteration of each"
                                      + ("" if args.half else " -- the memory ladder's last rung, NOT asked for by a flag")
                                      if half else ""),
                       "global_dims": gdims, "process_grid": grid, "m": m, "shifts": shifts, "transport": transport,
                       "rccl_communicators": n_comms, "half_volume_solves": bool(half),
                       "memory_ladder_rung": chosen["label"], "memory_ladder": notes},
            "iterations_per_sec": its,
            # bytes_alg of SURVEY.md section 8d, V[(14 + 4(S-1)) 48 m + 2 g] x iterations/s: the metric's definition, a RATE IN
            # THE METRIC'S UNITS and not traffic -- since round 3 the iteration moves a third fewer bytes than that sequence of
            # passes (Q rho^-1 is not stored, the shifted systems are updated four iterations at a time).
            "hbm_GBps_algorithmic": hbm_gbps, "bytes_alg_per_iteration": bytes_iter_total,
            # what was ACHIEVED, per GPU: the bytes this fusion has to move (sum of the per-launch algorithmic bytes of every
            # kernel launched in the timed region) / time, and -- where profiles/hbm_traffic.json holds this shape -- the bytes
            # the PMC counters saw the same launches move
            "hbm_GBps_moved_per_gpu": moved_per_gpu / (dt / K) / 1e9,
            "hbm_GBps_pmc_per_gpu": (pmc_iter / (dt / K) / 1e9) if pmc_iter else None,
            "bytes_moved_minimum_per_iteration": moved_per_gpu * world,
            "bytes_pmc_per_iteration": pmc_iter,
            "hbm_frac_of_bytes_actually_moved": moved_per_gpu / (dt / K) / 1e9 / HBM_PEAK_GBPS,
            "hbm_frac_pmc": (pmc_iter / (dt / K) / 1e9 / HBM_PEAK_GBPS) if pmc_iter else None,
            "residual_after_timed_steps": residual,
            "kernel_ms": {k: round(v["ms"], 3) for k, v in prof.items() if not k.startswith("stencil_form_")},
            "stencil_kernel_launches": {k[len("stencil_form_"):]: v["count"] for k, v in prof.items() if k.startswith("stencil_form_")},
            "capacity_ring_slices": capacity,
            "device_bytes_planned": planned_now,
            "device_bytes_in_use": bytes_in_use, "device_bytes_total": mem_total,  # max over ranks
            "comm_ms_per_iteration": {k: round(v["ms"] / K, 4) for k, v in prof.items()
                                      if k.startswith("halo_exchange") or k in ("allreduce", "pack_faces")} if world > 1 else None,
            "roofline": roof,
        }
        if world == 1 and not args.no_cpu_baseline:
            try:
                out["cpu_baseline"] = cpu_baseline(m, S, shifts, MASS)
                out["gpu_over_cpu_site_iter_rate"] = out["value"] / out["cpu_baseline"]["value"]
            except Exception as e:  # the bench line must still print
                out["cpu_baseline"] = {"error": repr(e)}
        print(json.dumps(out), flush=True)
    teardown()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
