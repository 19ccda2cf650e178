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
    --config 2 / 3 are the defaults at N = 1 / N > 1 and change nothing."""
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



def roofline_of(prof, local_dims, m, S, capacity, world, traffic_path=None):
    """The `roofline` object of the bench line: the kernel class with the most time in the timed region (HIP events taken by
    the library on its stream), priced against the roofline it is closest to -- HBM bytes or fp64 flops.
    `bytes` / `flops` = the ALGORITHMIC bytes and fp64 flops of the launches timed under a name (accumulated by the library
    per launch: per-site figures of DESIGN.md section 4 x the sites the launch processes), so split launches (phase C in
    several launches, capacity-mode windows) are priced right.  Returns None when nothing was profiled."""
    kernels = {k: v for k, v in prof.items() if not k.startswith("stencil_form_") and v.get("bytes", 0) > 0}
    if not kernels:
        return None
    name = max(kernels, key=lambda k: kernels[k]["ms"])
    e = kernels[name]
    avg_ms = e["ms"] / e["count"]
    kb = e["bytes"] / e["count"]
    # HBM bytes per launch from the PMC counters are measured in separate rocprofv3 passes
    # (tools/profile_round.sh -> profiles/hbm_traffic.json); they are quoted only for the shape they were taken at
    traffic, traffic_source, stencil_ratio = None, None, None
    tpath = traffic_path or os.path.join(ROOT, "profiles", "hbm_traffic.json")
    if os.path.exists(tpath):
        tj = json.load(open(tpath))
        shape = tj.get("_shape", {})
        if (shape.get("local_dims") == list(local_dims) and shape.get("m") == m and shape.get("n_shifts") == S
                and shape.get("capacity", 0) == capacity and world == 1):
            traffic = tj.get(name, {}).get("bytes_per_launch")
            traffic_source = f"profiles/hbm_traffic.json ({shape.get('measured', 'separate rocprofv3 --pmc passes')})"
            sr = {}
            for kn in ("hop", "hop_shifted_gram"):
                if kn in tj and kn in kernels:
                    sr[kn] = tj[kn]["bytes_per_launch"] / (kernels[kn]["bytes"] / kernels[kn]["count"])
            stencil_ratio = sr or None
    ach = kb / (avg_ms * 1e-3) / 1e9
    per_kernel = {k: round(v["bytes"] / (v["ms"] * 1e-3) / 1e9 / HBM_PEAK_GBPS, 4) for k, v in kernels.items() if v["ms"] > 0}
    # second roofline: the fp64 flops of the same launches against the chip's fp64 rate (matrix pipe = VALU rate on
    # gfx950: one v_mfma_f64_16x16x4_f64 per 64 cycles and SIMD, measured by tools/microbench/mfma_f64_rate.hip).
    # It binds the grouped phase C (k_phaseC_multi), whose fields are read once per several iterations.
    per_kernel_flop = {k: round(v.get("flops", 0.0) / (v["ms"] * 1e-3) / 1e12 / FP64_PEAK_TFLOPS, 4)
                       for k, v in kernels.items() if v["ms"] > 0}
    hbm_frac = ach / HBM_PEAK_GBPS
    tf = e.get("flops", 0.0) / e["count"] / (avg_ms * 1e-3) / 1e12
    mfma_frac = tf / FP64_PEAK_TFLOPS
    # which roof binds the launch: its arithmetic intensity (algorithmic flops / algorithmic bytes) against the ridge point
    # of the two peaks -- a property of the work, not of which of the two fractions came out larger
    intensity = (e.get("flops", 0.0) / e["bytes"]) if e["bytes"] > 0 else 0.0
    ridge = FP64_PEAK_TFLOPS * 1e12 / (HBM_PEAK_GBPS * 1e9)
    roof = {"bound": "hbm", "kernel": name, "achieved": ach, "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": hbm_frac}
    if intensity > ridge:
        roof = {"bound": "mfma", "kernel": name, "achieved": tf, "peak": FP64_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": mfma_frac}
    roof.update({"intensity_flop_per_byte": intensity, "ridge_flop_per_byte": ridge})
    roof.update({"traffic": traffic, "traffic_source": traffic_source,
                 "avg_launch_ms": avg_ms, "launches": e["count"], "algorithmic_bytes_per_launch": kb,
                 "flops_per_launch": e.get("flops", 0.0) / e["count"], "hbm_frac": hbm_frac, "fp64_frac": mfma_frac,
                 "stencil_traffic_ratio": stencil_ratio, "per_kernel_frac": per_kernel,
                 "per_kernel_fp64_frac": per_kernel_flop})
    # Algorithmic bytes and flops are lower bounds of the work, so a fraction above 1 means either an accounting error or
    # a shape whose fields fit the 256 MB Infinity Cache (small --local-dims): flagged, never fatal -- the line must print
    # (tests/test_bench_launcher.py keeps the strict check on the accounting itself).
    over = sorted(k for k, f in list(per_kernel.items()) + list(per_kernel_flop.items()) if f > 1.0)
    roof["accounting_suspect"] = bool(over)
    if over:
        sys.stderr.write(f"bench.py: roofline fraction above 1 for {over} (cache-resident shape, or an accounting error)\n")
    return roof

def plan_only(args, world, default_shape):
    """`--plan-only`: the launch geometry of this run without touching a GPU.  Each rank calls the same host-side entry
    points the real run uses -- comm.grid_for / coords_of, bcg_halo_plan (the face messages: peers, offsets, sizes),
    bcg_sbcgrq_plan_bytes -- and rank 0 gathers them over the gloo control plane into one JSON line."""
    import ctypes
    from blockcg_amd import _lib
    from blockcg_amd.comm import coords_of, grid_for, grid_for_half
    lib = _lib.load()
    rank = int(os.environ.get("RANK", "0"))
    ndim = len(args.local_dims)
    dist = None
    if world > 1:
        import torch.distributed as dist
        dist.init_process_group(backend="gloo")
    grid = (grid_for_half(world, ndim) if args.half else grid_for(world, ndim, keep_last=args.capacity > 0)) if world > 1 else [1] * ndim
    if os.environ.get("BCG_BENCH_GRID"):
        grid = [int(x) for x in os.environ["BCG_BENCH_GRID"].split(",")]
    coords = coords_of(rank, grid)
    gdims = [l * g for l, g in zip(args.local_dims, grid)]
    m, S = args.m, args.shifts
    iv = lambda v: (ctypes.c_int * 4)(*(list(v) + [1] * (4 - len(v))))  # noqa: E731
    ps, pr = (ctypes.c_int * 8)(), (ctypes.c_int * 8)()
    os_, or_, nb = (ctypes.c_size_t * 8)(), (ctypes.c_size_t * 8)(), (ctypes.c_size_t * 8)()
    ghost = ctypes.c_int64()
    cv = (ctypes.c_int * 4)(*(list(coords) + [0] * (4 - len(coords))))
    # (half-volume fields: every face holds half its sites -- the same messages at half the bytes per site)
    n = lib.bcg_halo_plan(ndim, iv(gdims), iv(grid), cv, 3 * m * 16 // (2 if args.half else 1), ps, pr, os_, or_, nb, ctypes.byref(ghost))
    assert n >= 0
    transport = os.environ.get("BCG_BACKEND", "rccl")
    overlapped = bool(args.capacity >= 4 and world > 1)  # every transport offers the split callbacks
    chunk = ((args.capacity - 2) // 2 if overlapped else args.capacity - 2) if args.capacity else 0
    depth = 1 if S < 2 else (2 if (args.capacity or m == 32) else 4)  # pair_shifts_depth in blockcg_capi.hip
    planned = ctypes.c_size_t()
    rc = lib.bcg_sbcgrq_plan_bytes(ndim, iv(gdims), iv(grid), m, S, 1, args.capacity, 1 if overlapped else 0, depth, ctypes.byref(planned))
    assert rc == 0, rc
    if args.half:  # one half-volume solve at a time: every work field (X_s, P_s, Q, T, tmp, the further residual buffers) holds
        # half the sites; links, face buffers and scratch stay (bcg_sbcgrq_device_bytes_half needs a context: same arithmetic)
        V_local = 1
        for l in args.local_dims:
            V_local *= l
        planned.value -= (2 * S + 3 + max(0, depth - 2)) * (V_local // 2) * 3 * m * 16
    mine = {"rank": rank, "coords": coords, "ghost_sites": ghost.value,
            "messages": [{"send_to": ps[k], "recv_from": pr[k], "send_offset": os_[k], "recv_offset": or_[k], "bytes": nb[k]} for k in range(n)],
            "device_bytes_planned": planned.value}
    ranks = [mine]
    if dist is not None:
        ranks = [None] * world if rank == 0 else None
        dist.gather_object(mine, ranks, dst=0)
    if rank == 0:
        L3 = args.local_dims[-1]
        out = {"plan_only": True, "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
               "config": {"workload": f"SBCGrQ V={'x'.join(map(str, gdims))} ({'x'.join(map(str, args.local_dims))} per GPU), m={m}, {S} shifts",
                          "global_dims": gdims, "process_grid": grid, "m": m, "shifts": sorted(SHIFTS[:S]), "transport": transport,
                          "headline_ladder": bool(default_shape and world > 1), "half_volume_solves": bool(args.half)},
               "capacity_ring_slices": args.capacity, "ring_overlapped": overlapped, "ring_chunk_slices": chunk,
               "ring_chunks": ([min(chunk, L3 - lo) for lo in range(0, L3, chunk)] if chunk else []),
               "shift_group_depth": depth, "device_bytes_planned": max(r["device_bytes_planned"] for r in ranks), "ranks": ranks}
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
    ap.add_argument("--config", type=int, default=None, choices=[2, 3, 4],
                    help="a BASELINE.json configuration by number: 2 = 64^4, m=16, 4 shifts on one GPU (the default at N=1); "
                         "3 = the 128^4 ladder (the default at N>1); 4 = the wide-block stress, m=32 and 8 shifts, on the "
                         "largest volume that fits: 64^3 x 32 sites per GPU (19 fields of 12.9 GB + links = 250 GB; the "
                         "BASELINE's 128^4 would need 8 TB), i.e. 128^3 x 32 on 8 GPUs")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--generic", action="store_true", help="force the generic VALU kernels")
    ap.add_argument("--plan-only", action="store_true",
                    help="no GPU work: every rank derives its share of the run (process grid, coordinates, face messages, "
                         "capacity ring and chunking, planned device bytes) with the library's host-side functions and rank 0 "
                         "prints them as one JSON line -- a rehearsal of the launch geometry on a machine without the GPUs")
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
    default_shape = args.local_dims is None and args.capacity is None
    args.local_dims, args.capacity, headline_ladder = resolve_shape(world, args.local_dims, args.capacity, args.half)

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
    comm = None
    ndim = len(args.local_dims)
    transport = "none"
    if world > 1:
        import torch.distributed as dist
        from blockcg_amd.comm import TorchDistComm, coords_of, grid_for, grid_for_half
        torch.cuda.set_device(device)
        transport = os.environ.get("BCG_BACKEND", "rccl")
        if transport not in ("rccl", "torch-nccl", "gloo"):
            sys.exit(f"BCG_BACKEND={transport}: expected rccl (native, default), torch-nccl or gloo")
        if transport == "torch-nccl":  # torch's "nccl" backend is RCCL on ROCm
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", device))
        else:
            dist.init_process_group(backend="gloo")  # control plane only when transport == "rccl"
        grid = grid_for_half(world, ndim) if args.half else grid_for(world, ndim, keep_last=args.capacity > 0)
        if os.environ.get("BCG_BENCH_GRID"):  # rehearsal aid: an explicit process grid, e.g. "1,1,2,1"
            grid = [int(x) for x in os.environ["BCG_BENCH_GRID"].split(",")]
            assert len(grid) == ndim and int(__import__("math").prod(grid)) == world
        coords = coords_of(rank, grid)
        gdims = [l * g for l, g in zip(args.local_dims, grid)]
        if transport == "rccl":
            from blockcg_amd import rccl
            uid = [rccl.get_unique_id() if rank == 0 else None]
            dist.broadcast_object_list(uid, src=0)
            ctx = bc.Context(gdims, device=device, grid=grid, coords=coords)
            comm = rccl.RcclComm(ctx, uid[0], rank, world)
        else:
            comm = TorchDistComm(device)
            ctx = bc.Context(gdims, device=device, grid=grid, coords=coords, stream=comm.stream_ptr)
            comm.attach(ctx)
    else:
        grid = [1] * ndim
        gdims = list(args.local_dims)
        ctx = bc.Context(gdims, device=device)
    if args.generic:
        ctx.force_generic(True)
    if args.capacity:
        ctx.capacity_mode(args.capacity)

    m, S = args.m, args.shifts
    shifts = sorted(SHIFTS[:S])
    D = bc.dirac_op(ctx, MASS, seed=1)

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()
        ctx.synchronize()

    # one solve on all sites, or (--half) one per site parity, each timed over exactly K iterations between barriers
    dt, bytes_in_use, residual = 0.0, 0, None
    for par in ((0, 1) if args.half else (None,)):
        B = bc.block_fermion_field(ctx, m, parity=par).setRandom(seed=2)
        X = [bc.block_fermion_field(ctx, m, parity=par) for _ in shifts]
        st = bc.SBCGrQState(X, B, D, shifts, 0.0, 0.0, consume_B=True)
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
        if par == 0:  # the even solve's fields go before the odd solve's are made
            st.end()
            del st, X, B
    prof = ctx.profile()
    if dist is not None:
        t = torch.tensor([dt, float(bytes_in_use)], dtype=torch.float64, device="cuda" if dist.get_backend() == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt, bytes_in_use = float(t[0].item()), int(t[1].item())
    n_comms = comm.communicators if transport == "rccl" else None  # 2: the split exchange has a communicator of its own
    st.end()
    if comm is not None and comm.error is not None:
        raise comm.error
    if transport == "rccl":
        comm.close()

    if rank == 0:
        Vg = 1
        for d in gdims:
            Vg *= d
        K = args.steps
        its = K / dt
        bytes_iter_total = ctx.bytes_per_iteration(m, S) * world
        hbm_gbps = bytes_iter_total * its / 1e9
        # (--half: the PMC traffic file was taken on full-volume launches -- passing a world of 2 keeps it from being quoted)
        roof = roofline_of(prof, list(args.local_dims), m, S, args.capacity, 2 if args.half else world)
        out = {
            "metric": "SBCGrQ lattice-site iterations/sec (iterations/sec x global volume), fp64",
            "value": Vg * its, "unit": "site-iter/s", "n_gpus": world, "steps": K, "warmup": args.warmup,
            "ms_per_step": dt / K * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f64", "data": "synthetic",
            "config": {"workload": f"SBCGrQ V={'x'.join(map(str, gdims))} ({'x'.join(map(str, args.local_dims))} per GPU), "
                                   f"m={m}, {S} shifts, mass={MASS}, fixed-work (eps=0)"
                                   + ("; BASELINE configs[4] (m=32, 8 shifts) on the largest volume that fits 288 GB per GPU"
                                      if args.config == 4 else "")
                                   + ("; ladder to the 128^4 headline: N=1 64^4, N=2 64x64x128x128, N=4 64x128^3, N=8 128^4 "
                                      "(64^3x128 per GPU, capacity ring 32)" if default_shape and not args.half else "")
                                   + ("; --half: two half-volume solves (one per site parity), a step = one iteration of each"
                                      if args.half else ""),
                       "global_dims": gdims, "process_grid": grid, "m": m, "shifts": shifts, "transport": transport,
                       "rccl_communicators": n_comms},
            "iterations_per_sec": its,
            # bytes_alg of SURVEY.md section 8d, V[(14 + 4(S-1)) 48 m + 2 g]: the metric's definition.  The iteration itself
            # moves one field pass less since round 3 (Q rho^-1 is not stored): see bytes_moved_minimum
            "hbm_GBps_algorithmic": hbm_gbps, "hbm_GBps_per_gpu": hbm_gbps / world,
            "bytes_alg_per_iteration": bytes_iter_total,
            "bytes_moved_minimum_per_iteration": sum(v.get("bytes", 0.0) for k, v in prof.items()
                                                     if not k.startswith("stencil_form_")) / K * world,
            # on the SURVEY's bytes_alg (the reference's sequence of passes): a rate in the metric's units, NOT the share of the
            # HBM peak in use -- the iteration moves fewer bytes than that (next key)
            "hbm_roofline_frac_whole_iteration": hbm_gbps / world / HBM_PEAK_GBPS,
            "hbm_frac_of_bytes_actually_moved": sum(v.get("bytes", 0.0) for k, v in prof.items()
                                                    if not k.startswith("stencil_form_")) / dt / 1e9 / HBM_PEAK_GBPS,
            "residual_after_timed_steps": residual,
            "kernel_ms": {k: round(v["ms"], 3) for k, v in prof.items() if not k.startswith("stencil_form_")},
            "stencil_kernel_launches": {k[len("stencil_form_"):]: v["count"] for k, v in prof.items() if k.startswith("stencil_form_")},
            "capacity_ring_slices": args.capacity,
            "device_bytes_planned": (ctx.sbcgrq_device_bytes_half(m, S, consume_B=True) if args.half
                                     else ctx.sbcgrq_device_bytes(m, S, consume_B=True)),
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
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
