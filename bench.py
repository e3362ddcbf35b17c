#!/usr/bin/env python3
"""bench.py -- the north-star line: GLS interpolate() on a synthetic 10 M-cell hexahedron mesh.

    python bench.py --gpus 1 --steps K --warmup W                      (one GPU)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A "step" is one pass of the hot path over the rank's shard: the GLS weight kernel over every node
(inputs resident in HBM, output written in CSR position on the device) and, for N > 1, the one
all-gather over RCCL that reassembles the (count, column, value) triplets on every rank.  For N > 1 the line
also decomposes itself: `compute_only_Mnodes_s` (the kernels with no exchange), `exchange_ms` (the all-gather of one
step's values timed alone) and `apply_Mnodes_s` -- the sharded W . u (each rank applies its own row block, one
all-gather of 8 B per node), which is what the reference's callers do with the matrix.

Workload (BASELINE.json: "GLS, 10M-cell hex mesh"): 216^3 = 10,077,696 hexahedra per GPU,
nodes jittered U(-0.15h, 0.15h), K = the ALH tensor at the centroids, all-Dirichlet boundary flags.
With N GPUs the box grows to 216 x 216 x 216 N cells (weak scaling; at N = 8 the same 80.6 M cells as SURVEY's
432^3 in a different aspect): rank r owns a contiguous block of node planes, replicates one cell layer on each interior
side, and no rank ever holds the whole mesh.  `--edge-global G` instead fixes the WHOLE mesh at G^3 cells and splits its
node planes over the ranks (strong scaling; BASELINE config [4] as SURVEY wrote it: --gpus 8 --edge-global 432).

Prints ONE JSON line on rank 0.  Besides the contract fields it carries
  roofline      of the dominant kernel, duration measured with HIP events on the launch stream.  GLS: bound "fp64":
                `achieved` = ALGORITHMIC flops of the multifrontal formulation (tools/count_algorithmic_flops.py:
                15 869 per cube node) x the nodes the kernel processes / time against the FP64 vector peak; the flops the
                kernel EXECUTES (ISA count) ride beside it as `executed_*`, the HBM figure -- algorithmic bytes of
                SURVEY 8d / time against 8 TB/s -- as `hbm` / `hbm_frac`; IDW, LS: bound "hbm".  `traffic` from
                profiles/traffic.json when a PMC run of exactly these kernel sources has been recorded, else null
  cpu_baseline  our C restatement of the reference's method kernel (kind "port"; the reference itself never
                travels to the GPU box) timed on this box's host cores on a bounded sample, with the container-measured
                t_port / t_reference calibration beside it
  fp64          GLS is FP64-ALU-bound, not HBM-bound (SURVEY finding 2): executed and reference-equivalent
                FLOP/s next to the 78.6 TFLOP/s vector peak.
  baseline_configs  one row per single-GPU entry of BASELINE.json's `configs`.
"""
import argparse
import hashlib
import json
import os
import sys
import time

os.environ.setdefault("OPENBLAS_NUM_THREADS", "1")   # before anything loads numpy / scipy (OpenBLAS reads it at load)

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0       # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured copy)
FP64_PEAK_TFLOPS = 78.6     # vector FP64
REF_FLOPS_PER_NODE_HEX = 74.8e3   # dgels 44 x 25 x 8 per interior hexahedron node
# FP64 operations the multifrontal kernel EXECUTES per interior hexahedron node (fma = 2, mul / add = 1, the 4 lanes
# of a node summed), counted on the kernel's ISA by tools/count_fp64.py (profiles/r03/hex8w2_isa_mix.txt; round 2's
# one-wave kernel: 20.5e3, profiles/r02/hex8mf_isa_mix.txt)
EXEC_FLOPS_PER_NODE_HEX = 16.7e3
# ALGORITHMIC FP64 flops per node of the multifrontal formulation -- useful arithmetic only, no role masks, no redundant
# panel work (tools/count_algorithmic_flops.py, profiles/r03/algorithmic_flops.txt) -- by node kind: (fronts, dense cells)
ALG_FLOPS = {"cube": 15869.0}


PLAN_KERNEL_NAMES = {   # kernel of the launch plan -> its name in a rocprofv3 trace
    "block1": "nin_gls_block_kernel<1, *>", "block2": "nin_gls_block_kernel<2, *>", "block4": "nin_gls_block_kernel<4, *>",
    "block8": "nin_gls_block_kernel<8, *>", "scratch": "nin_gls_team_kernel", "hex8": "nin_gls_hex8w2_kernel",
    "mfw_large": "nin_gls_mfw_kernel<12, 12, true, false, true>", "mfw_small": "nin_gls_mfw_kernel<6, 6, true, false, false>",
    "mfw_general": "nin_gls_mfw_kernel<12, 15, true, true, false>", "small4": "nin_gls_small_kernel<4>",
    "small8": "nin_gls_small_kernel<8>", "small12": "nin_gls_small_kernel<12>", "quad4": "nin_gls_quad4_kernel", "mfx_6x10": "nin_gls_mfx_kernel<6, 10, false>", "mfx_7x11": "nin_gls_mfx_kernel<7, 11, false>",
    "mfx_8x13": "nin_gls_mfx_kernel<8, 13, false>", "mfx_9x15": "nin_gls_mfx_kernel<9, 15, false>", "mfx_10x16": "nin_gls_mfx_kernel<10, 16, false>",
    "mfx_boundary": "nin_gls_mfx_kernel<7, 11, true>", "mfg_tiles": "nin_gls_mfg_kernel", "mfx_4x7": "nin_gls_mfx_kernel<4, 7, false>", "mfx_7x12": "nin_gls_mfx_kernel<7, 12, false>"}


def gls_kernel_rows(grid, launch, time_launches, reps=3):
    """Every GLS kernel of the launch plan priced and timed on its own: nodes in its list, nodes it
    computes (Dirichlet boundary nodes get the zero row), ALGORITHMIC flops of the formulation it runs on them
    (nin_gls_plan_flops: per node from its own descriptor), kernel ms (NIN_GLS_ONLY=<k>: only that kernel is launched; HIP
    events) and the fraction of the FP64 vector peak.  Returns (rows, total algorithmic flops, total reference-equivalent flops,
    computed nodes whose kernel has no price)."""
    plan = grid.gls_plan()
    flops = grid.gls_plan_flops()
    rows, alg_t, ref_t, unpriced = {}, 0.0, 0.0, 0
    for k, name in enumerate(grid.PLAN_KERNELS):
        if not plan[name]:
            continue
        alg, ref, comp = flops[name]
        os.environ["NIN_GLS_ONLY"] = str(k)
        try:
            launch()
            ms = time_launches(reps)
        finally:
            del os.environ["NIN_GLS_ONLY"]
        alg_t += alg
        ref_t += ref
        if comp and not alg:
            unpriced += comp
        rows[name] = {"nodes": plan[name], "computed": comp, "ms": round(ms, 4), "algorithmic_gflop": round(alg / 1e9, 4),
                      "kflop_per_computed_node": round(alg / comp / 1e3, 1) if comp else None,
                      "fp64_frac": round(alg / (ms * 1e-3) / 1e12 / FP64_PEAK_TFLOPS, 4) if ms > 0 else None,
                      "ns_per_computed_node": round(ms * 1e6 / comp, 1) if comp else None, "kernel": PLAN_KERNEL_NAMES[name]}
    return rows, alg_t, ref_t, unpriced


def kernel_source_hash():
    """sha256 over the kernel sources: stamps profiles/traffic.json, so a stale PMC figure is never reported."""
    h = hashlib.sha256()
    d = os.path.join(ROOT, "ninpol_amd", "csrc")
    for f in sorted(os.listdir(d)):
        if f.endswith((".hip", ".hpp", ".cpp")):
            h.update(f.encode())
            h.update(open(os.path.join(d, f), "rb").read())
    return h.hexdigest()[:16]


def cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def cpu_baseline(method, n_sample, jitter):
    """The CPU number beside the GPU one: our C restatement of the reference's method kernel (oracle/ninpol_oracle.c,
    kind "port" -- the reference's Cython never travels to the GPU box) on a bounded sample: the same mesh
    recipe at n_sample^3 cells.  Two runs: min(16, cores) OpenMP threads as gls.pyx:87 / idw.pyx:55 schedule it, and
    all cores.  `calibration` is t_port / t_reference measured in the dev container (tools/cpu_calibration.py)."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import ninpol_oracle as O
    from ninpol_amd import mesh as M
    ncpu = os.cpu_count() or 1
    cores = min(16, ncpu)
    m = M.hex_mesh(n_sample, jitter=jitter, seed=0)
    M.attach_fields(m, "u", perm="ALH")
    o = O.OracleInterpolator("port", threads=cores)
    o.load_mesh(m)
    P = o.grid.n_points
    dt = 1e30
    for _ in range(3):   # best of three: ~10 s of host work at the default 128^3 sample
        t0 = time.time()
        o.prepare(method, "u")
        dt = min(dt, time.time() - t0)
    out = {"value": round(P / dt / 1e6, 4), "unit": "Mnodes/s", "cores": cores, "kind": "port",
           "sample": f"{method.upper()} method kernel (C restatement of the reference, OpenMP) on a {n_sample}^3-cell hex "
                     f"mesh of the same recipe ({P} nodes, best of 3: {dt:.2f} s)",
           "cpu_model": cpu_model(), "logical_cpus": ncpu, "OMP_threads": cores, "OPENBLAS_NUM_THREADS": "1 (unused by the port)"}
    if ncpu > cores:
        o.threads = ncpu
        t0 = time.time()
        o.prepare(method, "u")
        dt2 = time.time() - t0
        out["all_cores"] = {"value": round(P / dt2 / 1e6, 4), "cores": ncpu, "seconds": round(dt2, 2)}
    try:
        cal = json.load(open(os.path.join(ROOT, "profiles", "cpu_calibration.json")))
        r = cal[f"{method}_t_port_over_t_ref"]
        out["calibration"] = {"t_port_over_t_ref": r, "reference_equivalent_value": round(out["value"] * r, 4),
                              "measured_on": f"{cal['cpu_model']} ({cal['threads']} threads, dev container, "
                                             "tools/cpu_calibration.py)"}
    except Exception:
        out["calibration"] = None
    return out


def relaunch_command(n_gpus, argv, port=None):
    """`python bench.py --gpus N` from a bare shell (WORLD_SIZE unset): the command the parent starts as a CHILD process --
    the driver's own launch line, one rank per GPU over RCCL.  The parent has not touched the GPU (torch is imported later)."""
    if port is None:
        import socket
        with socket.socket() as so:
            so.bind(("127.0.0.1", 0))
            port = so.getsockname()[1]
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n_gpus}",
            "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + list(argv)


def self_launch(n_gpus, argv):
    """Run the N-rank job as a child, relay its ONE JSON line (rank 0's) on stdout, everything else on stderr, and its exit code."""
    import subprocess
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")     # dmabuf IPC: RCCL across processes needs it on this host driver
    cmd = relaunch_command(n_gpus, argv)
    print("bench.py: WORLD_SIZE unset and --gpus %d: starting %s" % (n_gpus, " ".join(cmd)), file=sys.stderr, flush=True)
    child = subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=None, env=env, text=True)
    line = None
    for out in child.stdout:
        t = out.strip()
        if t.startswith("{") and '"metric"' in t:
            line = t
        else:
            sys.stderr.write(out)
    rc = child.wait()
    if line is not None:
        print(line, flush=True)
    elif rc == 0:
        print("bench.py: the child exited 0 without a JSON line", file=sys.stderr)
        rc = 1
    return rc


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--method", default="gls", choices=["gls", "idw", "ls"])
    ap.add_argument("--edge", dest="n", type=int, default=216, help="cells per edge per GPU (216^3 = 10,077,696)")
    ap.add_argument("--edge-global", type=int, default=0,
                    help="strong scaling: the WHOLE mesh is G^3 cells, its node planes split over the ranks (e.g. 432 with --gpus 8)")
    ap.add_argument("--jitter", type=float, default=0.15)
    ap.add_argument("--cpu-sample", type=int, default=128, help="edge of the CPU-baseline sample mesh (0 = skip)")
    ap.add_argument("--grid-build", default="host", choices=["host", "device"],
                    help="where the Grid connectivity is built (north_star: host, pushed to HBM; device = SURVEY 8 f1)")
    ap.add_argument("--no-other-meshes", action="store_true",
                    help="skip the GLS-on-tet/mixed-mesh context (profiling runs: keeps per-kernel averages clean)")
    ap.add_argument("--no-extras", action="store_true", help="skip the IDW/LS context numbers and the e2e timing")
    ap.add_argument("--no-e2e", action="store_true",
                    help="skip the end-to-end interpolate() timing (profiling runs: it launches the kernels in four pieces, which would dilute per-dispatch averages)")
    ap.add_argument("--check", action="store_true",
                    help="(small --edge only) rank 0 recomputes the whole mesh on its GPU and compares the gathered triplets")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ and "RANK" not in os.environ:
        # the driver's shape `python bench.py --gpus N ...`: nothing has initialised the GPU in this process yet
        raise SystemExit(self_launch(args.gpus, sys.argv[1:]))

    import numpy as np
    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        if world == 1 and args.gpus > 1:
            raise SystemExit("--gpus N > 1 must be launched with torch.distributed.run (one rank per GPU)")
        raise SystemExit(f"--gpus {args.gpus} does not match WORLD_SIZE {world}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the weight kernels are HIP only (no CPU fallback)")
    # Rehearsal on a one-GPU box: NIN_BENCH_REHEARSAL=1 puts every rank on device 0 and runs the exchange over gloo
    # through host copies.  It exercises the slab / offset / padding logic only; it is never a measurement.
    rehearsal = os.environ.get("NIN_BENCH_REHEARSAL") == "1"
    if rehearsal:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        if rehearsal:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=dev)

    import ninpol_amd
    from ninpol_amd import mesh as M
    from ninpol_amd.partition import ShardedInterpolator, node_block

    strong = args.edge_global > 0
    n = args.edge_global if strong else args.n
    nz_global = n if strong else n * world
    zlen = 1.0 if strong else float(world)
    plane_lo, plane_hi = node_block(nz_global + 1, rank, world)
    t0 = time.time()
    mesh, node_off, cell_off, own_lo, own_hi = M.hex_slab(n, n, nz_global, plane_lo, plane_hi,
                                                          lengths=(1.0, 1.0, zlen), jitter=args.jitter, seed=0)
    M.attach_fields(mesh, "u", perm="ALH")
    t_gen = time.time() - t0
    t0 = time.time()
    # torch.distributed.run exports OMP_NUM_THREADS=1 to its workers; the host grid build is OpenMP code, so give
    # every rank its share of the host cores explicitly (setup only, outside the timed region)
    host_threads = max(1, (os.cpu_count() or 1) // world) if world > 1 else 0
    t_load_dev = None
    if world > 1:
        # the library's multi-GPU path (ninpol_amd/partition.py): slab loader -- no rank holds the whole mesh --,
        # device-resident triplets, columns / counts gathered once, values + Neumann array gathered per step
        S = ShardedInterpolator(device=local_rank, comm_on_host=rehearsal, grid_build=args.grid_build,
                                num_threads=host_threads)
        S.load_shard(mesh, node_off, cell_off, (own_lo, own_hi), (n + 1) * (n + 1) * (nz_global + 1), n * n * nz_global)
        I = S.local
        t_load = time.time() - t0
    else:
        I = ninpol_amd.Interpolator(device=local_rank, num_threads=host_threads, grid_build=args.grid_build)
        I.load_mesh(mesh_obj=mesh)
        t_load = time.time() - t0
        if not args.no_extras and args.grid_build == "host":
            # context: the same load_mesh with the connectivity built by HIP kernels on the GPU
            t0 = time.time()
            I2 = ninpol_amd.Interpolator(device=local_rank, grid_build="device")
            I2.load_mesh(mesh_obj=mesh)
            t_load_dev = time.time() - t0
            del I2
    del mesh
    g = I.grid
    t0 = time.time()
    if world > 1:
        splan = S.device_plan("u", args.method)
        plan = splan.plan
    else:
        plan = I.device_plan("u", args.method)
    t_push = time.time() - t0
    P_loc, n_owned = g.n_points, own_hi - own_lo
    gls_counts = g.gls_plan() if args.method == "gls" else None

    stream = torch.cuda.current_stream()
    if world == 1:
        out = torch.empty(plan.nnz, dtype=torch.float64, device=dev)
        nws = torch.empty(P_loc, dtype=torch.float64, device=dev)

    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]

    def step(i=None):
        if world > 1:   # kernel + the single exchange step of the path, asynchronous (two rotating buffer sets)
            splan.step(ev[i] if i is not None else None)
            return
        if i is not None:
            ev[i][0].record(stream)
        plan.launch(out.data_ptr(), nws.data_ptr(), stream.cuda_stream, add_neumann=True)
        if i is not None:
            ev[i][1].record(stream)

    def drain_all():
        if world > 1:
            splan.drain_all()

    for _ in range(args.warmup):
        step()
    drain_all()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(args.steps):
        step(i)
    drain_all()                   # every step's triplets have been delivered on every rank
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    kern_ms = sum(a.elapsed_time(b) for a, b in ev) / max(args.steps, 1)
    if world > 1:
        rdev = "cpu" if rehearsal else dev
        t = torch.tensor([elapsed, kern_ms], dtype=torch.float64, device=rdev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed, kern_ms = float(t[0]), float(t[1])
        tot = torch.tensor([n_owned], dtype=torch.int64, device=rdev)
        dist.all_reduce(tot)
        total_nodes = int(tot[0])
    else:
        total_nodes = n_owned

    if world > 1:   # which device each rank really sits on
        dv = [None] * world
        dist.all_gather_object(dv, f"cuda:{torch.cuda.current_device()}")
        ranks_devices = dv
    else:
        ranks_devices = [f"cuda:{torch.cuda.current_device()}"]

    def timed_leg(body, finish):
        """K calls of body(i) + finish(), bracketed like the main loop; max over ranks, in ms per call."""
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for i in range(args.steps):
            body(i)
        finish()
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t1
        if world > 1:
            tt = torch.tensor([dt], dtype=torch.float64, device="cpu" if rehearsal else dev)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            dt = float(tt[0])
        return dt / max(args.steps, 1) * 1e3

    decomposition = {}
    if world > 1:
        # (1) the kernels alone, no exchange; (2) one step's exchange alone (its buffers hold the last step's values);
        # (3) the sharded apply: W_block . u on the device + the all-gather of node values (ShardedPlan.apply_step)
        decomposition["compute_only_ms"] = timed_leg(lambda i: splan.step(exchange=False), lambda: None)
        decomposition["exchange_ms"] = timed_leg(lambda i: (splan.drain(i % 2), splan.exchange(i % 2)), splan.drain_all)
        # (2b) the same exchange as direct peer-to-peer writes through the C ABI (nin_exchange_*: every rank's UNPADDED block straight into
        # the peers' gathered buffers, one copy stream per peer -- seven xGMI links at once).  Measured beside the all-gather, never
        # `value`'s path; a failure here must not cost the line
        try:
            from ninpol_amd.partition import P2PExchange
            px = P2PExchange(local_rank, rank, world, 8 * max(splan.mx_nnz, 1))
            blk = lambda i: splan.out[i % 2][splan.eb:splan.ee]
            px.push(blk(0))
            px.complete()
            ok = bool(torch.equal(px.slot(rank, torch.float64, splan.ee - splan.eb), blk(0)))
            decomposition["exchange_p2p_ms"] = timed_leg(lambda i: px.push(blk(i)), px.complete)
            decomposition["exchange_p2p_own_slot_ok"] = ok
            px.close()
        except Exception as e:   # noqa: BLE001
            decomposition["exchange_p2p_error"] = f"{type(e).__name__}: {e}"[:300]
        u_loc = torch.from_numpy(np.ascontiguousarray(
            np.asarray(I.cells_data[I.variable_to_index["cells"]["u"]])[:g.n_elems].reshape(1, -1))).to(dev)
        splan.apply_step(u_loc)
        splan.drain_all()
        decomposition["apply_ms"] = timed_leg(lambda i: splan.apply_step(u_loc), splan.drain_all)
    else:
        u_loc = torch.from_numpy(np.ascontiguousarray(
            np.asarray(I.cells_data[I.variable_to_index["cells"]["u"]])[:g.n_elems].reshape(1, -1))).to(dev)
        node_vals = torch.empty((1, P_loc), dtype=torch.float64, device=dev)
        plan.launch_apply(u_loc.data_ptr(), 1, node_vals.data_ptr(), nws.data_ptr(), stream.cuda_stream)
        decomposition["apply_ms"] = timed_leg(
            lambda i: plan.launch_apply(u_loc.data_ptr(), 1, node_vals.data_ptr(), nws.data_ptr(), stream.cuda_stream), lambda: None)
        del node_vals

    check = None
    if args.check and world > 1 and rank == 0:
        # reassemble what the all-gather delivered and compare with the whole mesh computed on this GPU
        whole = M.hex_mesh(n, n, nz_global, lengths=(1.0, 1.0, zlen), jitter=args.jitter, seed=0)
        M.attach_fields(whole, "u", perm="ALH")
        Iw = ninpol_amd.Interpolator(device=local_rank)
        Iw.load_mesh(mesh_obj=whole)
        pw = Iw.device_plan("u", args.method)
        ow = torch.empty(pw.nnz, dtype=torch.float64, device=dev)
        nw = torch.empty(pw.n_points, dtype=torch.float64, device=dev)
        pw.launch(ow.data_ptr(), nw.data_ptr(), stream.cuda_stream, add_neumann=True)
        torch.cuda.synchronize()
        b_last = (splan.n_steps - 1) % 2
        cat = lambda t, which: torch.cat(splan.pieces(t, which)).cpu().numpy()
        gw = Iw.grid
        check = bool(np.array_equal(cat(splan.counts, 1), np.diff(gw.esup_ptr)) and np.array_equal(cat(splan.cols, 0), gw.esup)
                     and np.array_equal(cat(splan.vals[b_last], 0), ow.cpu().numpy())
                     and np.array_equal(cat(splan.neumann[b_last], 1), nw.cpu().numpy()))

    if rank == 0:
        ms_per_step = elapsed / args.steps * 1e3
        value = total_nodes * args.steps / elapsed / 1e6
        B_alg = plan.algorithmic_bytes
        ach = B_alg / (kern_ms * 1e-3) / 1e9
        traffic = traffic_bounds = None   # PMC bytes per launch, only if recorded for exactly these kernel sources
        tpath = os.path.join(ROOT, "profiles", "traffic.json")
        if os.path.exists(tpath):
            try:
                tj = json.load(open(tpath))
                if tj.get("kernel_source_sha16") == kernel_source_hash():
                    traffic = tj.get(f"{args.method}_n{n}_bytes_per_launch")
                    traffic_bounds = tj.get(f"{args.method}_n{n}_bounds")
            except Exception:
                traffic = None
        hbm = {"achieved": round(ach, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(ach / HBM_PEAK_GBS, 5)}
        if args.method == "gls":
            # GLS is FP64-ALU-bound, not HBM-bound (SURVEY finding 2): the roofline of record is the FP64 vector peak,
            # priced on the flops the kernel EXECUTES; the HBM fraction (BASELINE's metric) rides beside it
            # the flops are priced on the nodes the kernel PROCESSES (Dirichlet boundary nodes are skipped: 2.7 % at 216^3)
            n_cube = gls_counts["hex8"]
            alg = ALG_FLOPS["cube"] * n_cube / (kern_ms * 1e-3) / 1e12
            ex = EXEC_FLOPS_PER_NODE_HEX * n_cube / (kern_ms * 1e-3) / 1e12
            roof = {"bound": "fp64", "achieved": round(alg, 3), "peak": FP64_PEAK_TFLOPS, "unit": "TFLOP/s",
                    "frac": round(alg / FP64_PEAK_TFLOPS, 4), "algorithmic_flops_per_node": ALG_FLOPS["cube"],
                    "nodes_processed": n_cube,
                    "executed_flops_per_node": EXEC_FLOPS_PER_NODE_HEX, "executed_tflops": round(ex, 3),
                    "executed_frac": round(ex / FP64_PEAK_TFLOPS, 4),
                    "hbm": hbm, "hbm_frac": hbm["frac"]}
        else:
            roof = dict(hbm, bound="hbm")
        roof.update({"traffic": traffic, "traffic_bounds": traffic_bounds, "kernel": plan.kernel_name, "kernel_ms": round(kern_ms, 4),
                     "algorithmic_bytes_per_launch": B_alg, "bytes_per_node": round(B_alg / P_loc, 1)})
        line = {
            "metric": "Mnodes/s interpolated (GLS, 10M-cell hex mesh) + achieved HBM GB/s vs peak"
                      if args.method == "gls" and n == 216 else f"Mnodes/s interpolated ({args.method.upper()}, {n}^3-cell hex mesh per GPU)",
            "value": round(value, 3), "unit": "Mnodes/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 4), "higher_is_better": True, "scaling": "strong" if strong else "weak",
            "vs_baseline": None,
            "dtype": "f64", "data": "synthetic",
            # what the process group itself reports (N > 1), so that a line can be checked against the launch
            "ranks_seen": {"world_size": dist.get_world_size() if world > 1 else 1,
                           "backend": dist.get_backend() if world > 1 else None,
                           "devices": ranks_devices},
            # key definitions (advisor, round 3): 3 = roofline.achieved / frac are ALGORITHMIC flops x cube nodes (round 2: executed
            # flops x all local nodes); e2e_interpolate_s = best of 3 steady-state calls (round 2: the second call), the median rides
            # beside it; 4 = every GLS node of a baseline_configs row is priced per kernel (`gls_kernels`)
            "metric_version": 4,
            "config": {"workload": f"{args.method.upper()} interpolate() weights, {n}x{n}x{nz_global} hexahedra "
                                   f"({n * n * nz_global} cells, {total_nodes} nodes), jitter {args.jitter}, ALH permeability, "
                                   "all-Dirichlet boundary; inputs resident in HBM, output CSR values on device"
                                   + ("; + RCCL all-gather of the (count, column, value) triplets and the Neumann array: columns "
                                      "and counts once per mesh, values per step, overlapped with the next step's kernel "
                                      "(ShardedPlan, two buffer sets)" if world > 1 else ""),
                       "cells_per_gpu": n * n * nz_global // world, "nodes_total": total_nodes, "nnz_esup_rank0": plan.nnz,
                       "parallelism": f"node-block shards x{world}, neighbour cells replicated" if world > 1 else "single GPU"},
            "roofline": roof,
            "setup_s": {"mesh_gen": round(t_gen, 2), "load_mesh": round(t_load, 2), "push_to_hbm": round(t_push, 2),
                        "grid_build": args.grid_build},
        }
        if "compute_only_ms" in decomposition:
            line["compute_only_Mnodes_s"] = round(total_nodes / decomposition["compute_only_ms"] / 1e3, 3)
            line["exchange_ms"] = round(decomposition["exchange_ms"], 4)
            if "exchange_p2p_ms" in decomposition:
                line["exchange_p2p_ms"] = round(decomposition["exchange_p2p_ms"], 4)
                line["exchange_p2p_note"] = ("the same value blocks as direct peer-to-peer writes (nin_exchange_*, no padding, one copy stream per "
                                             f"peer), timed alone; own slot verified: {decomposition['exchange_p2p_own_slot_ok']}")
            if "exchange_p2p_error" in decomposition:
                line["exchange_p2p_error"] = decomposition["exchange_p2p_error"]
            line["exchange_note"] = ("all-gather of one step's CSR values (8 B per entry" +
                                     (" + 8 B per row of neumann_ws" if splan.gather_neumann else "") +
                                     f"), timed alone: {splan.mx_nnz * 8 * (world - 1) / 1e9:.3f} GB received per GPU")
        line["apply_Mnodes_s"] = round(total_nodes / decomposition["apply_ms"] / 1e3, 3)
        line["apply_ms_per_step"] = round(decomposition["apply_ms"], 4)
        line["apply_note"] = ("W . u for one cell field: weights + row-block apply on the device" +
                              (f" + ONE all-gather of node values ({splan.mx_rows * 8 * (world - 1) / 1e6:.1f} MB received per GPU)"
                               if world > 1 else "") + "; what the reference's callers do with the matrix (analytical.py:236)")
        if t_load_dev is not None:
            line["setup_s"]["load_mesh_with_device_grid_build"] = round(t_load_dev, 2)
        if check is not None:
            line["gather_check_bit_identical_to_single_gpu"] = check
        if rehearsal:
            line["data"] = "synthetic (REHEARSAL: all ranks on one GPU, gloo through host copies -- not a measurement)"
        if args.method == "gls":
            flops = REF_FLOPS_PER_NODE_HEX * gls_counts["hex8"] / (kern_ms * 1e-3) / 1e12
            line["fp64"] = {"ref_equiv_tflops": round(flops, 3), "peak_tflops": FP64_PEAK_TFLOPS,
                            "ref_equiv_frac": round(flops / FP64_PEAK_TFLOPS, 4),
                            "algorithmic_tflops": roof["achieved"], "algorithmic_frac": roof["frac"],
                            "executed_tflops": roof["executed_tflops"], "executed_frac": roof["executed_frac"],
                            "note": "ref_equiv prices the node rate at the reference's dense dgels 44x25x8 = 74.8 kflop/node "
                                    "(SURVEY 8d); the multifrontal formulation needs 15.9 kflop/node (algorithmic) and the kernel "
                                    "executes 16.7 (the reflector scalars in all four lanes of a node, the quad sums), so ref_equiv may pass 1"}
        if world == 1 and not args.no_extras:
            # context: the two HBM-bound methods on the same grid, and end-to-end interpolate()
            for meth in ("idw", "ls"):
                if meth == args.method:
                    continue
                p2 = I.device_plan("u", meth)
                for _ in range(3):
                    p2.launch(out.data_ptr(), nws.data_ptr(), stream.cuda_stream)
                # an event pair around EVERY launch, as the main loop does (one pair around ten 0.4 ms launches also counts the gaps
                # between them: 0.45 against the 0.41 ms rocprofv3 reports for the kernel)
                pairs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(10)]
                for a, b in pairs:
                    a.record(stream)
                    p2.launch(out.data_ptr(), nws.data_ptr(), stream.cuda_stream)
                    b.record(stream)
                torch.cuda.synchronize()
                ms = sum(a.elapsed_time(b) for a, b in pairs) / len(pairs)
                line[meth] = {"kernel_ms": round(ms, 4), "Mnodes_per_s": round(P_loc / ms / 1e3, 1),
                              "achieved_GBps": round(p2.algorithmic_bytes / (ms * 1e-3) / 1e9, 1),
                              "frac_hbm": round(p2.algorithmic_bytes / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)}
            if args.method == "gls":
                # the kernels of the north-star mesh's own launch plan, one by one (the cube-node kernel; the Dirichlet boundary
                # nodes' passes through the quad-node / small-node kernels)
                def _time(nrep):
                    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    a.record(stream)
                    for _ in range(nrep):
                        plan.launch(out.data_ptr(), nws.data_ptr(), stream.cuda_stream, add_neumann=True)
                    b.record(stream)
                    torch.cuda.synchronize()
                    return a.elapsed_time(b) / nrep
                krows, alg_t, ref_t, unpriced = gls_kernel_rows(g, lambda: plan.launch(out.data_ptr(), nws.data_ptr(), stream.cuda_stream, add_neumann=True), _time, 5)
                line["gls_kernels"] = krows
                line["nodes_not_priced"] = int(unpriced)
            del out, nws
            torch.cuda.empty_cache()
            # host-buffer (PCIe-inclusive) path, never `value`: the first call also pins its output buffers (recycled afterwards)
            if not args.no_e2e:
                t0 = time.time()
                W, _ = I.interpolate("u", args.method)
                line["e2e_interpolate_first_s"] = round(time.time() - t0, 3)
                del W
                ts = []
                for _ in range(5):      # steady state: the page-locked buffers of the first call are recycled
                    t0 = time.time()
                    W, _ = I.interpolate("u", args.method)
                    ts.append(time.time() - t0)
                    line["e2e_nnz"] = int(W.nnz)
                    del W
                line["e2e_interpolate_s"] = round(min(ts), 4)
                line["e2e_interpolate_median_s"] = round(sorted(ts)[len(ts) // 2], 4)
            del I
            # one row per single-GPU entry of BASELINE.json's `configs` (kernel only, HIP events): [1] IDW and
            # [2] GLS on the 1 M-cell hexahedron mesh, [3] GLS on the 10 M-cell hex | pyramid | tet mix; plus the Kuhn-tet
            # mesh the block kernel is tuned on.  ([0] is the reference's CPU plumbing case, [4] the 8-GPU run: --gpus 8.)
            if args.method == "gls" and not args.no_other_meshes:
                def timed(Io, meth, reps=3):
                    po = Io.device_plan("u", meth)
                    oo = torch.empty(po.nnz, dtype=torch.float64, device=dev)
                    no = torch.empty(po.n_points, dtype=torch.float64, device=dev)
                    launch = lambda: po.launch(oo.data_ptr(), no.data_ptr(), stream.cuda_stream)

                    def time_launches(n):
                        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                        a.record(stream)
                        for _ in range(n):
                            launch()
                        b.record(stream)
                        torch.cuda.synchronize()
                        return a.elapsed_time(b) / n
                    launch()
                    ms = time_launches(reps)
                    gb = po.algorithmic_bytes / (ms * 1e-3) / 1e9
                    row = {"kernel_ms": round(ms, 3), "Mnodes_per_s": round(Io.grid.n_points / ms / 1e3, 2),
                           "achieved_GBps": round(gb, 1), "frac_hbm": round(gb / HBM_PEAK_GBS, 4)}
                    if meth == "gls":   # the roof that binds GLS: FP64 vector, algorithmic flops of the formulation each kernel runs
                        krows, alg, ref, unpriced = gls_kernel_rows(Io.grid, launch, time_launches, reps)
                        row.update({"fp64_algorithmic_tflops": round(alg / (ms * 1e-3) / 1e12, 3),
                                    "fp64_frac": round(alg / (ms * 1e-3) / 1e12 / FP64_PEAK_TFLOPS, 4),
                                    "fp64_ref_equiv_frac": round(ref / (ms * 1e-3) / 1e12 / FP64_PEAK_TFLOPS, 4),
                                    "nodes_not_priced": int(unpriced), "gls_kernels": krows})
                    return row
                rows = {}
                for name, make, meths in (
                        ("[1],[2] hex 100^3", lambda: M.hex_mesh(100, jitter=args.jitter), ("idw", "gls")),
                        ("[3] hex|pyramid|tet mix 200x120x120", lambda: M.mixed_mesh(200, 120, 120, jitter=0.1), ("gls",)),
                        ("kuhn tets 40^3", lambda: M.tet_mesh(40, jitter=0.1), ("gls",)),
                        # round 4: UNSTRUCTURED tetrahedra, the mesh class of the reference's tetra numbers (performance.yaml:184-246):
                        # the Delaunay tetrahedrisation of a jittered body-centred cloud, ~2 M cells, 14 .. 40 cells around a node
                        ("unstructured tets (Delaunay of a jittered 54^3 body-centred cloud)", lambda: M.delaunay_tet_mesh(54, seed=0), ("gls",)),
                        # ... and of a uniformly RANDOM cloud (what scipy.spatial.Delaunay of random points gives: up to ~60 cells around a
                        # node, 7 % of the interior nodes beyond the wide kernel's 16 + 21 cells -> kernels_gls_mfg.hip, tiles in global memory)
                        ("unstructured tets (Delaunay of a uniformly random cloud, as many points as a 40^3 body-centred one)",
                         lambda: M.delaunay_tet_mesh(40, seed=0, lattice="random"), ("gls",)),
                        # UNSTRUCTURED prisms (the reference's "prism" family): a 2-D Delaunay triangulation extruded -- node valence 4 .. 8:
                        # 8 wedges = the cube graph, 12 / 16 two-coloured, 10 / 14 neither (the wide kernel's small class)
                        ("unstructured prisms (2-D Delaunay of a jittered 100^2 grid, 60 layers of wedges)",
                         lambda: M.delaunay_wedge_mesh(100, 60, seed=0), ("gls",))):
                    mo = make()
                    M.attach_fields(mo, "u", perm="ALH")
                    Io = ninpol_amd.Interpolator(device=local_rank, grid_build="device")
                    Io.load_mesh(mesh_obj=mo)
                    row = {"cells": int(Io.grid.n_elems), "nodes": int(Io.grid.n_points)}
                    for meth in meths:
                        row[meth] = timed(Io, meth)
                    row["gls_nodes_per_kernel"] = {k: v for k, v in Io.grid.gls_plan().items() if v}
                    n_int = int((~np.asarray(Io.grid.boundary_points).astype(bool)).sum())
                    pl = Io.grid.gls_plan()
                    row["interior_nodes_off_the_block_kernel"] = round(
                        (pl["hex8"] + pl["mfw_large"] + pl["mfw_small"] + pl["mfw_general"] + pl["mfx"] + pl["mfg_tiles"]) / max(n_int, 1), 4)   # (pl["mfx"]: interior classes only)
                    rows[name] = row
                    del Io, mo
                    torch.cuda.empty_cache()
                line["baseline_configs"] = rows
        if world == 1 and args.cpu_sample > 0:
            try:
                line["cpu_baseline"] = cpu_baseline(args.method, args.cpu_sample, args.jitter)
            except Exception as e:   # never lose the GPU line to the baseline leg
                line["cpu_baseline"] = {"value": None, "unit": "Mnodes/s", "cores": 0, "kind": "port",
                                        "sample": f"failed: {type(e).__name__}: {e}"}
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
