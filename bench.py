#!/usr/bin/env python3
"""bench.py -- headline benchmark of the MI355X RHS engine (contract: see the task statement).

A "step" is ONE explicit-RK right-hand-side evaluation (all phases of the hot path + halo exchange)
over the whole mesh with the state resident in HBM.  Workload at N=1: BASELINE.json configs[2],
"2D compressible Navier-Stokes, N=4, 512x512 quads, 1 MI355X" -- the configuration the metric and
the >=40 %-of-HBM-roofline target are quoted on: modal ESDG CNS algorithm of
examples/CompressibleNS/dg2D_CNS_cavity_optimized.jl (rhsRK!) on the reference quad element,
periodic isentropic-vortex box, Re=1000, both dissipations on.  For N>1 GPUs the mesh grows in y
(512 x 512N elements, one horizontal strip of 512x512 per rank = weak scaling) and the three face-trace
exchanges per RHS go over RCCL (torch.distributed "nccl").

--formulation hex runs BASELINE.json configs[4] per GPU instead: 3D hexahedral Euler, N=3, 128x128x16 elements per
GPU (z-slabs of the 128^3 box; `rhs` of examples/dg3D_euler_hex.jl, LF factor 0 as in the reference unless --lf).

Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0        # MI355X_MICROARCH.md: HBM3E peak 8.0 TB/s (spec)
FP64_VALU_PEAK_TFLOPS = 78.6  # 256 CU x 4 SIMD x 16 fp64 lanes x 2 x 2.4 GHz


def build_problem(N, Kx, Ky_total, e0, e1, formulation):
    from esdg_cns_amd import physics as ph
    from esdg_cns_amd import setup_dg as sd
    VX, VY, EToV = sd.uniform_quad_mesh(Kx, Ky_total)
    VX = 15 * (1 + VX) / 2
    VY = 5 * VY * (Ky_total / Kx)          # keep square elements of the 512^2 box as the mesh grows in y
    if formulation == "euler":
        rd = sd.init_reference_quad(N, sd.gauss_quad(0, 0, N))
        ops = sd.euler_quad_ops(rd)
    else:
        rd = sd.init_reference_quad(N)
        ops = sd.cns_ops(rd)
    md = sd.init_mesh((VX, VY), EToV, rd, elem_range=(e0, e1))
    sd.make_periodic(md, rd)
    md.mapB = np.zeros(0, dtype=np.int64)
    sd.interp_geofacs_to_hybrid(md, ops["Vh"])
    xx, yy = (md.xq, md.yq) if formulation == "euler" else (md.x, md.y)
    rho, u, v, p = ph.vortex(xx, yy, 0)
    Q = [np.asfortranarray(q) for q in ph.primitive_to_conservative(rho, u, v, p)]
    return rd, md, ops, Q


def build_hex_problem(N, Kx, Ky, Kz_total, e0, e1, curve=0.0):
    from esdg_cns_amd import physics as ph
    from esdg_cns_amd import setup_dg as sd
    VX, VY, VZ, EToV = sd.uniform_hex_mesh(Kx, Ky, Kz_total)
    rd = sd.init_reference_hex(N, sd.gauss_quad(0, 0, N))
    md = sd.init_mesh_3d((VX, VY, VZ), EToV, rd, elem_range=(e0, e1))
    sd.make_periodic_3d(md, rd)
    ops = sd.hex_ops(rd)
    # affine: one metric row per element (geo_ld = 1); curved (--hex-curve a, the script's mapping :67-73): all Nh rows
    sd.hex_driver_geometry(md, rd, hybrid=bool(curve), a=curve)
    x, y, z = md.xq, md.yq, md.zq
    rho = 2 + .5 * np.sin(np.pi * x) * np.cos(np.pi * y)
    u, v, w = .3 * np.sin(np.pi * z + .2), 1 + .1 * np.cos(np.pi * x), .1 * np.sin(np.pi * (x + y) + .3)
    p = 1 + .2 * np.cos(np.pi * z) * np.sin(np.pi * y)
    Q = [np.asfortranarray(q) for q in ph.primitive_to_conservative_3d(rho, u, v, w, p)]
    return rd, md, ops, Q


def cpu_baseline_hex(N, lf, budget_s=15.0):
    """oracle_hex_rhs (C restatement of dg3D_euler_hex.jl:122-222: 1344 flux evaluations per element at N=3), one
    thread, on a 12^3 sample of the same periodic box."""
    from oracle import oracle as orc
    Ks = 12
    p = orc.build_hex_problem(N, Ks)
    orc.lib().oracle_set_threads(1)
    o = orc.HexOracle(p, lf)
    Qs = orc.stack(p.Q)
    o.rhs_stacked(Qs)
    t0 = time.perf_counter()
    n = 0
    while True:
        o.rhs_stacked(Qs)
        n += 1
        if time.perf_counter() - t0 > budget_s or n >= 50:
            break
    dt = (time.perf_counter() - t0) / n
    K, Np = p.md.K, (N + 1) ** 3
    nthr = min(int(orc.lib().oracle_get_max_threads()), usable_cpus())   # OpenMP over elements on the usable host cores
    orc.lib().oracle_set_threads(nthr)
    o.rhs_stacked(Qs)
    t1 = time.perf_counter()
    m = 0
    while True:
        o.rhs_stacked(Qs)
        m += 1
        if time.perf_counter() - t1 > 5.0 or m >= 50:
            break
    dtm = (time.perf_counter() - t1) / m
    orc.lib().oracle_set_threads(1)
    return {"value": K * Np / dt, "unit": "DOF updates/s", "cores": 1, "kind": "port",
            "sample": f"hex N={N} {Ks}^3 periodic box, {n} RHS evals of oracle/oracle_rhs.c:oracle_hex_rhs "
                      f"(C restatement of the Julia reference, 1 thread), {dt * 1e3:.1f} ms/eval",
            "rhs_evals_per_s_at_sample": 1.0 / dt,
            "all_cores": {"value": K * Np / dtm, "cores": nthr, "ms_per_eval": dtm * 1e3}}


PREWARM_EVALS = 200   # untimed RHS evaluations before the warm-up steps (GPU clock ramp), see main()


def usable_cpus():
    """Host cores this process may actually use: affinity mask and cgroup CPU quota (a GPU box hands each job a share
    of a 128-core host; 128 OpenMP threads on a 16-core share run slower than one)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:                                                   # cgroup v2
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        try:                                               # cgroup v1
            quota = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            period = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if quota > 0:
                n = min(n, max(1, quota // period))
        except (OSError, ValueError):
            pass
    return max(1, n)


def cpu_baseline(N, formulation, budget_s=15.0):
    """Reference algorithm restated in C (oracle/oracle_rhs.c, the reference's loop structure: 825
    visited pairs/element in flux_differencing!), timed single-threaded like the Julia reference, on a
    bounded sample of the same workload (same N, same vortex box, fewer elements)."""
    from oracle import oracle as orc
    Ks = 96
    p = orc.build_cns_problem(N, Ks, Ks, bc="periodic") if formulation == "cns" else orc.build_euler_problem(N, Ks, Ks)
    orc.lib().oracle_set_threads(1)
    if formulation == "cns":
        o = orc.CnsOracle(p)
        Qs = orc.stack(p.Q)
        fn = lambda: o.rhsRK_stacked(Qs, False)
    else:
        o = orc.EulerOracle(p)
        Qs = orc.stack(p.Q)
        fn = lambda: o.rhs_stacked(Qs)
    fn()
    t0 = time.perf_counter()
    n = 0
    while True:
        fn()
        n += 1
        if time.perf_counter() - t0 > budget_s or n >= 50:
            break
    dt = (time.perf_counter() - t0) / n
    K, Np = p.md.K, (N + 1) ** 2
    # the same restatement with OpenMP over elements on every host core (variant ii)
    nthr = min(int(orc.lib().oracle_get_max_threads()), usable_cpus())
    orc.lib().oracle_set_threads(nthr)
    fn()
    t1 = time.perf_counter()
    m = 0
    while True:
        fn()
        m += 1
        if time.perf_counter() - t1 > 5.0 or m >= 50:
            break
    dtm = (time.perf_counter() - t1) / m
    orc.lib().oracle_set_threads(1)
    return {"value": K * Np / dt, "unit": "DOF updates/s", "cores": 1, "kind": "port",
            "sample": f"{formulation} N={N} {Ks}x{Ks} periodic vortex box, {n} RHS evals of oracle/oracle_rhs.c "
                      f"(C restatement of the Julia reference, 1 thread), {dt * 1e3:.1f} ms/eval",
            "rhs_evals_per_s_at_sample": 1.0 / dt,
            "all_cores": {"value": K * Np / dtm, "cores": nthr, "ms_per_eval": dtm * 1e3}}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--N", type=int, default=None, help="degree (default 4; 3 for hex)")
    ap.add_argument("--kx", type=int, default=None, help="elements in x (default 512; 128 for hex, also used for y)")
    ap.add_argument("--ky-per-gpu", type=int, default=512, help="element rows per GPU (weak scaling, 2D)")
    ap.add_argument("--kz-per-gpu", type=int, default=16, help="element layers per GPU (weak scaling, hex)")
    ap.add_argument("--lf", type=float, default=0.0, help="hex: LF factor (the reference has 0*.25)")
    ap.add_argument("--hex-curve", type=float, default=0.0, help="hex: amplitude a of the script's curved mapping (0 = affine)")
    ap.add_argument("--formulation", choices=["cns", "euler", "hex"], default="cns")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend for N>1 (nccl = RCCL; gloo only to "
                    "rehearse the multi-rank logic on one GPU: traces are staged through the host)")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    from esdg_cns_amd import engine

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if rank == 0:
            print(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}; launch with torch.distributed.run", file=sys.stderr)
        sys.exit(2)
    ndev = torch.cuda.device_count()
    torch.cuda.set_device(local_rank % max(ndev, 1))
    if world > 1:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank % max(ndev, 1)))
        else:
            dist.init_process_group(args.backend)

    hexw = args.formulation == "hex"
    N = args.N if args.N is not None else (3 if hexw else 4)
    Kx = args.kx if args.kx is not None else (128 if hexw else 512)
    if hexw:
        Kz_total = args.kz_per_gpu * world
        rank_offsets = np.array([Kx * Kx * args.kz_per_gpu * r for r in range(world + 1)], dtype=np.int64)   # z-slabs
        e0, e1 = int(rank_offsets[rank]), int(rank_offsets[rank + 1])
        rd, md, ops, Q = build_hex_problem(N, Kx, Kx, Kz_total, e0, e1, args.hex_curve)
        eng = engine.RhsEngine(rd, md, ops, engine.EULER_HEX_COLLOCATED, lf_scale=args.lf, rank=rank, nranks=world,
                               rank_offsets=rank_offsets)
        K_total = Kx * Kx * Kz_total
    else:
        Ky_total = args.ky_per_gpu * world
        rows = [args.ky_per_gpu * r for r in range(world + 1)]
        rank_offsets = np.array([Kx * r for r in rows], dtype=np.int64)   # elements are numbered x-fastest
        e0, e1 = int(rank_offsets[rank]), int(rank_offsets[rank + 1])
        rd, md, ops, Q = build_problem(N, Kx, Ky_total, e0, e1, args.formulation)
        form = engine.CNS_MODAL if args.formulation == "cns" else engine.EULER_COLLOCATED
        eng = engine.RhsEngine(rd, md, ops, form, rank=rank, nranks=world, rank_offsets=rank_offsets)
        K_total = Kx * Ky_total
    Qd = eng.upload(Q)
    out = eng.new_state()
    Np = eng.Np
    K_local = eng.K
    nfld = eng.nfld

    def sync_all():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    # Clock ramp: an idle MI355X needs some tens of milliseconds of load before it reaches its sustained clocks (measured:
    # the first ~50 evaluations run up to 25 % slower; 20 steps after 3 warm-ups 1.04 ms, after 50 warm-ups 0.92 ms).
    # A fixed number of untimed evaluations (the same on every rank) precedes the W warm-up steps so that the timed
    # region measures sustained throughput whatever W the caller picks; reported as config.prewarm_evals.
    for _ in range(PREWARM_EVALS):
        eng.rhs_into(Qd, out)
    sync_all()
    for _ in range(args.warmup):
        eng.rhs_into(Qd, out)
    sync_all()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        eng.rhs_into(Qd, out)
    sync_all()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    ms_per_step = elapsed / args.steps * 1e3
    evals_per_s = args.steps / elapsed
    value = K_total * Np * evals_per_s

    # --- roofline of the dominant kernel (last phase: k_rhs), timed live with events on the launch stream
    import ctypes as C
    last = eng.nphases - 1
    nrep = 20
    stream = torch.cuda.current_stream()
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(nrep)]
    for a, b in evs:
        for ph in range(last):
            engine.check(eng.L.esdg_rhs_phase(eng.ctx, ph, C.c_void_p(Qd.data_ptr()), C.c_void_p(out.data_ptr()), eng._stream()))
        a.record(stream)
        engine.check(eng.L.esdg_rhs_phase(eng.ctx, last, C.c_void_p(Qd.data_ptr()), C.c_void_p(out.data_ptr()), eng._stream()))
        b.record(stream)
    torch.cuda.synchronize()
    kdur_ms = float(np.mean([a.elapsed_time(b) for a, b in evs]))
    alg_bytes = 16.0 * nfld * Np * K_local       # read state once + write rhs once
    achieved = alg_bytes / (kdur_ms * 1e-3) / 1e9
    traffic = None
    prof_us = None
    pmc = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    if os.path.exists(pmc):
        try:
            key = f"hex_N{N}_{Kx}x{Kx}x{args.kz_per_gpu}" if hexw else f"{args.formulation}_N{N}_{Kx}x{args.ky_per_gpu}"
            rec = json.load(open(pmc)).get(key, {})
            traffic = rec.get("k_rhs_hbm_bytes_per_launch")
            prof_us = rec.get("k_rhs_rocprofv3_avg_us")
        except Exception:
            traffic = None
    roofline = {"bound": "hbm", "kernel": "kh_rhs (last phase: surface flux + flux differencing + lift)" if hexw else
                "k_rhs (last phase: flux differencing + viscous divergence)",
                "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                "traffic": traffic, "kernel_ms": kdur_ms,
                # the same kernel's average under rocprofv3 --kernel-trace --stats (committed profile of this command,
                # profiles/*_kernel_stats.csv): agrees with the live figure to 1-2 % on one box
                "kernel_ms_rocprofv3": None if prof_us is None else prof_us / 1e3,
                "whole_rhs_frac": (alg_bytes * evals_per_s / 1e9) / HBM_PEAK_GBS}

    if hexw:
        workload = f"euler3d_hex_N{N}_{Kx}x{Kx}x{Kz_total}_periodic_box_lf{args.lf:g}" + (f"_curved{args.hex_curve:g}" if args.hex_curve else "")
    else:
        workload = (f"{args.formulation}2d_N{N}_{Kx}x{Ky_total}_quads_periodic_vortex"
                    + ("_Re1000_inviscid+viscous_dissipation" if args.formulation == "cns" else ""))
    result = {
        "metric": "element-DOF updates/sec (RHS evals/s) at N=4, 2D CNS quad mesh" if not hexw else
                  "element-DOF updates/sec (RHS evals/s) at N=3, 3D hex Euler",
        "value": value, "unit": "DOF updates/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f64", "data": "synthetic",
        "config": {"workload": workload,
                   "elements": K_total, "elements_per_gpu": K_local, "Np": Np, "nfields": nfld,
                   "parallelism": f"element-{'slabs' if hexw else 'strips'} x{world}", "prewarm_evals": PREWARM_EVALS},
        "rhs_evals_per_s": evals_per_s, "elements_per_s": K_total * evals_per_s,
        "roofline": roofline,
    }
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        result["cpu_baseline"] = cpu_baseline_hex(N, args.lf) if hexw else cpu_baseline(N, args.formulation)
    elif rank == 0:
        result["cpu_baseline"] = None
    if rank == 0:
        print(json.dumps(result))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
