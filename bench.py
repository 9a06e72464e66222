#!/usr/bin/env python3
"""bench.py -- headline benchmark of the MI355X RHS engine (contract: see the task statement).

A "step" is ONE explicit-RK right-hand-side evaluation (all phases of the hot path + halo exchange) over the whole mesh
with the state resident in HBM.

  --gpus 1 (default): BASELINE.json configs[2], "2D compressible Navier-Stokes, N=4, 512x512 quads, 1 MI355X" -- the
      configuration the metric and the >=40 %-of-HBM-roofline target are quoted on: modal ESDG CNS algorithm of
      examples/CompressibleNS/dg2D_CNS_cavity_optimized.jl (rhsRK!) on the reference quad element, periodic isentropic-vortex
      box, Re=1000, both dissipations on.
  --gpus N > 1: BASELINE.json configs[3]'s per-GPU load, one horizontal strip of 2048x256 elements per rank (N = 8 IS
      "2D CNS N=4, 2048x2048 quads sharded across 8xMI355X"; weak scaling in between), the three face-trace exchanges per
      RHS over the library's own RCCL transport (esdg_comm_init; --transport torch drives torch.distributed P2P instead).
      `python bench.py --gpus N` launches its N ranks itself (a torch.distributed.run child, before this process touches
      a GPU); under an existing launcher (WORLD_SIZE set) it just runs as one rank.
  --formulation hex: BASELINE.json configs[4] per GPU, 3D hexahedral Euler, N=3, 128x128x16 elements per GPU (z-slabs of
      the 128^3 box; `rhs` of examples/dg3D_euler_hex.jl, LF factor 0 as in the reference unless --lf).

Prints ONE JSON line on rank 0.
"""
import argparse
import hashlib
import json
import os
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0         # MI355X_MICROARCH.md: HBM3E peak 8.0 TB/s (spec)
FP64_VALU_PEAK_TFLOPS = 78.6  # 256 CU x 4 SIMD x 16 fp64 lanes x 2 flop x 2.4 GHz
# What this pool's MI355X boxes deliver to the simplest kernels (tools/ubench, output committed as profiles/r03_ubench.txt):
# a plain device-to-device copy and independent fp64 FMAs from 4 waves x 8 chains per SIMD.  The two floors of
# `roofline.floors_ms` are traffic / PRACTICAL_HBM_GBS and fp64 flops / MEASURED_FP64_TFLOPS; `roofline.bound` names the larger.
PRACTICAL_HBM_GBS = 4700.0    # round 4: a hand-written float4 / double copy (read + write) on this pool, tools/ubench/fetch_calib.hip ->
                              # profiles/r04_fetch_calibration.txt: 4.6-4.8 TB/s; read-only streams 6.2, write-only 6.5-6.9 (round 3
                              # used a torch copy_ measured at 5.3 on other boxes of the pool)
# Read- and write-only streams on the same boxes (profiles/r04_fetch_calibration.txt): a kernel's HBM floor uses a rate weighted
# by ITS read share f = reads / (reads + writes), piecewise linear through the three calibrated points f = 0 (write-only), 0.5
# (the copy), 1 (read-only) -- the plain copy rate made a stream that is 84 % reads look memory-bound when it is not
READ_ONLY_HBM_GBS = 6200.0
WRITE_ONLY_HBM_GBS = 6700.0   # 6.5-6.9 measured
NORTH_STAR_TARGET_FRAC = 0.40  # BASELINE.json: ">= 40 % of MI355X fp64 HBM roofline" on cfg3 (whole RHS, algorithmic bytes)
MEASURED_FP64_TFLOPS = 66.5
VALU_CYCLES_PER_INST = 4.7    # measured issue cost of one fp64 wave-instruction per SIMD (tools/ubench/fma64.hip, 4 waves x 8 chains)
N_SIMD = 1024                 # 256 CUs x 4
SCLK_HZ = 2.4e9
PREWARM_EVALS = 200           # untimed RHS evaluations before the warm-up steps (GPU clock ramp), see run()
REPS = 5                      # repetitions of the K-step timed region (the first one is the contract's; median/min reported)


def build_problem(N, Kx, Ky_total, e0, e1, formulation):
    from esdg_cns_amd import physics as ph
    from esdg_cns_amd import setup_dg as sd
    VX, VY, EToV = sd.uniform_quad_mesh(Kx, Ky_total)
    VX = 15 * (1 + VX) / 2
    VY = 5 * VY * (Ky_total / Kx)          # keep square elements as the mesh grows in y
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


def rough_state(Q, seed=20250117):
    """SURVEY.md section 8(d) robustness variant of a 2D state: rho and p multiplied by 1 + 0.01 xi, xi in [-1, 1) from
    splitmix64(seed), xi = 2 (x >> 11) 2^-53 - 1, node-major order.  On this state no wave of the last-phase kernel finds its
    densities within 1e-4 of each other, so every log-mean takes the reference's logarithm branch and none of the
    data-dependent short cuts of kt3_rhs (all-series flux variant, logarithms skipped) applies."""
    rho, ru, rv, E = [np.array(q, dtype=np.float64, order="F") for q in Q]
    n = rho.size
    x = (np.arange(1, 2 * n + 1, dtype=np.uint64) * np.uint64(0x9E3779B97F4A7C15) + np.uint64(seed)) & np.uint64(0xFFFFFFFFFFFFFFFF)
    with np.errstate(over="ignore"):
        z = x
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        z = z ^ (z >> np.uint64(31))
    xi = 2.0 * (z >> np.uint64(11)).astype(np.float64) * 2.0 ** -53 - 1.0
    u, v = ru / rho, rv / rho
    p = 0.4 * (E - .5 * rho * (u * u + v * v))
    a = (1 + 0.01 * xi[:n]).reshape(rho.shape, order="F")
    b = (1 + 0.01 * xi[n:]).reshape(rho.shape, order="F")
    rho2, p2 = rho * a, p * b
    return [np.asfortranarray(q) for q in (rho2, rho2 * u, rho2 * v, p2 / 0.4 + .5 * rho2 * (u * u + v * v))]


def build_hex_problem(N, Kx, Ky, Kz_total, e0, e1, curve=0.0, per_node=False):
    from esdg_cns_amd import physics as ph
    from esdg_cns_amd import setup_dg as sd
    VX, VY, VZ, EToV = sd.uniform_hex_mesh(Kx, Ky, Kz_total)
    rd = sd.init_reference_hex(N, sd.gauss_quad(0, 0, N))
    md = sd.init_mesh_3d((VX, VY, VZ), EToV, rd, elem_range=(e0, e1))
    sd.make_periodic_3d(md, rd)
    ops = sd.hex_ops(rd)
    # affine: one metric row per element (geo_ld = 1); curved (--hex-curve a, the script's mapping :67-73): all Nh rows
    sd.hex_driver_geometry(md, rd, hybrid=bool(curve) or per_node, a=curve)
    x, y, z = md.xq, md.yq, md.zq
    rho = 2 + .5 * np.sin(np.pi * x) * np.cos(np.pi * y)
    u, v, w = .3 * np.sin(np.pi * z + .2), 1 + .1 * np.cos(np.pi * x), .1 * np.sin(np.pi * (x + y) + .3)
    p = 1 + .2 * np.cos(np.pi * z) * np.sin(np.pi * y)
    Q = [np.asfortranarray(q) for q in ph.primitive_to_conservative_3d(rho, u, v, w, p)]
    return rd, md, ops, Q


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


def _time_evals(fn, min_evals, budget_s, max_evals=50):
    fn()                                                   # untimed first call (page faults, thread start-up)
    ts = []
    t_end = time.perf_counter() + budget_s
    while len(ts) < min_evals or (time.perf_counter() < t_end and len(ts) < max_evals):
        t0 = time.perf_counter()
        fn()
        ts.append(time.perf_counter() - t0)
    return ts


def cpu_baseline(N, formulation, lf, rd, md, ops, Q, hexw):
    """The reference algorithm restated in C (oracle/oracle_rhs.c: the reference's loop structure, 825 visited pairs per
    element in flux_differencing!, 1344 in the hex sparse_hadamard_sum), timed on the SAME workload the GPU ran (same
    arrays): >= 3 single-threaded evaluations (the Julia reference is single-threaded), then OpenMP over elements on the
    usable host cores.  The reference itself (Julia) cannot run on this box."""
    from oracle import oracle as orc
    sample = f"the full GPU workload ({md.K} elements)"
    if hexw:
        # the oracle wants the nine metric arrays at all Nh hybrid nodes of every element (9 x 160 x K doubles); the GPU
        # workload keeps one row per affine element, so the hex baseline runs on a bounded 12^3 sample of the same box
        ps = orc.build_hex_problem(N, 12)
        rd, md, ops, Q = ps.rd, ps.md, ps.ops, ps.Q
        sample = f"a 12^3-element sample of the same periodic box ({md.K} elements)"
    p = orc.Problem()
    p.rd, p.md, p.ops, p.Q, p.N = rd, md, dict(ops), Q, N
    p.Re, p.mu, p.lam, p.Pr, p.BCTYPE = 1000.0, 1e-3, -2e-3 / 3, .71, 1
    o = p.ops
    if formulation == "euler" and "Qrsids" not in o:       # dg2D_euler_quad.jl:64
        o["Qrsids"] = []
        for i in range(o["Qrh_sparse"].shape[0]):
            a = list(np.nonzero(o["Qrh_sparse"][i])[0] + 1)
            o["Qrsids"].append(a + [j for j in list(np.nonzero(o["Qsh_sparse"][i])[0] + 1) if j not in a])
    if hexw and "Qnzids" not in o:                         # dg3D_euler_hex.jl:57-64
        for a, b in (("Qrh_sparse", "Qrhskew"), ("Qsh_sparse", "Qshskew"), ("Qth_sparse", "Qthskew")):
            M = np.array(o[b], dtype=float)
            M[np.abs(M) < 1e-12] = 0.0
            o[a] = M
        o["Qnzids"] = []
        for i in range(o["Qrh_sparse"].shape[0]):
            ids = []
            for M in (o["Qrh_sparse"], o["Qsh_sparse"], o["Qth_sparse"]):
                ids += [j for j in list(np.nonzero(M[i])[0] + 1) if j not in ids]
            o["Qnzids"].append(ids)
    L = orc.lib()
    Qs = orc.stack(Q)
    if hexw:
        oc = orc.HexOracle(p, lf)
        fn = lambda: oc.rhs_stacked(Qs)
        what = "oracle_hex_rhs"
    elif formulation == "cns":
        oc = orc.CnsOracle(p)
        fn = lambda: oc.rhsRK_stacked(Qs, False)
        what = "oracle_cns_rhsRK"
    else:
        oc = orc.EulerOracle(p)
        fn = lambda: oc.rhs_stacked(Qs)
        what = "oracle_euler_rhs"
    L.oracle_set_threads(1)
    t1 = _time_evals(fn, 3, 20.0)
    nthr = min(int(L.oracle_get_max_threads()), usable_cpus())
    L.oracle_set_threads(nthr)
    tm = _time_evals(fn, 3, 5.0)
    L.oracle_set_threads(1)
    dof = md.K * Q[0].shape[0]
    med = float(np.median(t1))
    return {"value": dof / med, "unit": "DOF updates/s", "cores": 1, "kind": "port",
            "sample": f"{sample}, {len(t1)} evaluations of oracle/oracle_rhs.c:{what} "
                      f"(C restatement of the Julia reference's loop structure, 1 thread): median {med * 1e3:.0f} ms, "
                      f"min {min(t1) * 1e3:.0f} ms per evaluation",
            "rhs_evals_per_s": 1.0 / med,
            "all_cores": {"value": dof / float(np.median(tm)), "cores": nthr, "ms_per_eval": float(np.median(tm)) * 1e3,
                          "evals": len(tm)}}


def mix_rate_gbs(read_bytes, write_bytes):
    """Practical HBM rate for a stream with the given read / write mix (see READ_ONLY_HBM_GBS above)."""
    tot = float(read_bytes) + float(write_bytes)
    if tot <= 0:
        return PRACTICAL_HBM_GBS
    f = float(read_bytes) / tot
    if f >= 0.5:
        return PRACTICAL_HBM_GBS + (READ_ONLY_HBM_GBS - PRACTICAL_HBM_GBS) * (f - 0.5) / 0.5
    return WRITE_ONLY_HBM_GBS + (PRACTICAL_HBM_GBS - WRITE_ONLY_HBM_GBS) * f / 0.5


GATHER_OVERFETCH = 1.6   # neighbour-trace gathers: 160-byte runs inside 640-byte blocks fetch two 128-byte lines per run
                         # (profiles/r04_fetch_calibration.txt, trace_nbr32 0.800 of the x2-corrected FETCH_SIZE)


def design_bytes(formulation, Np, Nfq, hexw=False, hex_delta=False):
    """Design bytes per element and evaluation, by array family and phase: what every phase reads (r) and writes (w) once, no
    cache reuse assumed; `g` marks the reads that are neighbour gathers (counted x GATHER_OVERFETCH in `with_overfetch`).
    DESIGN.md sections 3-4 (quads), 6 (hexahedra)."""
    if hexw:
        ph = [{"Q": ("r", 40 * Np), "A_U": ("w", 40 * Nfq)},
              {"Q": ("r", 40 * Np), "A_U": ("r", 40 * Nfq), "A_U(nbr)": ("g", 40 * Nfq), "rhs": ("w", 40 * Np),
               "geometry+mapP": ("r", 36 * 8 + 8 * Nfq + ((12 * Np + 8 * Nfq) if hex_delta else 0))}]
    elif formulation == "cns":
        ph = [{"Q": ("r", 32 * Np), "A_U": ("w", 32 * Nfq)},
              {"Q": ("r", 32 * Np), "A_U(nbr)": ("g", 32 * Nfq), "SG": ("w", 24 * Np), "B": ("w", 24 * Nfq),
               "geometry+mapP": ("r", 17 * 8 + 4 * Nfq + 8 * Nfq)},
              {"Q": ("r", 32 * Np), "A_U": ("r", 32 * Nfq), "A_U(nbr)": ("g", 32 * Nfq), "SG": ("r", 24 * Np), "B": ("r", 24 * Nfq),
               "B(nbr)": ("g", 24 * Nfq), "rhs": ("w", 32 * Np), "geometry+mapP": ("r", 17 * 8 + 4 * Nfq + 8 * Nfq)}]
    else:
        ph = [{"Q": ("r", 32 * Np), "A_U": ("w", 32 * Nfq)},
              {"Q": ("r", 32 * Np), "A_U": ("r", 32 * Nfq), "A_U(nbr)": ("g", 32 * Nfq), "rhs": ("w", 32 * Np),
               "geometry+mapP": ("r", 17 * 8 + 4 * Nfq + 8 * Nfq)}]
    fam = {}
    per_phase = []
    for d in ph:
        r = sum(b for k, (m, b) in d.items() if m in "rg")
        w = sum(b for k, (m, b) in d.items() if m == "w")
        g = sum(b for k, (m, b) in d.items() if m == "g")
        per_phase.append({"read": r, "write": w, "gathered": g, "with_overfetch": r + w + (GATHER_OVERFETCH - 1.0) * g})
        for k, (m, b) in d.items():
            name = k.replace("(nbr)", "")
            e = fam.setdefault(name, {"read": 0.0, "write": 0.0, "with_overfetch": 0.0})
            e["write" if m == "w" else "read"] += b
            e["with_overfetch"] += b * (GATHER_OVERFETCH if m == "g" else 1.0)
    return fam, per_phase


def kernel_source_hash():
    """sha256 over the kernel sources: the PMC-derived fields of profiles/pmc_traffic.json carry the hash of the sources
    they were measured on and are dropped (null + pmc_stale) when the kernels have changed since."""
    h = hashlib.sha256()
    d = os.path.join(ROOT, "esdg_cns_amd", "csrc")
    for f in sorted(os.listdir(d)):
        if f.endswith(".hpp") or (f.startswith("esdg_kernels") and f.endswith(".hip")):   # device code only (not the host API)
            h.update(open(os.path.join(d, f), "rb").read())
    return h.hexdigest()[:16]


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--N", type=int, default=None, help="degree (default 4; 3 for hex)")
    ap.add_argument("--kx", type=int, default=None, help="elements in x (default 512 on one GPU, 2048 on several; 128 for hex, also y)")
    ap.add_argument("--ky-per-gpu", type=int, default=None, help="element rows per GPU (default 512 on one GPU, 256 on several)")
    ap.add_argument("--kz-per-gpu", type=int, default=16, help="element layers per GPU (weak scaling, hex)")
    ap.add_argument("--lf", type=float, default=0.0, help="hex: LF factor (the reference has 0*.25)")
    ap.add_argument("--hex-curve", type=float, default=0.0, help="hex: amplitude a of the script's curved mapping (0 = affine)")
    ap.add_argument("--hex-geometry", choices=["per-node", "element"], default="per-node",
                    help="hex: per-node = the metric arrays at all hybrid nodes, as the reference script holds them (the library "
                         "then reproduces its per-node use: geometry mode 2 of kh_rhs); element = one row per element (mode 0)")
    ap.add_argument("--formulation", choices=["cns", "euler", "hex"], default="cns")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-rough-state", action="store_true",
                    help="skip the extra timings -- the state without smooth regions (ms_per_step_rough_state) and the RK stages "
                         "(lsrk_*): the rocprofv3 passes use it so that their per-kernel averages and counters cover the headline "
                         "evaluation only")
    ap.add_argument("--state", choices=["vortex", "rough"], default="vortex",
                    help="2D, one GPU: rough = the whole measurement on the state without smooth regions (SURVEY 8d's robustness variant; "
                         "profiling aid: the headline is the vortex state BASELINE specifies, config.workload says which one ran)")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend of the process group (nccl = RCCL; gloo only to "
                    "rehearse the multi-rank logic on fewer GPUs: traces are then staged through the host)")
    ap.add_argument("--transport", choices=["rccl", "torch"], default=None,
                    help="halo transport: rccl = the library's own communicator and schedule (default with --backend nccl), "
                         "torch = torch.distributed P2P driven from Python (default with --backend gloo)")
    ap.add_argument("--oversubscribe", action="store_true", help="rehearsal: allow more ranks than GPUs (recorded in the JSON)")
    ap.add_argument("--master-port", type=int, default=29577)
    ap.add_argument("--require-rccl", action="store_true",
                    help="N > 1: exit 4 (after printing the JSON line) unless the library's RCCL transport ran on all N ranks")
    ap.add_argument("--no-weak-base", action="store_true",
                    help="N > 1: skip the single-GPU run of the same per-rank shard (config.weak_scaling_base_ms)")
    return ap.parse_args(argv)


def attach_with_agreement(attach, destroy, prove, dist, flag_device, rank, watchdog_s=300.0, _exit=os._exit):
    """N > 1: attach the library's RCCL transport on every rank and agree on the outcome.  Returns (transport, rccl_ranks, note).
    Stage 1: `attach()` (the id hand-off and esdg_comm_init: ncclCommInitRank + plan cross-check with the neighbours) on every
    rank, then ONE agreement over the bootstrap group before any evaluation is posted.  A failure up to here has left no send or
    receive pending, so every rank falls back to the torch.distributed transport together (the JSON line says so:
    config.transport / config.transport_note; --require-rccl turns that into exit code 4).
    Stage 2: `prove()` (two evaluations).  A rank that fails there may leave its peers inside a posted ncclRecv, which no
    collective agreement can reach any more: it prints the reason and exits with code 5 (no fallback, no re-exec); its peers are
    ended by the watchdog, which bounds both stages (exit code 6), so that a hung communicator ends the run instead of the round.
    (`_exit` is the process exit used by stage 2 and the watchdog; the CPU test of this function replaces nothing else.)"""
    import threading
    import torch

    def _hung():
        print(f"bench.py: rank {rank}: RCCL attach / proving evaluations did not finish within {watchdog_s:g} s -- giving up", file=sys.stderr, flush=True)
        _exit(6)
    dog = threading.Timer(watchdog_s, _hung)
    dog.daemon = True
    dog.start()
    err, rccl_ranks = "", 0
    try:
        rccl_ranks = attach()
    except Exception as e:  # noqa: BLE001
        err = f"{type(e).__name__}: {e}"
    flag = torch.tensor([1.0 if err else 0.0], device=flag_device)
    dist.all_reduce(flag, op=dist.ReduceOp.MAX)
    if float(flag.item()) > 0:
        try:
            destroy()
        except Exception:  # noqa: BLE001
            pass
        dog.cancel()
        note = "library RCCL transport could not be attached on a rank (" + (err or "another rank") + "); fell back to torch.distributed P2P"
        if rank == 0:
            print("bench.py: " + note, file=sys.stderr)
        return "torch", 0, note
    try:
        prove()
    except Exception as e:  # noqa: BLE001
        print(f"bench.py: rank {rank}: evaluation over the library's RCCL transport failed: {type(e).__name__}: {e}", file=sys.stderr, flush=True)
        _exit(5)
    dog.cancel()
    return "rccl", rccl_ranks, None


def exit_code(world, require_rccl, transport, rccl_ranks):
    """Process exit code after the JSON line: 4 when --require-rccl was given and the library's RCCL transport did not run on
    all ranks, else 0."""
    return 4 if (world > 1 and require_rccl and (transport != "rccl" or rccl_ranks != world)) else 0


def launch_ranks(args):
    """`python bench.py --gpus N` as typed: start the N ranks as a torch.distributed.run child (fresh processes; this one has
    not touched a GPU) and pass their output and exit code on."""
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(args.master_port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    return subprocess.call(cmd, env=env)


def run(args):
    import torch
    import torch.distributed as dist
    from esdg_cns_amd import engine

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    ndev = torch.cuda.device_count()                        # does not initialise the GPU
    if world != args.gpus:
        if rank == 0:
            print(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}", file=sys.stderr)
        return 2
    oversub = world > max(ndev, 1)
    if oversub and not args.oversubscribe:
        if rank == 0:
            print(f"bench.py: {world} ranks but {ndev} visible GPU(s); pass --oversubscribe for a rehearsal", file=sys.stderr)
        return 3
    if oversub and args.backend == "nccl":
        if rank == 0:
            print("bench.py: RCCL needs one GPU per rank; use --backend gloo for an oversubscribed rehearsal", file=sys.stderr)
        return 3
    dev = local_rank % max(ndev, 1)
    torch.cuda.set_device(dev)
    transport = args.transport or ("rccl" if args.backend == "nccl" else "torch")
    if world > 1:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", dev))
        else:
            dist.init_process_group(args.backend)

    hexw = args.formulation == "hex"
    N = args.N if args.N is not None else (3 if hexw else 4)
    Kx = args.kx if args.kx is not None else (128 if hexw else (512 if world == 1 else 2048))
    kyr = args.ky_per_gpu if args.ky_per_gpu is not None else (512 if world == 1 else 256)
    if hexw:
        Kz_total = args.kz_per_gpu * world
        rank_offsets = np.array([Kx * Kx * args.kz_per_gpu * r for r in range(world + 1)], dtype=np.int64)   # z-slabs
        e0, e1 = int(rank_offsets[rank]), int(rank_offsets[rank + 1])
        rd, md, ops, Q = build_hex_problem(N, Kx, Kx, Kz_total, e0, e1, args.hex_curve, args.hex_geometry == "per-node")
        eng = engine.RhsEngine(rd, md, ops, engine.EULER_HEX_COLLOCATED, lf_scale=args.lf, rank=rank, nranks=world,
                               rank_offsets=rank_offsets)
        K_total = Kx * Kx * Kz_total
    else:
        Ky_total = kyr * world
        rank_offsets = np.array([Kx * kyr * r for r in range(world + 1)], dtype=np.int64)   # elements are numbered x-fastest
        e0, e1 = int(rank_offsets[rank]), int(rank_offsets[rank + 1])
        rd, md, ops, Q = build_problem(N, Kx, Ky_total, e0, e1, args.formulation)
        if args.state == "rough" and world == 1:
            Q = rough_state(Q)
        form = engine.CNS_MODAL if args.formulation == "cns" else engine.EULER_COLLOCATED
        eng = engine.RhsEngine(rd, md, ops, form, rank=rank, nranks=world, rank_offsets=rank_offsets)
        K_total = Kx * Ky_total
    rccl_ranks = 0
    Qd = eng.upload(Q)
    out = eng.new_state()
    transport_note = None
    if world > 1 and transport == "rccl":
        def _prove():
            for _ in range(2):
                eng.rhs_into(Qd, out)
            torch.cuda.synchronize()

        def _destroy():
            eng.L.esdg_comm_destroy(eng.ctx)
        flag_dev = torch.device("cuda", dev) if args.backend == "nccl" else "cpu"
        transport, rccl_ranks, transport_note = attach_with_agreement(eng.attach_rccl, _destroy, _prove, dist, flag_dev, rank)
        if transport != "rccl":
            eng.transport = "torch"
    Np, K_local, nfld = eng.Np, eng.K, eng.nfld

    def sync_all():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    # Clock ramp: an idle MI355X needs some tens of milliseconds of load before it reaches its sustained clocks (measured:
    # the first ~50 evaluations run up to 25 % slower).  A fixed number of untimed evaluations (the same on every rank)
    # precedes the W warm-up steps so that the timed region measures sustained throughput whatever W the caller picks;
    # reported as config.prewarm_evals.
    for _ in range(PREWARM_EVALS):
        eng.rhs_into(Qd, out)
    sync_all()
    for _ in range(args.warmup):
        eng.rhs_into(Qd, out)
    reps = []
    for _ in range(REPS):                                    # reps[0] is the contract's timed region of exactly K steps
        sync_all()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            eng.rhs_into(Qd, out)
        sync_all()
        el = time.perf_counter() - t0
        if world > 1:
            t = torch.tensor([el], dtype=torch.float64, device="cuda" if args.backend == "nccl" else "cpu")
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            el = float(t.item())
        reps.append(el)
    elapsed = reps[0]
    ms_per_step = elapsed / args.steps * 1e3
    evals_per_s = args.steps / elapsed
    value = K_total * Np * evals_per_s

    # --- per-phase kernel durations, live, with events on the launch stream (one-launch phases, no exchange) ---------
    import ctypes as C
    nph = eng.nphases
    nrep = 20
    stream = torch.cuda.current_stream()
    # (round 4: `nrep` back-to-back launches of ONE phase between two events, phase by phase, after a complete evaluation has
    # filled the trace buffers -- an event between every two kernels of the normal sequence added its own gap to each of them,
    # and the phases then summed to 3 % more than ms_per_step)
    q, o = C.c_void_p(Qd.data_ptr()), C.c_void_p(out.data_ptr())
    eng.rhs_into(Qd, out)
    phase_ms = []
    for ph in range(nph):
        e0_, e1_ = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        for _ in range(3):
            engine.check(eng.L.esdg_rhs_phase(eng.ctx, ph, q, o, eng._stream()))
        e0_.record(stream)
        for _ in range(nrep):
            engine.check(eng.L.esdg_rhs_phase(eng.ctx, ph, q, o, eng._stream()))
        e1_.record(stream)
        torch.cuda.synchronize()
        phase_ms.append(float(e0_.elapsed_time(e1_)) / nrep)
    kdur_ms = phase_ms[-1]
    # The same evaluation on a state without smooth regions (2D, one GPU): kt3_rhs takes data-dependent short cuts where a whole
    # wave's densities agree to 1e-4 -- most of the vortex's far field -- with bit-identical results; this is the time without them.
    rough_ms = None
    if not hexw and world == 1 and not args.no_rough_state:
        Qr = eng.upload(rough_state(Q))
        for _ in range(20):
            eng.rhs_into(Qr, out)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            eng.rhs_into(Qr, out)
        torch.cuda.synchronize()
        rough_ms = (time.perf_counter() - t0) / args.steps * 1e3
        del Qr
        eng.rhs_into(Qd, out)
    # One low-storage RK stage (dg2D_euler_quad.jl:204-205: resQ = a resQ + dt rhsQ; Q += b resQ) on one GPU: as the library's
    # fused stage (esdg_rhs_lsrk: the state update inside the last phase, no rhs array) and as the evaluation followed by the
    # update kernel (esdg_rhs + esdg_lsrk_update) -- what a time loop pays per stage either way.  dt = 0 keeps the state fixed.
    lsrk_stage_ms = lsrk_stage_unfused_ms = lsrk45_step_stage_ms = None
    if world == 1 and not args.no_rough_state:
        Qw, res = Qd.clone(), eng.new_state()
        res.zero_()

        def _time(fn, n=20, reps=3):
            # median of `reps` windows of n calls: one 12 ms window right after the clone above read 13 % high on one box
            for _ in range(5):
                fn()
            w = []
            for _ in range(reps):
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(n):
                    fn()
                torch.cuda.synchronize()
                w.append((time.perf_counter() - t0) / n * 1e3)
            return sorted(w)[reps // 2]
        lsrk_stage_ms = _time(lambda: eng.rhs_lsrk_fused(Qw, res, -0.4178904745, 0.1496590219993, 0.0))

        def _unfused():
            eng.rhs_into(Qw, out)
            eng.lsrk_update(Qw, res, out, -0.4178904745, 0.1496590219993, 0.0)
        lsrk_stage_unfused_ms = _time(_unfused)
        # ... and per stage of the library's whole five-stage step (esdg_lsrk45_step: five fused stages; the cross-stage fusion of
        # round 4 -- phase 0 of the next stage emitted by the last phase -- measured slower and was removed in round 5)
        import ctypes as C2
        if not hexw:
            qp, rp = C2.c_void_p(Qw.data_ptr()), C2.c_void_p(res.data_ptr())
            lsrk45_step_stage_ms = _time(lambda: engine.check(eng.L.esdg_lsrk45_step(eng.ctx, qp, rp, 0.0, eng._stream())), n=8) / 5.0
        del Qw, res
        eng.rhs_into(Qd, out)
    # One attempted DOPRI45 step of the CNS drivers' time loop (dg2D_CNS_cavity_optimized.jl:999-1037; six right-hand sides, the
    # stage combinations, the error norm) through esdg_dopri45_attempt -- with the combinations and the norm
    # inside the last phase of each stage (StageFuse) -- and from the library's building blocks (esdg_axpy_stages, esdg_rhs,
    # esdg_dopri_error: 41 state-sized sweeps beside the six evaluations).  Accepting swaps the two state buffers; includes the host's read of the estimate.
    dopri45_attempt_ms = dopri45_attempt_pieces_ms = None
    if world == 1 and not args.no_rough_state:    # (round 5: the fused attempt serves every formulation)
        from esdg_cns_amd import timestep
        for pieces in (False, True):
            Qw = Qd.clone()
            dp = timestep.Dopri45(eng, Qw, 1e-5, err_tol=1e-5, pieces=pieces, swap=True)   # (accept = buffer swap, as the C driver does)
            for _ in range(3):
                dp.step()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(10):
                dp.step()
            torch.cuda.synchronize()
            ms = (time.perf_counter() - t0) / 10 * 1e3
            if pieces: dopri45_attempt_pieces_ms = ms
            else: dopri45_attempt_ms = ms
            del dp, Qw
        eng.rhs_into(Qd, out)
    alg_bytes = 16.0 * nfld * Np * K_local       # read state once + write rhs once
    achieved = alg_bytes / (kdur_ms * 1e-3) / 1e9

    # --- PMC-derived figures (rocprofv3 passes of this command, tools/profile_round.sh -> profiles/pmc_traffic.json) ----
    traffic = whole_traffic = prof_us = valu_frac = whole_valu_frac = fp64_flops = insts_k = insts_whole = None
    pmc_stale = None
    rec = {}
    pmc = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    key = f"hex_N{N}_{Kx}x{Kx}x{args.kz_per_gpu}" if hexw else f"{args.formulation}_N{N}_{Kx}x{kyr}"
    if os.path.exists(pmc):
        try:
            rec = json.load(open(pmc)).get(key, {})
            pmc_stale = bool(rec) and rec.get("kernel_src_sha") != kernel_source_hash()
            if rec and not pmc_stale:
                traffic = rec.get("k_rhs_hbm_bytes_per_launch")
                prof_us = rec.get("k_rhs_rocprofv3_avg_us")
                whole_traffic = rec.get("whole_rhs_hbm_bytes")
                fp64_flops = rec.get("k_rhs_fp64_flops_per_launch")
                insts_k, insts_whole = rec.get("k_rhs_insts_valu_per_launch"), rec.get("whole_rhs_insts_valu")
                if fp64_flops:
                    valu_frac = fp64_flops / (kdur_ms * 1e-3) / (FP64_VALU_PEAK_TFLOPS * 1e12)
                if rec.get("whole_rhs_fp64_flops"):
                    whole_valu_frac = rec["whole_rhs_fp64_flops"] / (sum(phase_ms) * 1e-3) / (FP64_VALU_PEAK_TFLOPS * 1e12)
        except Exception:
            pass
    kname = "kh_rhs_l (last phase: surface flux + flux differencing + lift)" if hexw else \
        "kt3_rhs (last phase: flux differencing + viscous divergence + projection)"
    # --- which roof binds: computed, per formulation, from the numbers of this line ------------------------------------
    # HBM floor = bytes moved / the pool's practical copy rate; fp64 floor = counted fp64 flops / the measured vector peak.
    # Bytes: the PMC traffic of this command where the committed profile is current, else the design bytes per element
    # (what each phase reads and writes once, no cache reuse assumed).
    Nfq_ = eng.Nfq if hasattr(eng, "Nfq") else (6 * (N + 1) ** 2 if hexw else 4 * (N + 1))
    fam, per_phase = design_bytes(args.formulation, Np, Nfq_, hexw, hexw and args.hex_geometry == "per-node" and not args.hex_curve)
    design_k = per_phase[-1]["read"] + per_phase[-1]["write"]
    design_whole = sum(p_["read"] + p_["write"] for p_ in per_phase)
    # reads / writes of the dominant kernel and of the evaluation: PMC (x2-corrected FETCH_SIZE, WRITE_SIZE) where current
    kern = (rec.get("kernels") or {}) if (rec and not pmc_stale) else {}
    krec = next((v for k_, v in kern.items() if "_rhs<" in k_ or "_rhs_l<" in k_), None)
    if krec:
        rd_k, wr_k = krec["fetch_bytes_x2"], krec["write_bytes"]
        mains = [v for k_, v in kern.items() if any(t in k_ for t in ("project", "sigma", "_rhs"))]
        rd_w, wr_w = sum(v["fetch_bytes_x2"] for v in mains), sum(v["write_bytes"] for v in mains)
    else:
        rd_k, wr_k = per_phase[-1]["read"] * K_local, per_phase[-1]["write"] * K_local
        rd_w, wr_w = sum(p_["read"] for p_ in per_phase) * K_local, sum(p_["write"] for p_ in per_phase) * K_local
    bytes_k, bytes_whole = rd_k + wr_k, rd_w + wr_w
    rate_k, rate_whole = mix_rate_gbs(rd_k, wr_k), mix_rate_gbs(rd_w, wr_w)
    whole_flops = rec.get("whole_rhs_fp64_flops") if (rec and not pmc_stale) else None
    # compute-only time (every global address folded into an L2-resident window, same instruction stream: tools/compute_only.sh ->
    # profiles/compute_only.json; valid for the kernel sources it was measured on)
    compute_only = None
    cof = os.path.join(ROOT, "profiles", "compute_only.json")
    if os.path.exists(cof):
        try:
            c_ = json.load(open(cof)).get(key, {})
            if c_.get("kernel_src_sha") == kernel_source_hash():
                compute_only = c_.get("phase_ms")
        except Exception:
            pass
    # compute-side floor: ALL vector instructions the kernel issues (SQ_INSTS_VALU: fp64 arithmetic, logs, reciprocals, selects,
    # integer and address work alike) at the measured issue cost of one wave-instruction per SIMD -- the fp64-flop floor alone
    # ignored a quarter of the instruction stream
    def _issue_ms(n):
        return None if not n else n * VALU_CYCLES_PER_INST / (N_SIMD * SCLK_HZ) * 1e3
    floors = {"hbm_kernel": bytes_k / (rate_k * 1e9) * 1e3,
              "hbm_kernel_at_copy_rate": bytes_k / (PRACTICAL_HBM_GBS * 1e9) * 1e3,
              "fp64_kernel": None if not fp64_flops else fp64_flops / (MEASURED_FP64_TFLOPS * 1e12) * 1e3,
              "valu_issue_kernel": _issue_ms(insts_k),
              "hbm_whole_rhs": bytes_whole / (rate_whole * 1e9) * 1e3,
              "hbm_whole_rhs_at_copy_rate": bytes_whole / (PRACTICAL_HBM_GBS * 1e9) * 1e3,
              "fp64_whole_rhs": None if not whole_flops else whole_flops / (MEASURED_FP64_TFLOPS * 1e12) * 1e3,
              "valu_issue_whole_rhs": _issue_ms(insts_whole),
              "bytes_basis": "pmc" if krec else "design",
              "mix_rate_gbs": {"kernel": rate_k, "whole_rhs": rate_whole, "read_share_kernel": rd_k / max(bytes_k, 1.0),
                               "read_share_whole_rhs": rd_w / max(bytes_whole, 1.0),
                               "calibration": {"read_only": READ_ONLY_HBM_GBS, "copy": PRACTICAL_HBM_GBS, "write_only": WRITE_ONLY_HBM_GBS}}}
    bytes_by_array = {
        "unit": "bytes per element and evaluation",
        "design": {k_: v for k_, v in sorted(fam.items())},
        "design_by_phase": per_phase,
        "design_total": design_whole, "design_total_with_gather_overfetch": sum(p_["with_overfetch"] for p_ in per_phase),
        "gather_overfetch": GATHER_OVERFETCH,
        # PMC can be split by kernel and direction, not by array: x2-corrected FETCH_SIZE and WRITE_SIZE per kernel
        "pmc_by_kernel": None if not kern else {k_.replace("esdg::", ""): {"read": v["fetch_bytes_x2"] / K_local, "write": v["write_bytes"] / K_local}
                                                 for k_, v in kern.items() if any(t in k_ for t in ("project", "sigma", "_rhs"))},
        "pmc_total": None if not kern else bytes_whole / K_local,
        "algorithmic": 16.0 * nfld * Np}

    def _bound(h, f):
        # no counted instruction stream for this build (stale or absent PMC file): no claim
        return None if f is None else ("hbm" if h >= f else "valu-issue")
    whole_alg_frac = (alg_bytes * evals_per_s / 1e9) / HBM_PEAK_GBS
    roofline = {
        # `bound`: the larger floor of the dominant kernel (see floors_ms: HBM floor = PMC bytes at the rate its own read / write mix
        # reaches on this pool; VALU-issue floor = all vector instructions at the measured issue cost); `frac` stays the HBM figure
        # the north star names (algorithmic bytes / kernel time / 8 TB/s), `valu_frac` the counted fp64 flops of the same kernel
        # against the fp64 vector peak; `whole_rhs_bound` the same comparison over all phases of one evaluation
        "bound": _bound(floors["hbm_kernel"], floors["valu_issue_kernel"]), "kernel": kname,
        "whole_rhs_bound": _bound(floors["hbm_whole_rhs"], floors["valu_issue_whole_rhs"]),
        # measured time over each floor (1 = at the floor)
        "kernel_over_floor": {"hbm": kdur_ms / floors["hbm_kernel"],
                              "valu_issue": None if not floors["valu_issue_kernel"] else kdur_ms / floors["valu_issue_kernel"]},
        "floors_ms": floors, "practical_hbm_gbs": PRACTICAL_HBM_GBS, "measured_fp64_tflops": MEASURED_FP64_TFLOPS,
        "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
        "traffic": traffic,
        "traffic_note": "FETCH_SIZE x2 + WRITE_SIZE per launch (rocprofv3 --pmc, separate passes): requests that reach the fabric, "
                        "Infinity-Cache hits included -- an upper bound on HBM bytes, not HBM bytes",
        "kernel_ms": kdur_ms, "phase_ms": phase_ms,
        # every phase with the memory system taken out (addresses folded into an L2-resident window; null when not measured on
        # these kernel sources): what is left of a phase when HBM costs nothing
        "compute_only_ms": None if not compute_only else compute_only[-1], "compute_only_phase_ms": compute_only,
        # the same kernel's average under rocprofv3 --kernel-trace --stats (committed profile of this command)
        "kernel_ms_rocprofv3": None if prof_us is None else prof_us / 1e3,
        "whole_rhs_frac": whole_alg_frac,
        # what the algorithm's counted fp64 work allows: algorithmic bytes / (counted fp64 flops at 100 % of the vector peak) /
        # 8 TB/s -- the north star's 0.40 lies above it, so valu_frac / bytes per element are the bars that can move
        "whole_rhs_fp64_ceiling_frac": None if not whole_flops else (alg_bytes / (whole_flops / (FP64_VALU_PEAK_TFLOPS * 1e12)) / 1e9) / HBM_PEAK_GBS,
        "north_star_target_frac": NORTH_STAR_TARGET_FRAC,
        "whole_rhs_traffic": whole_traffic,
        "bytes_by_array": bytes_by_array,
        "valu_peak_tflops": FP64_VALU_PEAK_TFLOPS, "fp64_flops_per_launch": fp64_flops, "valu_frac": valu_frac,
        "whole_rhs_valu_frac": whole_valu_frac, "pmc_stale": pmc_stale, "kernel_src_sha": kernel_source_hash()}

    # --- N > 1: the single-GPU time of the SAME per-rank shard (stand-alone periodic strip / slab, no exchange), so that a
    # scaling curve is read against its own weak-scaling base and not against the 512^2 point of --gpus 1 ------------------
    weak_base_ms = None
    if world > 1 and not args.no_weak_base:
        # every rank drops its engine -- and with it the library's RCCL communicator -- at the same point, then rank 0 runs the
        # base alone while the others wait at the barrier below (no rank keeps a communicator whose peer is gone)
        del eng
        torch.cuda.empty_cache()
        dist.barrier()
    if world > 1 and rank == 0 and not args.no_weak_base:
        try:
            if hexw:
                rd1, md1, ops1, Q1 = build_hex_problem(N, Kx, Kx, args.kz_per_gpu, 0, Kx * Kx * args.kz_per_gpu, args.hex_curve,
                                                          args.hex_geometry == "per-node")
                e1 = engine.RhsEngine(rd1, md1, ops1, engine.EULER_HEX_COLLOCATED, lf_scale=args.lf)
            else:
                rd1, md1, ops1, Q1 = build_problem(N, Kx, kyr, 0, Kx * kyr, args.formulation)
                e1 = engine.RhsEngine(rd1, md1, ops1, form)
            Q1d, o1 = e1.upload(Q1), e1.new_state()
            for _ in range(PREWARM_EVALS):
                e1.rhs_into(Q1d, o1)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(args.steps):
                e1.rhs_into(Q1d, o1)
            torch.cuda.synchronize()
            weak_base_ms = (time.perf_counter() - t0) / args.steps * 1e3
        except Exception as e:  # noqa: BLE001
            print(f"bench.py: weak-scaling base run failed: {type(e).__name__}: {e}", file=sys.stderr)

    if hexw:
        workload = (f"euler3d_hex_N{N}_{Kx}x{Kx}x{Kz_total}_periodic_box_lf{args.lf:g}" + (f"_curved{args.hex_curve:g}" if args.hex_curve else "")
                    + ("_pernode_geometry" if args.hex_geometry == "per-node" else "_element_geometry"))
        metric = f"element-DOF updates/sec (RHS evals/s) at N={N}, 3D hex Euler"
    else:
        workload = (f"{args.formulation}2d_N{N}_{Kx}x{Ky_total}_quads_periodic_{'vortex' if args.state == 'vortex' or world > 1 else 'rough_state'}"
                    + ("_Re1000_inviscid+viscous_dissipation" if args.formulation == "cns" else ""))
        metric = f"element-DOF updates/sec (RHS evals/s) at N={N}, 2D {'CNS' if args.formulation == 'cns' else 'Euler'} quad mesh"
    per = sorted(r / args.steps * 1e3 for r in reps)
    result = {
        "metric": metric, "value": value, "unit": "DOF updates/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f64", "data": "synthetic",
        "config": {"workload": workload, "elements": K_total, "elements_per_gpu": K_local, "Np": Np, "nfields": nfld,
                   "parallelism": f"element-{'slabs' if hexw else 'strips'} x{world}", "prewarm_evals": PREWARM_EVALS,
                   "backend": args.backend if world > 1 else None, "transport": transport if world > 1 else None, "transport_note": transport_note,
                   "rccl_ranks": rccl_ranks, "visible_gpus": ndev, "oversubscribed": oversub,
                   "weak_scaling_base_ms": weak_base_ms,
                   "hex_geometry_mode": (None if not hexw else ("curved per-node arrays (1)" if args.hex_curve else
                                         ("per-node 10-bit differences (2)" if args.hex_geometry == "per-node" else "element record (0)")))},
        "rhs_evals_per_s": evals_per_s, "elements_per_s": K_total * evals_per_s,
        "ms_per_step_rough_state": rough_ms,
        "lsrk_stage_ms": lsrk_stage_ms, "lsrk_stage_unfused_ms": lsrk_stage_unfused_ms, "lsrk45_step_stage_ms": lsrk45_step_stage_ms,
        "dopri45_attempt_ms": dopri45_attempt_ms, "dopri45_attempt_pieces_ms": dopri45_attempt_pieces_ms,
        "ms_per_step_median": per[len(per) // 2], "ms_per_step_min": per[0], "ms_per_step_reps": [r / args.steps * 1e3 for r in reps],
        "roofline": roofline,
    }
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        result["cpu_baseline"] = cpu_baseline(N, args.formulation, args.lf, rd, md, ops, Q, hexw)
    elif rank == 0:
        result["cpu_baseline"] = None
    if rank == 0:
        print(json.dumps(result), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    return exit_code(world, args.require_rccl, transport, rccl_ranks)


def main():
    args = parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(launch_ranks(args))
    sys.exit(run(args))


if __name__ == "__main__":
    main()
