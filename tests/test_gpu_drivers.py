"""The steps either side of the hot path on the GPU: the transliterated Euler-quad driver
(LSRK45, vortex L2 error, invariant (5): order N+1 convergence) and the DOPRI45 loop of the CNS drivers against
the same loop driven by the oracle."""
import math
import os
import sys

import numpy as np
import pytest

from common import product_cns_problem, rel_l2

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "examples"))


def test_euler_quad_driver_converges_at_order_N_plus_1():
    import dg2D_euler_quad as drv
    N = 3
    e1, rt1 = drv.run(N=N, K1D=9, T=0.5, verbose=False)
    e2, rt2 = drv.run(N=N, K1D=18, T=0.5, verbose=False)
    rate = math.log2(e1 / e2)
    print(f"vortex L2 error N={N}: K1D=9 {e1:.3e}, K1D=18 {e2:.3e}, rate {rate:.2f}")
    assert e2 < e1 and rate > N                       # dg2D_euler_quad.jl:218-233 error functional
    assert rt2 <= 1e-12                                     # LF penalty on: entropy dissipative (:186-191)


def test_dopri45_loop_matches_oracle_driven_loop(oracle_lib):
    from esdg_cns_amd import engine, setup_dg as sd, timestep
    from oracle import oracle as orc
    N, Kx, Ky = 2, 6, 6
    p = orc.build_cns_problem(N, Kx, Ky, bc="periodic")
    co = orc.CnsOracle(p)
    rka, rkE, _ = sd.dopri45_coeffs()
    dt0 = 0.5 * (2 / Kx) / ((N + 1) * (N + 2) / 2)
    # oracle-driven restatement of dg2D_CNS_cavity_optimized.jl:997-1037
    Q = [q.copy() for q in p.Q]
    k = [None] * 7
    k[0] = co.rhsRK(Q, False)[0]
    dt, prev, t, hist = dt0, 0.0, 0.0, []
    for i in range(6):
        for s in range(1, 7):
            Qt = [q + dt * sum(rka[s, j] * k[j][f] for j in range(s)) for f, q in enumerate(Q)]
            k[s] = co.rhsRK(Qt, False)[0]
        err = 0.0
        for f in range(4):
            e = sum(rkE[j] * k[j][f] for j in range(7))
            err += np.sum((np.abs(e) / (1e-5 * (1 + np.abs(Q[f])))) ** 2)
        err = math.sqrt(err / (Q[0].size * 4))
        if err < 1.0:
            Q, t, k[0] = Qt, t + dt, k[6]
        dtn = .8 * dt * (.9 / err) ** (.4 / 6)
        if i > 0:
            dtn *= (prev / max(1e-14, err)) ** (.3 / 6)
        dt, prev = max(min(10 * dt0, dtn), 1e-9), err
        hist.append((err, dt))
    rd, md, ops, Qp = product_cns_problem(N, Kx, Ky)
    eng = engine.RhsEngine(rd, md, ops, engine.CNS_MODAL, Re=p.Re, mu=p.mu, lam=p.lam, Pr=p.Pr)
    Qd = eng.upload(Qp)
    integ = timestep.Dopri45(eng, Qd, dt0)
    for i in range(6):
        ok, err = integ.step()
        assert abs(err - hist[i][0]) <= 1e-8 * max(1.0, hist[i][0]) and abs(integ.dt - hist[i][1]) <= 1e-9 * hist[i][1]
    # the step sizes follow from the error estimates (agreeing to ~1e-10 relative), so t and Q agree to that level
    assert abs(integ.t - t) < 1e-9 and rel_l2(eng.download(Qd), Q) <= 1e-9


def test_cavity_driver_runs_all_wall_types():
    import dg2D_CNS_quad as drv
    for bct in (1, 2, 3):
        Q, integ = drv.run("cavity", N=2, K1D=6, T=0.02, BCTYPE=bct, verbose=False)
        assert integ.t >= 0.02 and all(np.isfinite(q).all() for q in Q)
        assert np.abs(Q[1]).max() > 0 if bct != 3 else True      # the lid drags the fluid (no-slip types)


def test_hex_driver_rhstest_and_density_wave_convergence():
    """examples/dg3D_euler_hex.py: the script's own diagnostic (`@show rhstest`, dg3D_euler_hex.jl:224-226: entropy
    conservative with the LF term at factor 0) and its commented-out LSRK45 loop (:228-262) on an exact density wave."""
    import dg3D_euler_hex as drv
    rt = drv.run_rhstest(N=2, K1D=4, verbose=False)
    assert abs(rt) < 1e-11
    N = 2
    e1, _ = drv.run_wave(N=N, K1D=4, T=0.25, verbose=False)
    e2, rt2 = drv.run_wave(N=N, K1D=8, T=0.25, verbose=False)
    rate = math.log2(e1 / e2)
    print(f"hex density wave L2 error N={N}: K1D=4 {e1:.3e}, K1D=8 {e2:.3e}, rate {rate:.2f}; rhstest {rt2:.2e}")
    assert e2 < e1 and rate > N
    assert abs(rt2) < 1e-10


def test_lsrk45_hip_graph_replay_is_bitwise_equal_and_faster_on_cfg1():
    """BASELINE cfg1 (Euler N=3, 16x16): the LSRK45 loop replayed from one captured HIP graph per step gives the same
    bits as the stage-by-stage loop; wall time per step is reported."""
    import time
    import torch
    from common import product_euler_problem
    from esdg_cns_amd import engine, timestep
    rd, md, ops, Q = product_euler_problem(3, 16, 16)
    eng = engine.RhsEngine(rd, md, ops, engine.EULER_COLLOCATED)
    dt, nsteps = 0.025, 40
    Q1, Q2 = eng.upload(Q), eng.upload(Q)
    timestep.lsrk45_run(eng, Q1.clone(), dt, 2)          # warm-up
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    timestep.lsrk45_run(eng, Q1, dt, nsteps)
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    g = timestep.lsrk45_run_graph(eng, Q2, dt, 0)        # capture only
    torch.cuda.synchronize()
    # allocations and writes between capture and replay must not be able to alias the graph's residual buffer: the graph
    # object keeps it (and Q2) alive (g.esdg_buffers)
    assert g.esdg_buffers[0] is Q2
    junk = [torch.full_like(Q2, float("nan")) for _ in range(4)]
    junk.append(eng.rhs(Q1))
    torch.cuda.synchronize()
    t1b = time.perf_counter()
    for i in range(nsteps):
        g.replay()
        if i % 7 == 0:
            junk[i % 4] = torch.full_like(Q2, float(i))
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f"cfg1 LSRK45: {1e6 * (t1 - t0) / nsteps:.1f} us/step stage-by-stage, {1e6 * (t2 - t1b) / nsteps:.1f} us/step graph replay "
          f"(capture {1e3 * (t1b - t1):.1f} ms once)")
    assert torch.equal(Q1, Q2)


def test_shocktube_driver_tracks_the_exact_viscous_shock():
    """examples/dg2D_CNS_shocktube_quad.py (dg2D_CNS_modalESDG.jl on quads): the Becker travelling viscous shock is an
    exact Navier-Stokes solution, so the error against it at the final time must be small and fall under refinement --
    an end-to-end check of the CNS right-hand side, the BCTYPE 4 closures and the DOPRI45 loop that does not involve
    the oracle."""
    import dg2D_CNS_shocktube_quad as drv
    e1, i1, _ = drv.run(N=2, K1D=32, T=0.05, Ky=2, verbose=False)
    e2, i2, _ = drv.run(N=2, K1D=64, T=0.05, Ky=2, verbose=False)
    rate = math.log2(e1 / e2)
    print(f"Becker shock tube N=2, T=0.05: L2 error K1D=32 {e1:.3e}, K1D=64 {e2:.3e} (rate {rate:.2f}); Linf {i1:.2e} -> {i2:.2e}")
    assert rate > 2.0 and e2 < 5e-3     # measured: 1.76e-2 -> 2.51e-3 (rate 2.8); N=3: 4.1e-3 -> 2.0e-4 (rate 4.4)


def test_convergence_test_driver_boundary_error_matches_oracle_functional():
    """examples/dg2D_CNS_convergence_test.py (dg2D_CNS_convergence_test.jl on quads): cavity with the lid profile
    (1+cos(pi x))/2, DOPRI45, boundary-velocity error on the device == the oracle's functional of the final state."""
    import dg2D_CNS_convergence_test as drv
    from esdg_cns_amd import engine
    from oracle import oracle as orc
    from oracle import ref_errors as re
    N, K1D = 2, 6
    ex, wr, integ = drv.run_one(N, K1D, T=0.01, CFL=0.05)
    assert integ.t >= 0.01 and np.isfinite(ex) and 0 < ex < wr
    Q = engine.RhsEngine.download(integ.Q)
    p = orc.build_cns_problem(N, K1D, K1D, bc="cavity", BCTYPE=1)
    rex, rwr, _ = re.boundary_velocity_error(Q, p.rd, p.md, K1D, drv.vlid)
    print(f"convergence-test driver N={N} K1D={K1D}: err {ex:.6e} (oracle functional {rex:.6e}), all terms {wr:.6e}")
    assert abs(ex - rex) <= 1e-11 * rex and abs(wr - rwr) <= 1e-11 * rwr
    # the lid has started to drag the fluid: the state is no longer at rest
    assert max(np.abs(Q[1]).max(), np.abs(Q[2]).max()) > 1e-4


def test_pure_c_driver_on_the_c_abi_matches_python_path(tmp_path):
    """examples/c/dg2D_euler_quad.c: set-up (esdg_setup_*), engine and LSRK45 loop through the C ABI only, built with
    gcc against libesdg_hip.so and run as a child process; same numbers as the Python host path."""
    import subprocess
    import torch
    from esdg_cns_amd import engine, physics as ph, setup_dg as sd, timestep
    exe = str(tmp_path / "euler_quad_c")
    lib = os.path.join(ROOT, "esdg_cns_amd")
    subprocess.check_call(["gcc", "-O2", "-I", os.path.join(ROOT, "include"), os.path.join(ROOT, "examples", "c", "dg2D_euler_quad.c"),
                           "-o", exe, "-L", lib, "-lesdg_hip", "-lm", "-Wl,-rpath," + lib])
    N, K1D, T = 3, 9, 0.25
    out = subprocess.check_output([exe, str(N), str(K1D), str(T)], text=True)
    print(out.strip())
    vals = dict(kv.split("=") for kv in out.split() if "=" in kv)
    # the same run through the Python host mirror
    Kx, Ky = 4 * K1D // 3, K1D
    VX, VY, EToV = sd.uniform_quad_mesh(Kx, Ky)
    VX, VY = 15 * (1 + VX) / 2, 5 * VY
    rd = sd.init_reference_quad(N, sd.gauss_quad(0, 0, N))
    md = sd.init_mesh((VX, VY), EToV, rd)
    sd.make_periodic(md, rd)
    ops = sd.euler_quad_ops(rd)
    sd.interp_geofacs_to_hybrid(md, ops["Vh"])
    Q = ph.primitive_to_conservative(*ph.vortex(md.xq, md.yq, 0))
    CN = (N + 1) * (N + 2) / 2
    dt = 2.0 * (2 / K1D) / CN
    nsteps = int(np.ceil(T / dt))
    dt = T / nsteps
    eng = engine.RhsEngine(rd, md, ops, engine.EULER_COLLOCATED)
    Qd = eng.upload(Q)
    timestep.lsrk45_run(eng, Qd, dt, nsteps)
    Qn = eng.download(Qd)
    Qex = ph.primitive_to_conservative(*ph.vortex(md.xq, md.yq, T))
    err = np.sqrt(sum(np.sum(md.wJq * (a - b) ** 2) for a, b in zip(Qn, Qex)))
    integral = sum(np.sum(md.wJq * a) for a in Qn)
    assert int(vals["steps"]) == nsteps
    assert abs(float(vals["L2err_gauss"]) - err) <= 1e-9 * max(err, 1e-12) + 1e-13
    assert abs(float(vals["integral"]) - integral) <= 1e-11 * abs(integral)
    assert float(vals["rhstest"]) <= 1e-12


def test_pure_c_hex_driver(tmp_path):
    """examples/c/dg3D_euler_hex.c: the hex script on the C ABI alone (C set-up with the intended face-vertex sets,
    esdg_create_hex, one rhs, `@show rhstest`): entropy conservative to round-off."""
    import subprocess
    exe = str(tmp_path / "euler_hex_c")
    lib = os.path.join(ROOT, "esdg_cns_amd")
    subprocess.check_call(["gcc", "-O2", "-I", os.path.join(ROOT, "include"), os.path.join(ROOT, "examples", "c", "dg3D_euler_hex.c"),
                           "-o", exe, "-L", lib, "-lesdg_hip", "-lm", "-Wl,-rpath," + lib])
    out = subprocess.check_output([exe, "3", "4"], text=True)
    print(out.strip())
    vals = dict(kv.split("=") for kv in out.split() if "=" in kv)
    assert vals["fields"] == "5" and float(vals["max|rhs|"]) > 1e-3
    assert abs(float(vals["rhstest"])) < 1e-11


def test_whole_step_c_entry_points_match_the_python_loops():
    """esdg_lsrk45_step / esdg_dopri45_attempt / esdg_dopri45_next_dt against timestep.lsrk45_run / timestep.Dopri45 from the
    library's building blocks: identical bits and identical step-size histories.  (The attempt of an unsharded CNS context is
    the fused one -- stage combinations and error norm inside kt3_rhs -- so this is also fused against unfused.)"""
    import ctypes as C
    import torch
    from esdg_cns_amd import engine, timestep
    from esdg_cns_amd._lib import check
    rd, md, ops, Q = product_cns_problem(3, 6, 6)
    eng = engine.RhsEngine(rd, md, ops, engine.CNS_MODAL)
    L, ctx = eng.L, eng.ctx
    # LSRK45
    dt = 2e-3
    Q1, Q2 = eng.upload(Q), eng.upload(Q)
    timestep.lsrk45_run(eng, Q1, dt, 3)
    res = torch.zeros_like(Q2)
    for _ in range(3):
        check(L.esdg_lsrk45_step(ctx, C.c_void_p(Q2.data_ptr()), C.c_void_p(res.data_ptr()), dt, eng._stream()))
    assert torch.equal(Q1, Q2)
    # DOPRI45
    dt0 = 0.5 * (2 / 6) / 10
    Qa = eng.upload(Q)
    integ = timestep.Dopri45(eng, Qa, dt0, pieces=True)
    hist = []
    for _ in range(6):
        ok, err = integ.step()
        hist.append((ok, err, integ.dt))
    Qb = eng.upload(Q)
    k = [torch.zeros_like(Qb) for _ in range(7)]
    Qtmp = torch.empty_like(Qb)
    eng.rhs_into(Qb, k[0])
    dtc, prev, t = dt0, 0.0, 0.0
    for i in range(6):
        ptrs = (C.c_void_p * 7)(*[x.data_ptr() for x in k])
        e = C.c_double()
        check(L.esdg_dopri45_attempt(ctx, C.c_void_p(Qb.data_ptr()), C.c_void_p(Qtmp.data_ptr()), ptrs, dtc, 1e-5, C.byref(e), eng._stream()))
        ok = e.value < 1.0
        if ok:
            Qb.copy_(Qtmp)
            t += dtc
            k[0], k[6] = k[6], k[0]
        dtc = L.esdg_dopri45_next_dt(dtc, dt0, e.value, prev, i)
        prev = e.value
        assert ok == hist[i][0] and abs(e.value - hist[i][1]) <= 1e-12 * max(1.0, hist[i][1]) and abs(dtc - hist[i][2]) <= 1e-15 * dt0 + 1e-12 * dtc
    assert torch.equal(Qa, Qb) and abs(t - integ.t) < 1e-15


def test_pure_c_sharded_driver_over_the_library_rccl_transport(tmp_path):
    """examples/c/dg2D_CNS_sharded.c in its one-GPU mode: rank 0's strip of an 8-rank CNS mesh through esdg_comm_init
    (loopback) / esdg_rhs on the sharded context, no torch and no Python in the process; it compares the result with the
    same strip as a stand-alone periodic mesh and exits non-zero on a mismatch."""
    import subprocess
    exe = str(tmp_path / "cns_sharded_c")
    lib = os.path.join(ROOT, "esdg_cns_amd")
    subprocess.check_call(["gcc", "-O2", "-I", os.path.join(ROOT, "include"), os.path.join(ROOT, "examples", "c", "dg2D_CNS_sharded.c"),
                           "-o", exe, "-L", lib, "-lesdg_hip", "-lm", "-Wl,-rpath," + lib])
    for args in (["1", "4", "32", "4"], ["1", "3", "24", "2"]):
        out = subprocess.check_output([exe] + args, text=True, timeout=300)
        print(out.strip())
        assert out.count("OK") == 2 and "MISMATCH" not in out and "RCCL comm size 1" in out and "DOPRI45, 6 attempts" in out


@pytest.mark.parametrize("formulation", ["cns", "hex", "cavity"])
def test_two_gloo_ranks_sharing_the_gpu_match_the_single_engine(formulation):
    """The multi-rank product path end to end (RhsEngine + halo plan + overlapped schedule + HaloExchanger over
    torch.distributed, traces staged through the host with gloo): two fresh rank processes share the GPU and must
    reproduce the single engine -- tools/check_sharded.py exits 0 only for bitwise / round-off agreement of the RHS, the
    fused LSRK stage and the reduced error functional."""
    import socket
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(root, "tools", "check_sharded.py"), "--backend", "gloo", "--formulation", formulation]
    p = subprocess.run(cmd, cwd=root, env=env, capture_output=True, text=True, timeout=300)
    print(p.stdout[-1500:])
    assert p.returncode == 0, p.stderr[-3000:]
    assert "MISMATCH" not in p.stdout and "check_sharded" in p.stdout


@pytest.mark.parametrize("case", ["cns N=4 13x9", "cns N=2 10x7", "cns N=6 5x4", "cns N=7 4x3", "cavity N=4 9x8 BCTYPE=1", "cavity N=3 8x7 BCTYPE=2",
                                  "cavity N=1 7x6 BCTYPE=1", "cns N=4 256x256", "cns N=9 3x2", "euler N=4 12x9", "euler N=3 16x16", "euler N=7 4x3",
                                  "inviscid N=4 9x8", "cavity N=9 3x2 BCTYPE=1", "cavity N=10 2x2 BCTYPE=3", "cavity N=6 4x3 BCTYPE=1", "hex N=3 5x4x3", "hex N=1 6x5x4", "hex N=4 3x2x2", "hex N=7 2x2x1", "hex N=10 2x2x1", "hexcurved N=2 4x3x3"])
def test_fused_dopri45_attempt_is_bitwise_the_attempt_from_building_blocks(case):
    """esdg_dopri45_attempt on an unsharded 2D context (CNS; round 5: collocated Euler and the inviscid modal formulation too): the last phase of every stage also forms the next stage's state from the
    k_s it holds in registers, stage 6 leaves the error combination so far in k[6]'s array and stage 7 reduces the norm
    (StageFuse, kt3_rhs STG).  Claim: per node the same bits as esdg_axpy_stages + RHS + esdg_dopri_error
    (dg2D_CNS_cavity_optimized.jl:1002-1021) -- the stage state, all seven k, the accepted solution AND the error estimate (its
    terms are added in one order that depends on the number of entries alone); same accept / reject and step-size history.  Partial last groups, periodic and wall meshes, N1 = 2 ... 8.
    A context created with ESDG_DOPRI_FUSION=0 takes the unfused attempt inside the library: same bits again."""
    import torch
    from common import product_cavity_problem, product_euler_problem
    from esdg_cns_amd import engine, timestep
    kind, rest = case.split(" ", 1)
    N = int(rest.split()[0][2:]); Kx, Ky = (int(v) for v in rest.split()[1].split("x")[:2])
    kw, form = {}, engine.CNS_MODAL
    if kind.startswith("hex"):  # (round 5: kh_rhs_l's STG instantiation; affine and curved geometry modes)
        from common import product_hex_problem
        rd, md, ops, Q = product_hex_problem(N, Kx, Ky, int(rest.split()[1].split("x")[2]), a=0.1 if kind == "hexcurved" else 0.0)
        form, kw = engine.EULER_HEX_COLLOCATED, {"lf_scale": 0.25}
    elif kind == "cns":
        rd, md, ops, Q = product_cns_problem(N, Kx, Ky)
    elif kind == "euler":       # (round 5: the collocated Euler formulation and rhs_inviscid! alone take the fused attempt too)
        rd, md, ops, Q = product_euler_problem(N, Kx, Ky)
        form = engine.EULER_COLLOCATED
    elif kind == "inviscid":
        rd, md, ops, Q = product_cns_problem(N, Kx, Ky)
        form = engine.EULER_MODAL
    else:
        rd, md, ops, Q = product_cavity_problem(N, Kx, Ky)
        kw["BCTYPE"] = int(rest.split("=")[-1])
    eng = engine.RhsEngine(rd, md, ops, form, **kw)
    os.environ["ESDG_DOPRI_FUSION"] = "0"
    try:
        eng0 = engine.RhsEngine(rd, md, ops, form, ab_hooks=True, **kw)
    finally:
        del os.environ["ESDG_DOPRI_FUSION"]
    dt0 = 0.5 * (2 / Kx) / ((N + 1) * (N + 2) / 2) * (0.2 if kind.startswith("hex") else 1.0)
    integs = []
    for e, pieces in ((eng, False), (eng, True), (eng0, False)):     # (the first accepts by swapping its buffers, the others copy)
        integs.append(timestep.Dopri45(e, e.upload(Q), dt0, err_tol=1e-7 if case == "cns N=4 13x9" else 1e-5, pieces=pieces,
                                       swap=e is eng and not pieces))   # (tight: rejections first)
    fused, accepted = integs[0], 0
    for _ in range(2 if Kx >= 128 else 6):      # (cfg2's size: 10 923 workgroups, two attempts)
        # in lockstep, every attempt with the fused run's step size (the estimates are equal bit for bit since the norm has one
        # summation order; the lockstep keeps a failure of that claim from hiding the per-node comparisons behind it)
        outs, dt, prev = [], fused.dt, fused.prev_err
        for integ in integs:
            integ.dt, integ.prev_err = dt, prev
            outs.append(integ.step())
        torch.cuda.synchronize()
        accepted += bool(outs[0][0])
        assert torch.isfinite(fused.Q).all()
        for other, (ok, err) in zip(integs[1:], outs[1:]):
            assert ok == outs[0][0] and err == outs[0][1], (case, err, outs[0][1])   # (one summation order for the norm: esdg_kernels.hip k_dopri_err)
            assert torch.equal(fused.Q, other.Q), case
            if not ok:                                   # (rejected: both still hold the candidate; accepted: the swapping one holds the old state there)
                assert torch.equal(fused.Qtmp, other.Qtmp), case
            for a, b in zip(fused.k, other.k):
                assert torch.equal(a, b), case
    assert accepted > 0
