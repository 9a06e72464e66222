"""GPU tests of everything around the kernels: size-independent properties at BASELINE's full size,
element-index sharding (two engines on one GPU exchanging their halo segments exactly as two ranks
would), run-to-run bitwise reproducibility, the generic (pair-list) kernels against the tensor kernels,
the host-array drop-in entry point, the entropy diagnostic and the LSRK45 step against the oracle."""
import ctypes as C
import os

import numpy as np
import pytest
import torch

from common import product_cns_problem, product_euler_problem, rel_l2, steep_state

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def E():
    from esdg_cns_amd import engine
    return engine


def _rhs(eng, Q):
    return eng.download(eng.rhs(eng.upload(Q)))


def test_uses_tensor_kernels_and_generic_fallback_agree(E):
    rd, md, ops, Q = product_cns_problem(4, 8, 6)
    Qs = steep_state(md.x, md.y)
    fast = E.RhsEngine(rd, md, ops, E.CNS_MODAL)
    assert fast.L.esdg_uses_tensor_kernels(fast.ctx) == 1
    os.environ["ESDG_FORCE_GENERIC"] = "1"
    try:
        gen = E.RhsEngine(rd, md, ops, E.CNS_MODAL, ab_hooks=True)   # (the A/B build reads the environment switches)
    finally:
        del os.environ["ESDG_FORCE_GENERIC"]
    assert gen.L.esdg_uses_tensor_kernels(gen.ctx) == 0
    a, b = _rhs(fast, Qs), _rhs(gen, Qs)
    assert rel_l2(a, b) <= 1e-12          # two independent GPU implementations of the same path


@pytest.mark.parametrize("N,Kx", [(4, 96), (5, 48), (6, 40), (3, 64)])
def test_bitwise_reproducible(E, N, Kx):
    """Run-to-run determinism.  The ds_add_f64 accumulation order is fixed within a wave; elements whose lanes straddle
    two waves of a group (N=4: one of five, N=5,6: three of seven / five) accumulate into one copy per wave, summed in a
    fixed order, so wave scheduling cannot change a bit."""
    rd, md, ops, Q = product_cns_problem(N, Kx, Kx)
    eng = E.RhsEngine(rd, md, ops, E.CNS_MODAL)
    Qd = eng.upload(Q)
    r1 = eng.rhs(Qd).clone()
    for _ in range(5):
        r2 = eng.rhs(Qd)
        assert torch.equal(r1, r2)


@pytest.mark.parametrize("form", ["euler", "cns"])
def test_two_shards_on_one_gpu_match_unsharded(E, form):
    N, Kx, Ky = 3, 6, 8
    build = product_euler_problem if form == "euler" else product_cns_problem
    code = E.EULER_COLLOCATED if form == "euler" else E.CNS_MODAL
    rd, md, ops, Q = build(N, Kx, Ky)
    ref = _rhs(E.RhsEngine(rd, md, ops, code), Q)
    K = Kx * Ky
    offsets = np.array([0, 3 * Kx, K], dtype=np.int64)
    engs, Qd, out = [], [], []
    for r in range(2):
        rdr, mdr, opsr, Qr = build(N, Kx, Ky, elem_range=(int(offsets[r]), int(offsets[r + 1])))
        e = E.RhsEngine(rdr, mdr, opsr, code, rank=r, nranks=2, rank_offsets=offsets)
        assert e.halo is not None
        engs.append(e)
        Qd.append(e.upload(Qr))
        out.append(e.new_state())
    nph = engs[0].nphases
    for ph in range(nph):
        for r in range(2):
            E.check(engs[r].L.esdg_rhs_phase(engs[r].ctx, ph, C.c_void_p(Qd[r].data_ptr()), C.c_void_p(out[r].data_ptr()), None))
        torch.cuda.synchronize()
        # deliver the segments of every exchange produced by this phase (what isend/irecv would do)
        for x, (after, before, nc) in enumerate(engs[0].xinfo):
            if after != ph:
                continue
            for r in range(2):
                for (peer, so, sb, ro, rb) in engs[r].halo.segments[x]:
                    back = [s for s in engs[peer].halo.segments[x] if s[0] == r][0]
                    assert sb == back[4]
                    engs[peer].ws[back[3]:back[3] + back[4]] = engs[r].ws[so:so + sb]
        torch.cuda.synchronize()
    got = [np.concatenate([E.RhsEngine.download(out[0])[f], E.RhsEngine.download(out[1])[f]], axis=1) for f in range(4)]
    assert all(np.array_equal(a, b) for a, b in zip(got, ref))      # sharding must not change a single bit


def test_full_size_properties_cns_512(E):
    """BASELINE config 3 (N=4, 512x512): free stream, conservation, entropy inequality."""
    rd, md, ops, Q = product_cns_problem(4, 512, 512)
    eng = E.RhsEngine(rd, md, ops, E.CNS_MODAL, viscous_dissp=False)
    one = np.ones_like(Q[0])
    from esdg_cns_amd import physics as ph
    Qc = [np.asfortranarray(q) for q in ph.primitive_to_conservative(1.1 * one, .3 * one, -.2 * one, .9 * one)]
    r = eng.rhs(eng.upload(Qc))
    # round-off of O(1) fluxes is amplified by 1/J = 4/(hx*hy) ~ 7e3 on this mesh
    assert float(r.abs().max()) < 1e-11 / float(md.J.min())             # free-stream preservation
    # conservation on a de-correlated state: in the exactly uniform far field of the vortex every element commits the
    # SAME round-off (the free-stream residual above), which adds up coherently over 262144 elements
    from common import perturb
    Qp = perturb(Q)
    rh = eng.download(eng.rhs(eng.upload(Qp)))
    wJ = md.wJq
    scale = max(float(np.abs(wJ * (rd.Vq @ x)).sum()) for x in rh)
    for x in rh:                                                     # discrete conservation (no penalty, quirk Q3)
        assert abs(float((wJ * (rd.Vq @ x)).sum())) <= 2e-9 * max(scale, 1.0)
    Qd = eng.upload(Q)
    r = eng.rhs(Qd)
    assert eng.rhstest(Qd, r) < 0                                    # LF dissipation (this engine has viscous_dissp=False): entropy decays
    ec = E.RhsEngine(rd, md, ops, E.EULER_MODAL, inviscid_dissp=False)
    r2 = ec.rhs(Qd)
    assert abs(ec.rhstest(Qd, r2)) < 1e-9                            # entropy conservative without LF


@pytest.mark.parametrize("form,N,Kx,Ky", [("cns", 4, 9, 7), ("cns", 3, 7, 6), ("cns", 5, 5, 5), ("cns", 2, 8, 5), ("cns", 1, 6, 7),
                                          ("cns", 6, 4, 5), ("euler", 4, 9, 7), ("euler", 6, 4, 4), ("euler", 7, 3, 4), ("euler", 1, 5, 5),
                                          ("cns", 8, 4, 3), ("cns", 9, 3, 3), ("euler", 9, 3, 4)])
def test_ranged_launches_match_the_full_launch(E, form, N, Kx, Ky):
    """esdg_rhs_phase_range (what the halo-overlap schedule is built from): every phase run piecewise over an uneven
    partition of the elements, pieces in arbitrary order, must equal the one-launch evaluation BIT FOR BIT.  Pieces start
    at arbitrary elements, so elements land in different lanes / groups than in the full launch: the kernels' arithmetic
    must not depend on the slot (per-node tables; every accumulator cell receives at most two adds)."""
    build = product_euler_problem if form == "euler" else product_cns_problem
    code = E.EULER_COLLOCATED if form == "euler" else E.CNS_MODAL
    rd, md, ops, Q = build(N, Kx, Ky)
    eng = E.RhsEngine(rd, md, ops, code)
    Qd = eng.upload(Q)
    ref = eng.rhs(Qd).clone()
    K = Kx * Ky
    rng = np.random.default_rng(K + N)
    cuts = sorted(set([0, K] + [int(c) for c in rng.integers(1, K, size=4)]))
    pieces = [(a, b - a) for a, b in zip(cuts[:-1], cuts[1:])]
    out = torch.full_like(Qd, float("nan"))
    for ph in range(eng.nphases):
        for i in rng.permutation(len(pieces)):
            e0, n = pieces[i]
            E.check(eng.L.esdg_rhs_phase_range(eng.ctx, ph, e0, n, C.c_void_p(Qd.data_ptr()), C.c_void_p(out.data_ptr()), None))
    torch.cuda.synchronize()
    assert torch.isfinite(out).all()
    rel = float((out - ref).abs().max() / ref.abs().max())
    print(f"ranged {form} N={N}: pieces {pieces}, max rel diff {rel:.2e}")
    assert torch.equal(out, ref), rel
    with pytest.raises(Exception):                        # a range past the mesh is refused
        E.check(eng.L.esdg_rhs_phase_range(eng.ctx, 0, K - 1, 2, C.c_void_p(Qd.data_ptr()), C.c_void_p(out.data_ptr()), None))


@pytest.mark.parametrize("N,Kx,Ky,BCTYPE", [(4, 12, 9, 1), (3, 10, 7, 3), (2, 9, 8, 2)])
def test_ranged_launches_match_the_full_launch_on_wall_meshes(E, N, Kx, Ky, BCTYPE):
    """The same on the lid-driven cavity: the wall instantiations repeat the viscous operators of the elements with a boundary
    node in the nodal basis (MeshDev::wgeo) -- a decision taken per ELEMENT inside a branch taken per GROUP, so the result of an
    element must not depend on which other elements share its group."""
    from common import product_cavity_problem
    rd, md, ops, Q = product_cavity_problem(N, Kx, Ky)
    eng = E.RhsEngine(rd, md, ops, E.CNS_MODAL, BCTYPE=BCTYPE)
    Qd = eng.upload(Q)
    ref = eng.rhs(Qd).clone()
    K = Kx * Ky
    rng = np.random.default_rng(K + N)
    cuts = sorted(set([0, K] + [int(c) for c in rng.integers(1, K, size=5)]))
    pieces = [(a, b - a) for a, b in zip(cuts[:-1], cuts[1:])]
    out = torch.full_like(Qd, float("nan"))
    for ph in range(eng.nphases):
        for i in rng.permutation(len(pieces)):
            e0, n = pieces[i]
            E.check(eng.L.esdg_rhs_phase_range(eng.ctx, ph, e0, n, C.c_void_p(Qd.data_ptr()), C.c_void_p(out.data_ptr()), None))
    torch.cuda.synchronize()
    rel = float((out - ref).abs().max() / ref.abs().max())
    print(f"ranged cavity N={N} BCTYPE={BCTYPE}: pieces {pieces}, max rel diff {rel:.2e}")
    assert torch.equal(out, ref), rel


def test_trailing_idle_waves_read_no_geometry_past_the_mesh(E):
    """N=2 at 512x512: 262144 elements are not a multiple of the 20 elements a workgroup takes, so the last workgroup
    has waves without elements.  Their lanes used to form geometry addresses from an element index past the mesh
    (a memory access fault at this size; silent at small sizes).  Free stream and entropy decay must hold."""
    rd, md, ops, Q = product_cns_problem(2, 512, 512)
    eng = E.RhsEngine(rd, md, ops, E.CNS_MODAL)
    one = np.ones_like(Q[0])
    from esdg_cns_amd import physics as ph
    Qc = [np.asfortranarray(q) for q in ph.primitive_to_conservative(1.1 * one, .3 * one, -.2 * one, .9 * one)]
    r = eng.rhs(eng.upload(Qc))
    assert float(r.abs().max()) < 1e-11 / float(md.J.min())
    Qd = eng.upload(Q)
    assert eng.rhstest(Qd, eng.rhs(Qd)) < 0


def test_rhstest_matches_oracle(E, oracle_lib):
    from oracle import oracle as orc
    p = orc.build_euler_problem(3, 8, 8)
    _, rt = orc.EulerOracle(p).rhs(p.Q, .5, True)
    rd, md, ops, Q = product_euler_problem(3, 8, 8)
    eng = E.RhsEngine(rd, md, ops, E.EULER_COLLOCATED)
    Qd = eng.upload(Q)
    assert abs(eng.rhstest(Qd, eng.rhs(Qd)) - rt) < 1e-11


def test_host_dropin_entry_points(E, oracle_lib):
    from oracle import oracle as orc
    p = orc.build_euler_problem(2, 5, 4)
    Qs = steep_state(p.md.xq, p.md.yq)
    ref, _ = orc.EulerOracle(p).rhs(Qs)
    rd, md, ops, Q = product_euler_problem(2, 5, 4)
    eng = E.RhsEngine(rd, md, ops, E.EULER_COLLOCATED)
    assert rel_l2(eng.rhs_host(Qs), ref) <= 1e-11
    out, rt = E.rhs(tuple(Qs), md, ops, None, True, rd=rd)          # reference signature rhs(Q,md,ops,flux_fun,compute_rhstest)
    assert rel_l2(list(out), ref) <= 1e-11 and np.isfinite(rt)


def test_lsrk45_steps_match_oracle(E, oracle_lib):
    from esdg_cns_amd import setup_dg as sd
    from oracle import oracle as orc
    N, Kx, Ky = 3, 8, 8
    p = orc.build_euler_problem(N, Kx, Ky)
    eo = orc.EulerOracle(p)
    rk4a, rk4b, _ = sd.rk45_coeffs()
    dt = 2 * (2 / 8) / ((N + 1) * (N + 2) / 2) / 4
    Qo = [q.copy() for q in p.Q]
    res = [np.zeros_like(q) for q in Qo]
    nsteps = 3
    for _ in range(nsteps):                                           # dg2D_euler_quad.jl:198-207
        for k in range(5):
            r, _ = eo.rhs(Qo)
            res = [rk4a[k] * a + dt * b for a, b in zip(res, r)]
            Qo = [q + rk4b[k] * a for q, a in zip(Qo, res)]
    rd, md, ops, Q = product_euler_problem(N, Kx, Ky)
    eng = E.RhsEngine(rd, md, ops, E.EULER_COLLOCATED)
    Qd, resd, rhsd = eng.upload(Q), eng.new_state(), eng.new_state()
    for _ in range(nsteps):
        eng.lsrk45_step(Qd, resd, rhsd, dt, (rk4a, rk4b))
    assert rel_l2(eng.download(Qd), Qo) <= 1e-12


def test_fused_lsrk_stage_is_bitwise_the_unfused_one(E):
    from esdg_cns_amd import setup_dg as sd
    rd, md, ops, Q = product_cns_problem(4, 10, 9)
    eng = E.RhsEngine(rd, md, ops, E.CNS_MODAL)
    rk = sd.rk45_coeffs()
    dt = 1e-3
    Qa, ra, rhs = eng.upload(Q), eng.new_state(), eng.new_state()
    Qb, rb = eng.upload(Q), eng.new_state()
    for _ in range(2):
        eng.lsrk45_step(Qa, ra, rhs, dt, rk)
        eng.lsrk45_step_fused(Qb, rb, dt, rk)
    assert torch.equal(Qa, Qb) and torch.equal(ra, rb)


def test_check_state_reports_min_density_and_pressure(E):
    from common import product_hex_problem
    rd, md, ops, Q = product_cns_problem(3, 5, 4)
    eng = E.RhsEngine(rd, md, ops, E.CNS_MODAL)
    rho, p = Q[0], 0.4 * (Q[3] - .5 * (Q[1] ** 2 + Q[2] ** 2) / Q[0])
    mr, mp = eng.check_state(eng.upload(Q))
    assert abs(mr - rho.min()) < 1e-15 and abs(mp - p.min()) < 1e-14
    Qb = [q.copy() for q in Q]
    Qb[0][3, 2] = -0.5
    Qb[3][1, 7] = np.nan
    mr, mp = eng.check_state(eng.upload(Qb))
    assert mr == -0.5 and mp == -1e300
    rdh, mdh, opsh, Qh = product_hex_problem(2, 3, 2, 2)
    engh = E.RhsEngine(rdh, mdh, opsh, E.EULER_HEX_COLLOCATED)
    ph_ = 0.4 * (Qh[4] - .5 * (Qh[1] ** 2 + Qh[2] ** 2 + Qh[3] ** 2) / Qh[0])
    mr, mp = engh.check_state(engh.upload(Qh))
    assert abs(mr - Qh[0].min()) < 1e-15 and abs(mp - ph_.min()) < 1e-14


def test_full_size_properties_euler_256(E):
    """BASELINE config 2 (2D Euler, N=4, 256x256, collocated EC flux differencing) at its exact size: free stream,
    conservation, entropy conservation with the LF term off (|rhstest| ~ round-off), entropy decay with it on."""
    rd, md, ops, Q = product_euler_problem(4, 256, 256)
    eng = E.RhsEngine(rd, md, ops, E.EULER_COLLOCATED)
    one = np.ones_like(Q[0])
    from esdg_cns_amd import physics as ph
    from common import perturb
    Qc = [np.asfortranarray(q) for q in ph.primitive_to_conservative(1.1 * one, .3 * one, -.2 * one, .9 * one)]
    r = eng.rhs(eng.upload(Qc))
    assert float(r.abs().max()) < 1e-11 / float(md.J.min())             # free-stream preservation
    Qp = perturb(Q)
    rh = eng.download(eng.rhs(eng.upload(Qp)))
    wJ = md.wJq
    scale = max(float(np.abs(wJ * x).sum()) for x in rh)
    for x in rh:                                                          # discrete conservation
        assert abs(float((wJ * x).sum())) <= 2e-9 * max(scale, 1.0)
    Qd, Qpd = eng.upload(Q), eng.upload(Qp)
    rt = eng.rhstest(Qpd, eng.rhs(Qpd))
    assert rt < 0                                   # LF dissipation: entropy decays (on the smooth vortex at this
    r = eng.rhs(Qd)                                 # resolution the interface jumps, and with them the LF term, are ~1e-12)
    ec = E.RhsEngine(rd, md, ops, E.EULER_COLLOCATED, inviscid_dissp=False, lf_scale=0.0)
    r2 = ec.rhs(Qd)
    rt0 = ec.rhstest(Qd, r2)
    print(f"cfg2 euler N=4 256x256: rhstest with LF {rt:.3e}, without {rt0:.3e}")
    assert abs(rt0) < 1e-9                                                 # entropy conservative without LF
    assert torch.equal(eng.rhs(Qd), r)                                     # run-to-run bitwise


def _strip_periodic_state(x, y, LY):
    """Smooth state with period 15 in x and LY in y (one strip of the sharded box is then periodic on its own)."""
    from esdg_cns_amd import physics as ph
    cx, cy = 2 * np.pi * x / 15.0, 2 * np.pi * (y - y.min()) / LY
    rho = 1 + .2 * np.sin(cx + .3) * np.cos(cy + .1)
    u = .4 + .1 * np.cos(cx - .2) * np.sin(cy + .4)
    v = -.3 + .1 * np.sin(cx + .5) * np.sin(cy - .3)
    p = 1 + .15 * np.cos(cx + .7) * np.cos(cy + .2)
    return [np.asfortranarray(q) for q in ph.primitive_to_conservative(rho, u, v, p)]


def test_cfg4_rank0_strip_of_the_8_rank_mesh(E):
    """BASELINE config 4 (2D CNS, N=4, 2048x2048 over 8 GPUs): rank 0's shard -- the 2048x256 strip, 524 288 elements,
    with its ghost slots, send lists and pack kernels -- run on one GPU.  Its neighbours (ranks 1 and 7) are replaced by a
    local copy: for a state that is periodic over the strip, what rank 0 would receive from rank 1 is what it sends to
    rank 7 and vice versa (both sides order a segment by global node id, i.e. by x).  The result must equal, bit for bit,
    the same strip run as a stand-alone periodic mesh; free stream / conservation / entropy sign are checked on it."""
    from esdg_cns_amd import setup_dg as sd
    N, Kx, Kyr, nr = 4, 2048, 256, 8
    LY = 10.0 * Kyr / Kx                                     # strip height when the elements stay square (bench.py scaling)

    def build(Ky_total, e0, e1):
        VX, VY, EToV = sd.uniform_quad_mesh(Kx, Ky_total)
        VX = 15 * (1 + VX) / 2
        VY = 5 * VY * (Ky_total / Kx)
        rd = sd.init_reference_quad(N)
        md = sd.init_mesh((VX, VY), EToV, rd, elem_range=(e0, e1))
        sd.make_periodic(md, rd)
        md.mapB = np.zeros(0, dtype=np.int64)
        ops = sd.cns_ops(rd)
        sd.interp_geofacs_to_hybrid(md, ops["Vh"])
        return rd, md, ops

    rd, md, ops = build(Kyr * nr, 0, Kx * Kyr)               # rank 0 of the 8-rank mesh
    offsets = np.array([Kx * Kyr * r for r in range(nr + 1)], dtype=np.int64)
    sh = E.RhsEngine(rd, md, ops, E.CNS_MODAL, rank=0, nranks=nr, rank_offsets=offsets)
    assert sh.halo is not None and sh.K == 524288
    peers = sorted(s[0] for s in sh.halo.segments[0])
    assert peers == [1, 7]
    Q = _strip_periodic_state(md.x, md.y, LY)
    Qd, out = sh.upload(Q), sh.new_state()
    del Q
    for ph in range(sh.nphases):
        E.check(sh.L.esdg_rhs_phase(sh.ctx, ph, C.c_void_p(Qd.data_ptr()), C.c_void_p(out.data_ptr()), None))
        for x, (after, before, nc) in enumerate(sh.xinfo):
            if after != ph:
                continue
            seg = {s[0]: s for s in sh.halo.segments[x]}
            for a, b in ((1, 7), (7, 1)):                    # received from a := sent to b
                _, so, sb, _, _ = seg[b]
                _, _, _, ro, rb = seg[a]
                assert sb == rb and sb == Kx * (N + 1) * nc * 8
                sh.ws[ro:ro + rb] = sh.ws[so:so + sb]
    torch.cuda.synchronize()
    assert torch.isfinite(out).all()

    # the same strip as a stand-alone periodic mesh: the shard's own arrays (identical inputs) with the periodic strip's mapP
    _, md1s, _ = build(Kyr, 0, Kx * Kyr)
    import copy
    rd1, ops1, md1 = rd, ops, copy.copy(md)
    md1.mapP, md1.elem_offset, md1.Kglobal = md1s.mapP, 0, md.K
    del md1s
    one = E.RhsEngine(rd1, md1, ops1, E.CNS_MODAL)
    Q1, Qd1 = None, Qd
    ref = one.rhs(Qd1)
    assert torch.equal(out, ref)                             # the shard reproduces the stand-alone strip to the bit
    # properties at the shard's size
    rh = one.download(ref)
    wJ = md1.wJq
    scale = max(float(np.abs(wJ * (rd1.Vq @ x)).sum()) for x in rh)
    noq = E.RhsEngine(rd1, md1, ops1, E.CNS_MODAL, viscous_dissp=False)
    rn = noq.download(noq.rhs(Qd1))
    for x in rn:                                             # conservation (no penalty term, quirk Q3)
        assert abs(float((wJ * (rd1.Vq @ x)).sum())) <= 2e-9 * max(scale, 1.0)
    assert one.rhstest(Qd1, ref) < 0                         # LF + viscous dissipation
    c = np.ones((rd.Pq.shape[0], md.K))
    from esdg_cns_amd import physics as ph
    Qc = [np.asfortranarray(q) for q in ph.primitive_to_conservative(1.1 * c, .3 * c, -.2 * c, .9 * c)]
    r = sh.new_state()
    Qcd = sh.upload(Qc)
    for phs in range(sh.nphases):
        E.check(sh.L.esdg_rhs_phase(sh.ctx, phs, C.c_void_p(Qcd.data_ptr()), C.c_void_p(r.data_ptr()), None))
        for x, (after, before, nc) in enumerate(sh.xinfo):
            if after == phs:
                seg = {s[0]: s for s in sh.halo.segments[x]}
                for a, b in ((1, 7), (7, 1)):
                    sh.ws[seg[a][3]:seg[a][3] + seg[a][4]] = sh.ws[seg[b][1]:seg[b][1] + seg[b][2]]
    assert float(r.abs().max()) < 1e-11 / float(md.J.min())   # free stream through the ghost slots


@pytest.mark.parametrize("form,N,Kx,Kyr", [("cns", 4, 24, 3), ("cns", 3, 16, 2), ("euler", 4, 20, 2),
                                           ("cns", 3, 10, 24), ("euler", 4, 10, 16), ("cns", 8, 6, 3)])   # many rows: nested two-stream schedule
def test_rccl_transport_inside_the_library_loopback(E, form, N, Kx, Kyr):
    """The library's own RCCL transport and sharded schedule (esdg_comm_init / esdg_rhs on a sharded context: overlapped
    phases, packs, grouped ncclSend/ncclRecv on the comm stream) executed for real on one GPU: rank 0's strip of an 8-rank
    mesh with the communicator in loopback (every neighbour is this rank; what goes to the rank below comes back in as the
    ghost data of the rank above).  For a strip-periodic state that is what ranks 1 and 7 would send, so the result must
    equal the stand-alone periodic strip bit for bit -- also through the fused LSRK stage."""
    from esdg_cns_amd import setup_dg as sd
    nr = 8
    LY = 10.0 * Kyr / Kx

    def build(Ky_total, e0, e1):
        VX, VY, EToV = sd.uniform_quad_mesh(Kx, Ky_total)
        VX = 15 * (1 + VX) / 2
        VY = 5 * VY * (Ky_total / Kx)
        rd = sd.init_reference_quad(N) if form == "cns" else sd.init_reference_quad(N, sd.gauss_quad(0, 0, N))
        md = sd.init_mesh((VX, VY), EToV, rd, elem_range=(e0, e1))
        sd.make_periodic(md, rd)
        md.mapB = np.zeros(0, dtype=np.int64)
        ops = sd.cns_ops(rd) if form == "cns" else sd.euler_quad_ops(rd)
        sd.interp_geofacs_to_hybrid(md, ops["Vh"])
        return rd, md, ops

    code = E.CNS_MODAL if form == "cns" else E.EULER_COLLOCATED
    rd, md, ops = build(Kyr * nr, 0, Kx * Kyr)
    offsets = np.array([Kx * Kyr * r for r in range(nr + 1)], dtype=np.int64)
    sh = E.RhsEngine(rd, md, ops, code, rank=0, nranks=nr, rank_offsets=offsets)
    assert sh.attach_rccl(loopback=True) == 1 and sh.transport == "rccl"
    xs, ys = (md.x, md.y) if form == "cns" else (md.xq, md.yq)
    Q = _strip_periodic_state(xs, ys, LY)
    import copy
    _, md1s, _ = build(Kyr, 0, Kx * Kyr)                   # the stand-alone periodic strip: the shard's own arrays, its mapP
    md1 = copy.copy(md)
    md1.mapP, md1.elem_offset, md1.Kglobal = md1s.mapP, 0, md.K
    one = E.RhsEngine(rd, md1, ops, code)
    Qd = sh.upload(Q)
    # the overlapped schedule launches element ranges, so elements land in other lanes / groups than in the one-launch
    # evaluation: bit-equal wherever the kernels' arithmetic does not depend on the slot (see test_ranged_launches_*)
    same_inputs = True
    for rep in range(3):                                   # repeated evaluations reuse buffers, events and the comm stream
        got, ref = sh.rhs(Qd), one.rhs(Qd)
        torch.cuda.synchronize()
        rel = float((got - ref).abs().max() / ref.abs().max())
        assert rel <= 1e-11, rel
        if same_inputs:
            assert torch.equal(got, ref)
    # fused low-storage stage through the same transport
    q1, q2 = Qd.clone(), Qd.clone()
    r1, r2 = torch.zeros_like(Qd), torch.zeros_like(Qd)
    for k in range(3):
        sh.rhs_lsrk_fused(q1, r1, -0.4 * k, 0.3, 1e-3)
        one.rhs_lsrk_fused(q2, r2, -0.4 * k, 0.3, 1e-3)
    torch.cuda.synchronize()
    assert float((q1 - q2).abs().max()) <= 1e-12 * float(q2.abs().max())
    if same_inputs:
        assert torch.equal(q1, q2)
    assert sh.allreduce([1.5, -2.0], "sum") == [1.5, -2.0] and sh.allreduce([3.0], "max") == [3.0]
    # one DOPRI45 attempt (stages 2..7 + Hairer estimate, cavity_optimized.jl:1002-1021) through the sharded schedule: the
    # error norm is reduced over the communicator (here: this rank alone, so it must equal the stand-alone strip's)
    outs = []
    for eng in (sh, one):
        k = [torch.zeros_like(Qd) for _ in range(7)]
        eng.rhs_into(Qd, k[0])
        Qtmp = torch.empty_like(Qd)
        ptrs = (C.c_void_p * 7)(*[t.data_ptr() for t in k])
        err = C.c_double(0.0)
        E.check(eng.L.esdg_dopri45_attempt(eng.ctx, C.c_void_p(Qd.data_ptr()), C.c_void_p(Qtmp.data_ptr()), ptrs, 2e-3, 1e-5, C.byref(err),
                                           eng._stream()))
        torch.cuda.synchronize()
        outs.append((Qtmp, err.value))
    # (the norm's terms are added in one order whatever launches the last phase was cut into: same state bits, same estimate)
    assert torch.equal(outs[0][0], outs[1][0]) and outs[0][1] == outs[1][1] and outs[0][1] > 0, (outs[0][1], outs[1][1])
    if form == "cns":                # rhs_viscous!'s second return (visc_test, :802-806) through the sharded path: this rank's share
        vt = []
        for eng in (sh, one):
            v = C.c_double(0.0)
            E.check(eng.L.esdg_viscous_entropy_test(eng.ctx, C.c_void_p(Qd.data_ptr()), C.byref(v), eng._stream()))
            vt.append(v.value)
        assert vt[0] == vt[1] and vt[0] != 0.0


@pytest.mark.parametrize("case", ["cavity BCTYPE=1", "cavity BCTYPE=2", "cavity BCTYPE=3", "shocktube"])
def test_generic_fallback_serves_wall_meshes(E, case):
    """The generic pair-list kernels (the fallback when the operators do not factor into 1D tables; forced here with
    ESDG_FORCE_GENERIC=1 on the A/B build) apply the boundary closures too (round 5): init_BC_funs of
    dg2D_CNS_cavity_optimized.jl:135-265 (adiabatic no-slip, isothermal, slip; lid on y = +1) and the inflow / copy closures of
    dg2D_CNS_modalESDG.jl:161-217.  Gate: the oracle's, as for the tensor kernels (they keep one geometry record per element also in
    the wall elements' viscous operators, which the sum rhsRK! does not see at this size); and the two kernel sets agree to round-off."""
    from common import TOL, as_oracle_problem, becker_constants, product_cavity_problem, product_shocktube_problem, truth_gate
    from oracle import oracle as orc
    if case == "shocktube":
        st = becker_constants()
        rd, md, ops, Q = product_shocktube_problem(3, 8, 5)
        kw = dict(BCTYPE=4, viscous_dissp=False, mu=st["mu"], lam=st["lam"], Pr=st["Pr"], inflow=(st["rhoL"], st["uL"], st["vL"], st["pL"]))
        po = orc.build_cns_problem(3, 8, 5, bc="shocktube")
        p = as_oracle_problem(rd, md, ops, Q, Re=po.Re, mu=st["mu"], lam=st["lam"], Pr=st["Pr"], BCTYPE=4, inflow=kw["inflow"])
        o, q = orc.CnsOracle(p, viscous_dissp=False), orc.CnsOracle(p, quad=True, viscous_dissp=False)
    else:
        bct = int(case[-1])
        rd, md, ops, Q = product_cavity_problem(3, 6, 5)
        kw = dict(BCTYPE=bct)
        p = as_oracle_problem(rd, md, ops, Q, Re=1000.0, mu=1e-3, lam=-2e-3 / 3, Pr=.71, BCTYPE=bct)
        o, q = orc.CnsOracle(p), orc.CnsOracle(p, quad=True)
    os.environ["ESDG_FORCE_GENERIC"] = "1"
    try:
        gen = E.RhsEngine(rd, md, ops, E.CNS_MODAL, ab_hooks=True, **kw)
    finally:
        del os.environ["ESDG_FORCE_GENERIC"]
    assert gen.L.esdg_uses_tensor_kernels(gen.ctx) == 0
    ten = E.RhsEngine(rd, md, ops, E.CNS_MODAL, **kw)
    assert ten.L.esdg_uses_tensor_kernels(ten.ctx) == 1
    g, t = _rhs(gen, Q), _rhs(ten, Q)
    _, e_orc = truth_gate(f"generic kernels, {case} N=3", g, o.rhsRK(Q, False)[0], q.rhsRK(Q, False)[0])
    assert rel_l2(g, t) <= max(1e-11, 3 * e_orc)     # (two Float64 evaluations of the same statements)


@pytest.mark.parametrize("form", ["cns", "euler"])
def test_smooth_wave_short_cut_of_the_last_phase_is_bitwise_the_general_path(E, form):
    """kt3_rhs skips the logarithms of a wave whose densities / betas all agree to 0.49e-4 (every log-mean of such a wave is the
    reference's series branch, logmean.jl:23-27, which reads none).  Claim: bit-identical results.  ESDG_DBG=32 switches the short
    cut off; on the vortex box most waves of a 96x64 mesh take it (far field), the ones around the core do not, and on a state
    with 1 % node-to-node noise none does -- all three must agree bit for bit between the two settings."""
    import bench
    prob = product_cns_problem if form == "cns" else product_euler_problem
    rd, md, ops, Q = prob(4, 96, 64)
    states = {"vortex": Q, "rough": bench.rough_state(Q)}
    const = [np.full_like(q, v) for q, v in zip(Q, (1.3, 0.4, -0.2, 2.9))]
    states["uniform"] = const
    form_id = E.CNS_MODAL if form == "cns" else E.EULER_COLLOCATED
    lazy = E.RhsEngine(rd, md, ops, form_id)
    os.environ["ESDG_DBG"] = "32"
    try:
        full = E.RhsEngine(rd, md, ops, form_id, ab_hooks=True)
    finally:
        del os.environ["ESDG_DBG"]
    for name, S in states.items():
        a, b = _rhs(lazy, S), _rhs(full, S)
        for x, y in zip(a, b):
            assert np.array_equal(x, y), (form, name)


@pytest.mark.parametrize("form,N", [("cns", 4), ("cns", 2), ("euler", 3), ("euler", 6), ("cns", 8), ("euler", 8)])
def test_line_per_lane_and_node_per_lane_last_phase_kernels_agree(E, form, N):
    """kt3_rhs (production) against kt2_rhs (ESDG_V2=rhs): two mappings of the same formulas -- different summation orders, so
    round-off apart, not bitwise."""
    prob = product_cns_problem if form == "cns" else product_euler_problem
    rd, md, ops, Q = prob(N, 13, 9) if N < 8 else prob(N, 5, 4)   # 117 elements: a partial last group in both kernels
    Qs = steep_state(md.x, md.y) if form == "cns" else steep_state(md.xq, md.yq)
    form_id = E.CNS_MODAL if form == "cns" else E.EULER_COLLOCATED
    v3 = E.RhsEngine(rd, md, ops, form_id)
    os.environ["ESDG_V2"] = "rhs"
    try:
        v2 = E.RhsEngine(rd, md, ops, form_id, ab_hooks=True)
    finally:
        del os.environ["ESDG_V2"]
    assert rel_l2(_rhs(v3, Qs), _rhs(v2, Qs)) <= 1e-12


def test_degree_limits_are_refused_with_a_reason(E):
    """Quads: N = 1 ... 11 (tensor kernels; the generic pair-list kernels stop at N = 7), with or without walls.  Everything beyond is
    refused at esdg_create / at the call with a message, never run on a kernel that does not cover it.  (The visc_test diagnostic
    runs on kt2_sigma since round 5: every degree the context serves.)"""
    from common import product_cavity_problem
    rd, md, ops, Q = product_cns_problem(12, 2, 2)
    with pytest.raises(Exception, match="unsupported degree"):
        E.RhsEngine(rd, md, ops, E.CNS_MODAL)
    rd, md, ops, Q = product_cavity_problem(9, 2, 2)          # (round 5: walls at every degree the library serves)
    engw = E.RhsEngine(rd, md, ops, E.CNS_MODAL, BCTYPE=1)
    assert torch.isfinite(engw.rhs(engw.upload(Q))).all()
    rd, md, ops, Q = product_cns_problem(8, 2, 2)
    eng = E.RhsEngine(rd, md, ops, E.CNS_MODAL)
    Qd = eng.upload(Q)
    assert torch.isfinite(eng.rhs(Qd)).all()
    v = C.c_double(float("nan"))
    E.check(eng.L.esdg_viscous_entropy_test(eng.ctx, C.c_void_p(Qd.data_ptr()), C.byref(v), eng._stream()))
    assert np.isfinite(v.value)
    # ... and an attempt whose arrays coincide is refused before anything is launched
    k = [torch.zeros_like(Qd) for _ in range(7)]
    err = C.c_double(0.0)
    ptrs = (C.c_void_p * 7)(*[t.data_ptr() for t in k[:6]], k[0].data_ptr())
    with pytest.raises(Exception, match="nine distinct arrays"):
        E.check(eng.L.esdg_dopri45_attempt(eng.ctx, C.c_void_p(Qd.data_ptr()), C.c_void_p(k[6].data_ptr()), ptrs, 1e-3, 1e-5, C.byref(err), eng._stream()))
    os.environ["ESDG_FORCE_GENERIC"] = "1"
    try:
        with pytest.raises(Exception, match="tensor kernels only"):
            E.RhsEngine(rd, md, ops, E.CNS_MODAL, ab_hooks=True)
    finally:
        del os.environ["ESDG_FORCE_GENERIC"]
