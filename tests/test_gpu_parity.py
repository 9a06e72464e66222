"""GPU parity tests proper: the hand-written HIP path (through the C ABI) against the CPU oracle on IDENTICAL inputs
(the same set-up objects and state arrays go to the engine, to the Float64 oracle and to the truth evaluator).

Gate (tests/common.py:truth_gate), per configuration, max over the conserved fields of the relative L2 norm:

    e_gpu = |gpu - truth| / |truth|  <=  max(1e-12, 2 * e_orc),      e_orc = |oracle_f64 - truth| / |truth|

`truth` = the oracle's statements compiled for IEEE binary128 (oracle/liboracle_quad.so, same double inputs, rounded to
double once).  1e-12 is BASELINE.json's tolerance; e_orc is what a faithful Float64 evaluation of the reference's own
formulas loses on the same input (its logmean cancels up to four digits at |f| >= 1e-4), i.e. what Julia itself is away
from the exact result; on the BASELINE vortex states e_orc is 4e-12 ... 6e-11, so 1e-12 alone cannot be met by ANY
Float64 implementation there, the reference included.  On well-conditioned states (`steep_state`) the strict 1e-12 holds.
Measured values are printed and written to gpurun_out/parity_errors.json (committed copies: profiles/parity_r04.json, parity_r03.json, parity_r02.json).
"""
import numpy as np
import pytest

from common import (TOL, as_oracle_problem, product_cavity_problem, product_cns_problem, product_euler_problem, rel_l2,
                    steep_state, truth_gate)

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def eng_mod():
    from esdg_cns_amd import engine
    return engine


def _gpu_rhs(eng, Q):
    return eng.download(eng.rhs(eng.upload(Q)))


def _euler(p):
    from oracle import oracle as orc
    o, q = orc.EulerOracle(p), orc.EulerOracle(p, quad=True)
    return (lambda Q: o.rhs(Q)[0]), (lambda Q: q.rhs(Q)[0])


def _cns(p, **kw):
    from oracle import oracle as orc
    o, q = orc.CnsOracle(p, **kw), orc.CnsOracle(p, quad=True, **kw)
    return o, q


PHYS = dict(Re=1000.0, mu=1e-3, lam=-2e-3 / 3, Pr=.71, BCTYPE=1)


@pytest.mark.parametrize("N,Kx,Ky", [(3, 16, 16), (4, 12, 8), (4, 64, 64), (2, 9, 7), (1, 6, 6), (5, 5, 4), (6, 4, 3), (7, 3, 3), (8, 4, 3),
                                     (9, 3, 2), (10, 3, 2), (11, 2, 3)])
def test_euler_collocated_matches_oracle(eng_mod, oracle_lib, N, Kx, Ky):
    """`rhs` of examples/dg2D_euler_quad.jl:141-194; (3, 16, 16) is BASELINE config 1 at its exact size."""
    rd, md, ops, Q = product_euler_problem(N, Kx, Ky)
    p = as_oracle_problem(rd, md, ops, Q)
    f64, truth = _euler(p)
    eng = eng_mod.RhsEngine(rd, md, p.ops, eng_mod.EULER_COLLOCATED)
    truth_gate(f"euler N={N} {Kx}x{Ky} vortex", _gpu_rhs(eng, Q), f64(Q), truth(Q))
    Qw = steep_state(md.xq, md.yq)
    e_gpu, _ = truth_gate(f"euler N={N} {Kx}x{Ky} steep", _gpu_rhs(eng, Qw), f64(Qw), truth(Qw))
    if N >= 3:
        assert e_gpu <= TOL, e_gpu         # strict north-star bound where the reference is well conditioned (64x64: 4.3e-13)


@pytest.mark.parametrize("form,N,Kx,Ky", [("euler", 4, 1, 1), ("euler", 3, 2, 1), ("euler", 2, 1, 3), ("cns", 4, 1, 1), ("cns", 4, 2, 1),
                                          ("cns", 3, 1, 3), ("cns", 2, 1, 1)])
def test_degenerate_periodic_meshes_match_oracle(eng_mod, oracle_lib, form, N, Kx, Ky):
    """The smallest periodic meshes: one element that is its own neighbour on all four faces, two elements that meet on
    two faces each, a single column -- every group is a partial group, every mapP entry wraps."""
    if form == "euler":
        rd, md, ops, Q = product_euler_problem(N, Kx, Ky)
        p = as_oracle_problem(rd, md, ops, Q)
        f64, truth = _euler(p)
        eng = eng_mod.RhsEngine(rd, md, p.ops, eng_mod.EULER_COLLOCATED)
        Qw = steep_state(md.xq, md.yq)
        truth_gate(f"euler N={N} {Kx}x{Ky} tiny", _gpu_rhs(eng, Qw), f64(Qw), truth(Qw))
    else:
        rd, md, ops, Q = product_cns_problem(N, Kx, Ky)
        p = as_oracle_problem(rd, md, ops, Q, **PHYS)
        o, q = _cns(p)
        eng = eng_mod.RhsEngine(rd, md, ops, eng_mod.CNS_MODAL, Re=p.Re, mu=p.mu, lam=p.lam, Pr=p.Pr)
        Qw = steep_state(md.x, md.y)
        truth_gate(f"cns N={N} {Kx}x{Ky} tiny", _gpu_rhs(eng, Qw), o.rhsRK(Qw, False)[0], q.rhsRK(Qw, False)[0])


@pytest.mark.parametrize("N,Kx,Ky", [(4, 12, 8), (4, 64, 64), (3, 10, 10), (2, 7, 9), (1, 5, 5), (5, 4, 4), (4, 7, 3), (5, 5, 3), (6, 6, 5),
                                     (7, 4, 3), (8, 4, 3), (9, 3, 2), (10, 3, 2), (11, 2, 3)])
def test_cns_modal_matches_oracle(eng_mod, oracle_lib, N, Kx, Ky):
    """`rhsRK!` of dg2D_CNS_cavity_optimized.jl:955-972 on the periodic vortex box (BASELINE config 3's formulation)."""
    rd, md, ops, Q = product_cns_problem(N, Kx, Ky)
    p = as_oracle_problem(rd, md, ops, Q, **PHYS)
    o, q = _cns(p)
    eng = eng_mod.RhsEngine(rd, md, ops, eng_mod.CNS_MODAL, Re=p.Re, mu=p.mu, lam=p.lam, Pr=p.Pr)
    truth_gate(f"cns N={N} {Kx}x{Ky} vortex", _gpu_rhs(eng, Q), o.rhsRK(Q, False)[0], q.rhsRK(Q, False)[0])
    Qw = steep_state(md.x, md.y)
    e_gpu, _ = truth_gate(f"cns N={N} {Kx}x{Ky} steep", _gpu_rhs(eng, Qw), o.rhsRK(Qw, False)[0], q.rhsRK(Qw, False)[0])
    if N >= 3:                             # (N=2 7x9: the Float64 oracle itself is 1.7e-12 from the truth; 64x64 N=4: 7.7e-13)
        assert e_gpu <= TOL, e_gpu


def test_oracle_built_inputs_give_the_same_verdict(eng_mod, oracle_lib):
    """The same gate with the ORACLE's set-up objects (oracle/ref_setup.py, the statement-by-statement restatement of the
    reference set-up) fed to the engine: the C ABI takes the arrays a Julia driver holds, whoever built them."""
    from oracle import oracle as orc
    p = orc.build_cns_problem(4, 12, 8, bc="periodic")
    o, q = _cns(p)
    eng = eng_mod.RhsEngine(p.rd, p.md, p.ops, eng_mod.CNS_MODAL, Re=p.Re, mu=p.mu, lam=p.lam, Pr=p.Pr)
    truth_gate("cns N=4 12x8 vortex (oracle-built inputs)", _gpu_rhs(eng, p.Q), o.rhsRK(p.Q, False)[0], q.rhsRK(p.Q, False)[0])
    pe = orc.build_euler_problem(3, 16, 16)
    f64, truth = _euler(pe)
    eng = eng_mod.RhsEngine(pe.rd, pe.md, pe.ops, eng_mod.EULER_COLLOCATED)
    truth_gate("euler N=3 16x16 vortex (oracle-built inputs)", _gpu_rhs(eng, pe.Q), f64(pe.Q), truth(pe.Q))
    # ... the lid-driven cavity (walls, all three BC types) and the hexahedral box the same way
    for bct in (1, 2, 3):
        pc = orc.build_cns_problem(4, 6, 5, bc="cavity", BCTYPE=bct)
        o, q = _cns(pc)
        eng = eng_mod.RhsEngine(pc.rd, pc.md, pc.ops, eng_mod.CNS_MODAL, Re=pc.Re, mu=pc.mu, lam=pc.lam, Pr=pc.Pr, BCTYPE=bct)
        truth_gate(f"cavity BCTYPE={bct} N=4 6x5 (oracle-built inputs)", _gpu_rhs(eng, pc.Q), o.rhsRK(pc.Q, False)[0], q.rhsRK(pc.Q, False)[0])
    ph = orc.build_hex_problem(3, 3, 2, 2)
    ho, hq = orc.HexOracle(ph, 0.25), orc.HexOracle(ph, 0.25, quad=True)
    eng = eng_mod.RhsEngine(ph.rd, ph.md, ph.ops, eng_mod.EULER_HEX_COLLOCATED, lf_scale=0.25)
    truth_gate("hex N=3 (3, 2, 2) lf=0.25 (oracle-built inputs)", _gpu_rhs(eng, ph.Q), ho.rhs(ph.Q)[0], hq.rhs(ph.Q)[0])


def test_modal_euler_matches_oracle_inviscid(eng_mod, oracle_lib):
    rd, md, ops, Q = product_cns_problem(4, 8, 8)
    p = as_oracle_problem(rd, md, ops, Q, **PHYS)
    o, q = _cns(p)
    eng = eng_mod.RhsEngine(rd, md, ops, eng_mod.EULER_MODAL)
    for name, Qx in (("vortex", Q), ("steep", steep_state(md.x, md.y))):
        truth_gate(f"rhs_inviscid! N=4 8x8 {name}", _gpu_rhs(eng, Qx), o.rhs_inviscid(Qx), q.rhs_inviscid(Qx))


# The viscous part ALONE (rhs_viscous!, esdg_set_parts(2)) is held to the ordinary 2 x e_orc gate like everything else.  Until late
# in round 3 the adiabatic no-slip cavity (BCTYPE=1) stood out as the mesh was refined: 2.6 x at N=4 8x8, 8.9 x at 64x64.  Cause
# (tools/cavity_visc_probe.py + tools/cavity_visc_attribution.py): 98 % of the squared error sat in the lid / wall elements'
# momentum rows, identically for the v2 and the round-1 kernels, with the penalty on or off.  dg_grad! / dg_div!
# (dg2D_CNS_cavity_optimized.jl:549-611) multiply by the metric terms and divide by J NODE BY NODE in the nodal basis (rows 1:Np of
# rxJ ..., J[i,e]); on an affine element those arrays are constants plus the set-up's round-off (5.6e-14 relative at 8x8, 5.3e-13 at
# 64x64), and the kernels held one geometry record per element.  With BCTYPE=1 the lifted wall jump dominates those rows and the
# reference evaluates it almost exactly (e_orc 8e-16 / 8e-15 in fields 2 / 3), so the geometry round-off showed.  kt2_sigma now
# repeats gradient and volume divergence of the elements with a boundary node in the nodal basis with the driver's per-node
# arrays (MeshDev::wgeo), kt2_rhs divides their result by the per-node J: 1.2-1.3 x from 8x8 to 128x128.  ESDG_WALL_GEOMETRY=element switches that off (A/B; the old figures come back).
VISC_FACTOR = 2.0


@pytest.mark.parametrize("BCTYPE", [1, 2, 3])
@pytest.mark.parametrize("N,Kx,Ky", [(3, 6, 5), (4, 8, 8), (5, 4, 3), (4, 2, 2), (8, 3, 2), (9, 3, 2), (11, 2, 2)])   # (2x2: fewer elements than one group holds; N=8: kt2_rhs's wall instantiation; N=9 ... 11: kt3_rhs's, one wave per SIMD)
def test_cns_wall_boundary_conditions_match_oracle(eng_mod, oracle_lib, BCTYPE, N, Kx, Ky):
    """Lid-driven cavity walls (init_BC_funs, dg2D_CNS_cavity_optimized.jl:135-265): adiabatic no-slip (1),
    isothermal (2), slip (3), lid on y=+1."""
    rd, md, ops, Q = product_cavity_problem(N, Kx, Ky)
    assert md.mapB.size == 2 * (Kx + Ky) * (N + 1)
    p = as_oracle_problem(rd, md, ops, Q, **dict(PHYS, BCTYPE=BCTYPE))
    o, q = _cns(p)
    kw = dict(Re=p.Re, mu=p.mu, lam=p.lam, Pr=p.Pr, BCTYPE=BCTYPE)
    eng = eng_mod.RhsEngine(rd, md, ops, eng_mod.CNS_MODAL, **kw)
    # N >= 9: 2.5 x e_orc instead of 2 (measured 1.9 at N = 9, 2.2 at N = 11; N = 8: 1.6).  Not the kernels' arithmetic -- IEEE divisions,
    # the library logarithm, corrected quotients change nothing to two digits -- but the REPRESENTATION of the operators: the truth
    # evaluates the driver's dense VhP exactly, the 1e-16 ... 2e-15 of set-up round-off in its mathematically zero / one entries
    # included, the kernels apply the same operator from 1D tables; on this low-Mach state the momentum rows cancel so far that one ulp
    # per entry of VhP moves them by 0.3-0.6 x e_orc, growing with N (tools/lowmach_probe.py, profiles/experiments/r05_lowmach_probe.txt).
    truth_gate(f"cavity BCTYPE={BCTYPE} N={N} {Kx}x{Ky} rhsRK!", _gpu_rhs(eng, Q), o.rhsRK(Q, False)[0], q.rhsRK(Q, False)[0],
               factor=2.5 if N >= 9 else 2.0)
    # rhs_viscous! alone (esdg_set_parts(2)); field 1 is identically zero
    eng.set_parts(2)
    gv = _gpu_rhs(eng, Q)
    assert np.abs(gv[0]).max() == 0.0
    truth_gate(f"cavity BCTYPE={BCTYPE} N={N} {Kx}x{Ky} rhs_viscous!", gv[1:], o.rhs_viscous(Q)[0][1:], q.rhs_viscous(Q)[0][1:],
               factor=VISC_FACTOR)


def test_cns_variable_lid_velocity_matches_oracle(eng_mod, oracle_lib):
    """Per-node lid velocity (esdg_mesh_t.vlid): (1+cos(pi*xlid))/2 of dg2D_CNS_convergence_test.jl:72-76."""
    vl = lambda x: (1 + np.cos(np.pi * x)) / 2
    for N, Kx, Ky in [(3, 6, 5), (4, 4, 4)]:
        rd, md, ops, Q = product_cavity_problem(N, Kx, Ky)
        p = as_oracle_problem(rd, md, ops, Q, **PHYS)
        p.vlid = vl
        o, q = _cns(p)
        kw = dict(Re=p.Re, mu=p.mu, lam=p.lam, Pr=p.Pr, BCTYPE=1)
        eng = eng_mod.RhsEngine(rd, md, ops, eng_mod.CNS_MODAL, vlid=vl, **kw)
        got = _gpu_rhs(eng, Q)
        truth_gate(f"variable lid N={N}", got, o.rhsRK(Q, False)[0], q.rhsRK(Q, False)[0])
        ones = _gpu_rhs(eng_mod.RhsEngine(rd, md, ops, eng_mod.CNS_MODAL, **kw), Q)
        assert rel_l2(got, ones) > 1e-6
        # vlid given as an array of ones reproduces the default bit for bit
        same = _gpu_rhs(eng_mod.RhsEngine(rd, md, ops, eng_mod.CNS_MODAL, vlid=1.0, **kw), Q)
        assert all(np.array_equal(a, b) for a, b in zip(same, ones))


@pytest.mark.parametrize("bc,BCTYPE,N,Kx,Ky", [("periodic", 1, 3, 6, 5), ("cavity", 1, 3, 6, 5), ("cavity", 3, 3, 6, 5),
                                               ("periodic", 1, 8, 3, 2), ("cavity", 1, 8, 3, 2)])   # (N = 8: visc_test on kt2_sigma, round 5)
def test_rhs_inviscid_viscous_split_and_rhsRK_diagnostics(eng_mod, oracle_lib, bc, BCTYPE, N, Kx, Ky):
    """esdg_set_parts: 1 = rhs_inviscid! (:447), 2 = rhs_viscous! (:749) against the oracle's separate restatements, and
    the three returns of rhsRK! (:955-972): rhsQ, rhstest, rhstest_visc (visc_test from esdg_viscous_entropy_test)."""
    rd, md, ops, Q = (product_cns_problem if bc == "periodic" else product_cavity_problem)(N, Kx, Ky)
    p = as_oracle_problem(rd, md, ops, Q, **dict(PHYS, BCTYPE=BCTYPE))
    o, q = _cns(p)
    eng = eng_mod.RhsEngine(rd, md, ops, eng_mod.CNS_MODAL, BCTYPE=BCTYPE)
    Qd = eng.upload(Q)
    tot = eng.rhs(Qd)
    eng.set_parts(1)
    inv = eng.download(eng.rhs(Qd))
    eng.set_parts(2)
    vis = eng.download(eng.rhs(Qd))
    eng.set_parts(3)
    truth_gate(f"split {bc} BCTYPE={BCTYPE} rhs_inviscid!", inv, o.rhs_inviscid(Q), q.rhs_inviscid(Q))
    ref_v, visc_test = o.rhs_viscous(Q)
    truth_gate(f"split {bc} BCTYPE={BCTYPE} rhs_viscous!", vis[1:], ref_v[1:], q.rhs_viscous(Q)[0][1:], factor=VISC_FACTOR)
    assert np.abs(vis[0]).max() == 0.0
    assert rel_l2([a + b for a, b in zip(inv, vis)], eng.download(tot)) <= 1e-13
    ref, rt_ref, rtv_ref = o.rhsRK(Q)
    _, rt_q, rtv_q = q.rhsRK(Q)
    rt, rtv = eng.rhsRK_diagnostics(Qd, tot)
    print(f"{bc} BCTYPE={BCTYPE}: rhstest {rt:.6e} (oracle {rt_ref:.6e}, truth {rt_q:.6e})  rhstest_visc {rtv:.6e} "
          f"(oracle {rtv_ref:.6e}, truth {rtv_q:.6e}) visc_test {visc_test:.6e}")
    # the diagnostics are sums over the mesh of terms of both signs: gate them like the fields, against the truth values,
    # relative to the size of their largest part
    scale = max(abs(rt_q), abs(rtv_q), abs(visc_test), 1e-6)
    for name, g, r64, t in (("rhstest", rt, rt_ref, rt_q), ("rhstest_visc", rtv, rtv_ref, rtv_q)):
        assert abs(g - t) <= max(1e-11 * scale, 4 * abs(r64 - t)), (name, g, r64, t)
    # the reference-signature wrapper returns the same triple
    out, rt2, rtv2 = eng_mod.rhsRK(Q, rd, md, ops, BCTYPE=BCTYPE)
    assert rt2 == rt and rtv2 == rtv
    truth_gate(f"split {bc} BCTYPE={BCTYPE} rhsRK wrapper", out, ref, q.rhsRK(Q, False)[0])


def test_shocktube_inflow_outflow_closures_match_oracle(eng_mod, oracle_lib):
    """BCTYPE 4: the boundary closures of examples/CompressibleNS/dg2D_CNS_modalESDG.jl:161-217 (Dirichlet inflow state,
    copy on the outflow side, lam = lamP = 0, sigma+ = sigma-, no penalty), periodic in y, on quads."""
    from common import becker_constants, product_shocktube_problem
    from oracle import oracle as orc
    N, Kx, Ky = 3, 8, 5
    po = orc.build_cns_problem(N, Kx, Ky, bc="shocktube")
    rd, md, ops, Q = product_shocktube_problem(N, Kx, Ky)
    assert np.array_equal(md.mapP, po.md.mapP) and np.array_equal(np.sort(md.mapB), np.sort(po.md.mapB))
    st = becker_constants()
    p = as_oracle_problem(rd, md, ops, Q, Re=po.Re, mu=st["mu"], lam=st["lam"], Pr=st["Pr"], BCTYPE=4,
                          inflow=(st["rhoL"], st["uL"], st["vL"], st["pL"]))
    o, q = _cns(p, viscous_dissp=False)
    eng = eng_mod.RhsEngine(rd, md, ops, eng_mod.CNS_MODAL, BCTYPE=4, viscous_dissp=False, mu=st["mu"], lam=st["lam"], Pr=st["Pr"],
                            inflow=(st["rhoL"], st["uL"], st["vL"], st["pL"]))
    truth_gate(f"shock-tube closures N={N} {Kx}x{Ky}", _gpu_rhs(eng, Q), o.rhsRK(Q, False)[0], q.rhsRK(Q, False)[0])
    # uniform inflow state is steady
    c = [np.full_like(Q[0], v) for v in (st["rhoL"], st["rhoL"] * st["uL"], 0.0, st["pL"] / 0.4 + .5 * st["rhoL"] * st["uL"] ** 2)]
    assert max(np.abs(x).max() for x in _gpu_rhs(eng, c)) < 1e-10
    # the ABI refuses the penalty with these closures (the driver has that block commented out)
    with pytest.raises(Exception):
        eng_mod.RhsEngine(rd, md, ops, eng_mod.CNS_MODAL, BCTYPE=4, viscous_dissp=True, inflow=(1.0, 1.2, 0.0, 0.08))


@pytest.mark.parametrize("BCTYPE", [1, 3])
def test_sheared_parallelogram_mesh_matches_oracle(eng_mod, oracle_lib, BCTYPE):
    """Affine but not axis-aligned elements (x -> x + 0.35 y): all four metric terms rxJ, sxJ, ryJ, syJ and both
    components of every face normal are non-zero, walls are oblique.  CNS with wall closures and modal Euler."""
    N, Kx, Ky, sh = 3, 6, 5, 0.35
    rd, md, ops, Q = product_cavity_problem(N, Kx, Ky, shear=sh)
    assert np.abs(md.sxJ).min() > 1e-3 or np.abs(md.ryJ).min() > 1e-3      # the off-diagonal metrics are exercised
    p = as_oracle_problem(rd, md, ops, Q, **dict(PHYS, BCTYPE=BCTYPE))
    o, q = _cns(p)
    eng = eng_mod.RhsEngine(rd, md, ops, eng_mod.CNS_MODAL, BCTYPE=BCTYPE)
    truth_gate(f"sheared cavity BCTYPE={BCTYPE} rhsRK!", _gpu_rhs(eng, Q), o.rhsRK(Q, False)[0], q.rhsRK(Q, False)[0])
    eng = eng_mod.RhsEngine(rd, md, ops, eng_mod.EULER_MODAL, BCTYPE=BCTYPE)
    truth_gate(f"sheared cavity BCTYPE={BCTYPE} rhs_inviscid!", _gpu_rhs(eng, Q), o.rhs_inviscid(Q), q.rhs_inviscid(Q))


def test_graded_mesh_every_element_its_own_geometry(eng_mod, oracle_lib):
    """Non-uniform rectangles (vertices graded by x + g sin(pi x)/pi): J, metrics and normals differ from element to
    element, so any mix-up of per-element geometry records between the lanes/waves of a workgroup would show."""
    N, Kx, Ky, g = 4, 9, 7, 0.45
    rd, md, ops, Q = product_cns_problem(N, Kx, Ky, grade=g)
    assert md.J.max() / md.J.min() > 2
    p = as_oracle_problem(rd, md, ops, Q, **PHYS)
    o, q = _cns(p)
    eng = eng_mod.RhsEngine(rd, md, ops, eng_mod.CNS_MODAL)
    truth_gate("graded CNS mesh", _gpu_rhs(eng, Q), o.rhsRK(Q, False)[0], q.rhsRK(Q, False)[0])


def test_product_and_oracle_setups_feed_the_same_rhs(eng_mod, oracle_lib):
    """End to end with two independent set-ups: product set-up -> engine against oracle set-up -> oracle.  The set-ups agree
    to round-off in the float arrays (bit-exact in the maps, tests/test_setup.py); the two RHS results then differ by the
    oracle's own sensitivity to that input round-off, which is measured here by running the ORACLE on both input sets."""
    from oracle import oracle as orc
    N, Kx, Ky = 4, 16, 16
    po = orc.build_cns_problem(N, Kx, Ky, bc="periodic")
    rd, md, ops, Q = product_cns_problem(N, Kx, Ky)
    assert np.array_equal(md.mapP, po.md.mapP)
    pp = as_oracle_problem(rd, md, ops, Q, **PHYS)
    ref_o = orc.CnsOracle(po).rhsRK(po.Q, False)[0]
    ref_p = orc.CnsOracle(pp).rhsRK(Q, False)[0]
    setup_sens = rel_l2(ref_p, ref_o)
    got = _gpu_rhs(eng_mod.RhsEngine(rd, md, ops, eng_mod.CNS_MODAL), Q)
    d = rel_l2(got, ref_o)
    print(f"two set-ups, N={N} {Kx}x{Ky}: gpu(product inputs) vs oracle(oracle inputs) {d:.2e}; oracle on both inputs {setup_sens:.2e}")
    truth = orc.CnsOracle(pp, quad=True).rhsRK(Q, False)[0]
    assert d <= max(TOL, 2 * rel_l2(ref_p, truth) + 2 * setup_sens)


# ------------------------------------------------------------------------------------------------------------------------
# Oracle VALUES at the headline sizes.  The Float64 oracle runs on all usable host cores (element
# loops under OpenMP: 2-3 s at 512^2); the binary128 truth costs ~9 ms per CNS element and thread (512^2: ~4 min on the 16
# cores of a GPU box), so by default it runs at 256^2 and the 512^2 evaluation is held against the Float64 oracle with a
# bound derived from it; ESDG_TRUTH_512=1 runs the binary128 evaluation at 512^2 too (recorded once in
# profiles/parity_r03.json).  Measured on CPU (oracle alone): e_orc of the vortex state doubles with every refinement
# (CNS N=4: 1.9e-11, 3.0e-11, 6.0e-11, 1.2e-10 at 16^2 ... 128^2; Euler 7e-12 ... 4.5e-11): round-off of O(1) fluxes
# divided by J ~ h^2.
# ------------------------------------------------------------------------------------------------------------------------
def _all_cores():
    import os
    from oracle import oracle as orc
    n = len(os.sched_getaffinity(0))
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    orc.lib().oracle_set_threads(n)
    orc.lib_quad().oracle_set_threads(n)
    return n


def _one_core():
    from oracle import oracle as orc
    orc.lib().oracle_set_threads(1)
    orc.lib_quad().oracle_set_threads(1)


def test_cfg2_euler_256_exact_size_values_against_oracle_and_truth(eng_mod, oracle_lib):
    """BASELINE config 2 (2D Euler, N=4, 256x256) at its exact size: `rhs` of examples/dg2D_euler_quad.jl:141-194,
    GPU vs the Float64 oracle vs the binary128 truth on identical inputs, same gate as every other case."""
    rd, md, ops, Q = product_euler_problem(4, 256, 256)
    p = as_oracle_problem(rd, md, ops, Q)
    f64, truth = _euler(p)
    eng = eng_mod.RhsEngine(rd, md, p.ops, eng_mod.EULER_COLLOCATED)
    _all_cores()
    try:
        truth_gate("cfg2 euler N=4 256x256 vortex", _gpu_rhs(eng, Q), f64(Q), truth(Q))
        Qw = steep_state(md.xq, md.yq)
        # (no strict 1e-12 here: at this resolution neighbouring nodes of the steep state differ by |f| ~ 4e-4, close to
        # logmean's ill-conditioned window, and the oracle's own error grows like 1/h: 6e-14 at 12x8, 4e-13 at 64x64)
        truth_gate("cfg2 euler N=4 256x256 steep", _gpu_rhs(eng, Qw), f64(Qw), truth(Qw))
    finally:
        _one_core()


def test_cfg2_euler_256_exact_size_with_oracle_built_inputs(eng_mod, oracle_lib):
    """The same configuration at its exact size with the ORACLE's set-up objects (oracle/ref_setup.py: the reference set-up restated
    statement by statement -- Vandermonde inversions, distance-matrix connectivity) fed to engine, oracle and truth alike: the gate
    does not depend on whose set-up built the arrays (round 5; the small cases: test_oracle_built_inputs_give_the_same_verdict)."""
    from oracle import oracle as orc
    pe = orc.build_euler_problem(4, 256, 256)
    f64, truth = _euler(pe)
    eng = eng_mod.RhsEngine(pe.rd, pe.md, pe.ops, eng_mod.EULER_COLLOCATED)
    _all_cores()
    try:
        truth_gate("cfg2 euler N=4 256x256 vortex (oracle-built inputs)", _gpu_rhs(eng, pe.Q), f64(pe.Q), truth(pe.Q))
    finally:
        _one_core()


def test_cfg3_cns_512_exact_size_values_against_oracle(eng_mod, oracle_lib):
    """BASELINE config 3 (2D CNS, N=4, 512x512, the headline workload) at its exact size: `rhsRK!` of
    dg2D_CNS_cavity_optimized.jl:955-972, GPU vs the Float64 oracle vs the binary128 truth on identical inputs, all host
    cores (measured on a GPU box's 16-core share: the truth takes ~100 s; round 3: e_gpu 5.09e-10, e_orc 5.03e-10).
    ESDG_TRUTH_512=0 runs the truth at 256x256 instead and holds the 512x512 evaluation against the Float64 oracle alone:
    |gpu - f64| <= 3 x the e_orc expected there (twice the e_orc measured at 256x256; e_gpu + e_orc <= 3 e_orc under the gate)."""
    import os
    _all_cores()
    try:
        full_truth = os.environ.get("ESDG_TRUTH_512", "1") != "0"
        e_orc_256 = None
        if not full_truth:
            rd, md, ops, Q = product_cns_problem(4, 256, 256)
            p = as_oracle_problem(rd, md, ops, Q, **PHYS)
            o, q = _cns(p)
            eng = eng_mod.RhsEngine(rd, md, ops, eng_mod.CNS_MODAL, Re=p.Re, mu=p.mu, lam=p.lam, Pr=p.Pr)
            _, e_orc_256 = truth_gate("cns N=4 256x256 vortex", _gpu_rhs(eng, Q), o.rhsRK(Q, False)[0], q.rhsRK(Q, False)[0])
            del eng, o, q, p
        rd, md, ops, Q = product_cns_problem(4, 512, 512)
        p = as_oracle_problem(rd, md, ops, Q, **PHYS)
        o, q = _cns(p)
        eng = eng_mod.RhsEngine(rd, md, ops, eng_mod.CNS_MODAL, Re=p.Re, mu=p.mu, lam=p.lam, Pr=p.Pr)
        got, ref = _gpu_rhs(eng, Q), o.rhsRK(Q, False)[0]
        if full_truth:
            truth_gate("cfg3 cns N=4 512x512 vortex", got, ref, q.rhsRK(Q, False)[0])
        else:
            d = rel_l2(got, ref)
            print(f"cfg3 cns N=4 512x512 vortex: gpu-vs-oracle={d:.2e}  (e_orc at 256x256: {e_orc_256:.2e})")
            assert d <= 3 * 2 * e_orc_256, (d, e_orc_256)
    finally:
        _one_core()


def test_cfg3_cns_512_exact_size_with_oracle_built_inputs(eng_mod, oracle_lib):
    """The headline configuration at its exact size with the ORACLE's set-up objects (oracle/ref_setup.py, ~25 s at this size) fed to
    engine, Float64 oracle and binary128 truth alike -- the gate at the headline size does not rest on the product's own set-up
    arrays (parity: "none at a headline size").  ESDG_TRUTH_512=0 skips it (the truth takes ~100 s on 16 cores)."""
    import os
    if os.environ.get("ESDG_TRUTH_512", "1") == "0":
        pytest.skip("ESDG_TRUTH_512=0: the binary128 truth at 512x512 is switched off")
    from oracle import oracle as orc
    p = orc.build_cns_problem(4, 512, 512, bc="periodic")
    o, q = _cns(p)
    eng = eng_mod.RhsEngine(p.rd, p.md, p.ops, eng_mod.CNS_MODAL, Re=p.Re, mu=p.mu, lam=p.lam, Pr=p.Pr)
    _all_cores()
    try:
        truth_gate("cfg3 cns N=4 512x512 vortex (oracle-built inputs)", _gpu_rhs(eng, p.Q), o.rhsRK(p.Q, False)[0], q.rhsRK(p.Q, False)[0])
    finally:
        _one_core()


def test_cavity_state_with_exact_zeros_of_the_normal_velocity_looser_documented_bound(eng_mod, oracle_lib):
    """The cavity state WITHOUT the phase shift of common.cavity_state: u = .1 sin(pi x) cos(pi y) vanishes exactly on the
    element interfaces x = 0, +-1/2 ..., so rhoU_n there is an exact zero plus round-off and the reference's wavespeed
    sqrt(|u_n|) (quirk Q1, euler_variables.jl:7-10) turns 1e-17 into 3e-9: any two Float64 implementations -- the oracle
    and the truth evaluator included -- differ by that much in the LF term.  Documented looser bound: 1e-6 relative (measured
    values are printed); everything else in the suite uses the shifted state and the 2 x e_orc gate."""
    from esdg_cns_amd import physics as ph
    N, Kx, Ky = 4, 8, 8
    rd, md, ops, _ = product_cavity_problem(N, Kx, Ky)
    x, y = md.x, md.y
    rho = 1.0 + .2 * np.exp(-10 * (x ** 2 + y ** 2))
    u = .1 * np.sin(np.pi * x) * np.cos(np.pi * y)
    v = -.1 * np.cos(np.pi * x) * np.sin(np.pi * y)
    p = (1 / (.3 ** 2 * ph.GAMMA)) * rho ** ph.GAMMA
    Q = [np.asfortranarray(q) for q in ph.primitive_to_conservative(rho, u, v, p)]
    pr = as_oracle_problem(rd, md, ops, Q, **PHYS)
    o, q = _cns(pr)
    eng = eng_mod.RhsEngine(rd, md, ops, eng_mod.CNS_MODAL, Re=pr.Re, mu=pr.mu, lam=pr.lam, Pr=pr.Pr, BCTYPE=1)
    got, ref, tru = _gpu_rhs(eng, Q), o.rhsRK(Q, False)[0], q.rhsRK(Q, False)[0]
    e_gpu, e_orc = rel_l2(got, tru), rel_l2(ref, tru)
    print(f"unshifted cavity state N={N} {Kx}x{Ky}: e_gpu={e_gpu:.2e} e_orc={e_orc:.2e} gpu-vs-oracle={rel_l2(got, ref):.2e}")
    assert e_gpu <= 1e-6 and e_orc <= 1e-6


def test_cavity_64x64_rhsRK_and_rhs_viscous_alone_within_the_gate(eng_mod, oracle_lib):
    """VERDICT r02 item 4.  On a wall mesh at N=4, 64x64 (every closure active) both the sum `rhsRK!` -- what every driver
    integrates -- and `rhs_viscous!` ALONE pass the ordinary 2 x e_orc gate.  With one geometry record per element in the
    viscous operators of the wall elements (ESDG_WALL_GEOMETRY=element, the state of the library until late in round 3) the
    viscous part alone sits at 8.9 x its own e_orc: recorded beside it (see VISC_FACTOR above)."""
    _all_cores()
    try:
        rd, md, ops, Q = product_cavity_problem(4, 64, 64)
        p = as_oracle_problem(rd, md, ops, Q, **PHYS)
        o, q = _cns(p)
        kw = dict(Re=p.Re, mu=p.mu, lam=p.lam, Pr=p.Pr, BCTYPE=1)
        eng = eng_mod.RhsEngine(rd, md, ops, eng_mod.CNS_MODAL, **kw)
        truth_gate("cavity BCTYPE=1 N=4 64x64 rhsRK!", _gpu_rhs(eng, Q), o.rhsRK(Q, False)[0], q.rhsRK(Q, False)[0])
        eng.set_parts(2)
        tv, ov = q.rhs_viscous(Q)[0][1:], o.rhs_viscous(Q)[0][1:]
        truth_gate("cavity BCTYPE=1 N=4 64x64 rhs_viscous! alone", _gpu_rhs(eng, Q)[1:], ov, tv, factor=VISC_FACTOR)
        import os
        os.environ["ESDG_WALL_GEOMETRY"] = "element"
        try:
            old = eng_mod.RhsEngine(rd, md, ops, eng_mod.CNS_MODAL, ab_hooks=True, **kw)
        finally:
            del os.environ["ESDG_WALL_GEOMETRY"]
        old.set_parts(2)
        e_old, e_orc = rel_l2(_gpu_rhs(old, Q)[1:], tv), rel_l2(ov, tv)
        print(f"cavity BCTYPE=1 N=4 64x64 rhs_viscous! alone, one geometry record per element: e_gpu={e_old:.2e} = {e_old / e_orc:.1f} x e_orc")
        assert e_old > 4 * e_orc      # (what the nodal-basis path is there for; measured 8.9)
    finally:
        _one_core()
