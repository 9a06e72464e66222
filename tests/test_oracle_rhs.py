"""Pins the full-RHS ORACLE (CPU): the loop-structured C restatement (oracle/oracle_rhs.c) against the
independent vectorised numpy restatement (oracle/ref_rhs_numpy.py), against the committed golden
vectors, and against the invariants that pin the reference scheme: free-stream
preservation, conservation, entropy conservation with the LF penalty off, entropy dissipation with it
on, dissipativity of the viscous terms."""
import os

import numpy as np
import pytest

from common import noise_floor, rel_l2, steep_state
from oracle import oracle as orc
from oracle import ref_physics as ph
from oracle import ref_rhs_numpy as rr

GOLD = os.path.join(os.path.dirname(__file__), "golden")


@pytest.mark.parametrize("N,Kx,Ky", [(3, 8, 8), (2, 5, 4), (4, 4, 4)])
def test_euler_c_vs_numpy(oracle_lib, N, Kx, Ky):
    p = orc.build_euler_problem(N, Kx, Ky)
    eo = orc.EulerOracle(p)
    a, ta = rr.euler_rhs(p.Q, p.md, p.ops, True)
    b, tb = eo.rhs(p.Q, .5, True)
    floor = noise_floor(lambda q: eo.rhs(q)[0], p.Q)
    assert rel_l2(a, b) <= max(1e-12, 4 * floor)
    assert abs(ta - tb) < 1e-11 * max(1, abs(tb))
    Qs = steep_state(p.md.xq, p.md.yq)
    assert rel_l2(rr.euler_rhs(Qs, p.md, p.ops)[0], eo.rhs(Qs)[0]) <= 1e-12


@pytest.mark.parametrize("bc,BCTYPE", [("periodic", 1), ("cavity", 1), ("cavity", 2), ("cavity", 3)])
def test_cns_c_vs_numpy(oracle_lib, bc, BCTYPE):
    p = orc.build_cns_problem(3, 5, 5, bc=bc, BCTYPE=BCTYPE)
    co = orc.CnsOracle(p)
    bcf = rr.BCFuns(p.md, BCTYPE)
    a, t1, t2 = rr.rhsRK(p.Q, p.rd, p.md, p.ops, bcf, p.Re, p.lam, p.mu, p.Pr)
    b, s1, s2 = co.rhsRK(p.Q)
    floor = noise_floor(lambda q: co.rhsRK(q, False)[0], p.Q)
    assert rel_l2(a, b) <= max(1e-12, 4 * floor), (rel_l2(a, b), floor)
    assert abs(t1 - s1) <= 1e-9 * max(1.0, abs(s1)) and abs(t2 - s2) <= 1e-9 * max(1.0, abs(s2))
    av, _ = rr.rhs_viscous(p.Q, p.md, p.rd, bcf, p.Re, p.lam, p.mu, p.Pr)
    bv, _ = co.rhs_viscous(p.Q)
    assert rel_l2(av[1:], bv[1:]) <= 1e-11 and np.abs(bv[0]).max() == 0.0


def test_cns_variable_lid_velocity(oracle_lib):
    """Lid velocity (1+cos(pi*xlid))/2 of dg2D_CNS_convergence_test.jl:72-76 (cavity_optimized uses ones, :147)."""
    vl = lambda x: (1 + np.cos(np.pi * x)) / 2
    p = orc.build_cns_problem(3, 5, 4, bc="cavity", BCTYPE=1)
    base = orc.CnsOracle(p).rhsRK(p.Q)[0]
    p.vlid = vl
    co = orc.CnsOracle(p)
    b = co.rhsRK(p.Q)[0]
    a = rr.rhsRK(p.Q, p.rd, p.md, p.ops, rr.BCFuns(p.md, 1, vlid=vl), p.Re, p.lam, p.mu, p.Pr)[0]
    floor = noise_floor(lambda q: co.rhsRK(q, False)[0], p.Q)
    assert rel_l2(a, b) <= max(1e-12, 4 * floor), (rel_l2(a, b), floor)
    assert rel_l2(b, base) > 1e-6          # the lid profile is felt
    # only elements that touch the lid (top row) or their BR1 neighbours change
    changed = np.abs(np.stack(b) - np.stack(base)).max(axis=(0, 1)) > 0
    assert not changed[:5 * 2].any() and changed[-5:].all()


def test_golden_vectors(oracle_lib):
    g = np.load(os.path.join(GOLD, "rhs_euler_N2_3x3.npz"))
    p = orc.build_euler_problem(2, 3, 3)
    assert np.array_equal(np.stack(p.Q), g["Q"])
    out, rt = orc.EulerOracle(p).rhs(p.Q, .5, True)
    assert np.allclose(np.stack(out), g["rhs"], rtol=0, atol=1e-12) and abs(rt - g["rhstest"]) < 1e-12
    g = np.load(os.path.join(GOLD, "rhs_cns_N2_3x3.npz"))
    p = orc.build_cns_problem(2, 3, 3, bc="periodic")
    out, rt, rtv = orc.CnsOracle(p).rhsRK(p.Q)
    assert np.allclose(np.stack(out), g["rhs"], rtol=0, atol=1e-12)
    assert abs(rt - g["rhstest"]) < 1e-12 and abs(rtv - g["rhstest_visc"]) < 1e-12


def _const_state(shape):
    one = np.ones(shape)
    return [np.asfortranarray(q) for q in ph.primitive_to_conservative(1.1 * one, .3 * one, -.2 * one, .9 * one)]


def test_free_stream_and_conservation_euler(oracle_lib):
    p = orc.build_euler_problem(3, 6, 5)
    eo = orc.EulerOracle(p)
    r, _ = eo.rhs(_const_state(p.Q[0].shape))
    assert max(np.abs(x).max() for x in r) < 1e-12                     # invariant (2)
    r, _ = eo.rhs(p.Q)
    assert max(abs(np.sum(p.md.wJq * x)) for x in r) < 1e-11           # invariant (4)


def test_entropy_conservation_and_dissipation_euler(oracle_lib):
    p = orc.build_euler_problem(3, 6, 5)
    eo = orc.EulerOracle(p)
    _, rt0 = eo.rhs(p.Q, 0.0, True)        # LF off: entropy conservative, dg2D_euler_quad.jl:186-191
    _, rt1 = eo.rhs(p.Q, 0.5, True)
    assert abs(rt0) < 1e-12 and rt1 < -1e-6                            # invariant (3)


def test_cns_invariants(oracle_lib):
    p = orc.build_cns_problem(3, 5, 4, bc="periodic")
    co = orc.CnsOracle(p)
    r, _, _ = co.rhsRK(_const_state(p.Q[0].shape))
    assert max(np.abs(x).max() for x in r) < 1e-11
    r, rt, rtv = co.rhsRK(p.Q)
    Vq = p.rd.Vq
    # conservation holds without the entropy-variable jump penalty; the reference's penalty
    # tau*[[v]] uses the one-sided tau = -1/(Re*v4^-) and no sJ/J scaling (quirk Q3), so it is not conservative
    rc = orc.CnsOracle(p, viscous_dissp=False).rhsRK(p.Q, False)[0]
    assert max(abs(np.sum(p.md.wJq * (Vq @ x))) for x in rc) < 1e-11
    ec = orc.CnsOracle(p, inviscid_dissp=False, viscous_dissp=True)
    inv = ec.rhs_inviscid(p.Q)
    VU = ph.v_ufun(*[Vq @ q for q in p.Q])
    assert abs(sum(np.sum(p.md.wJq * v * (Vq @ x)) for v, x in zip(VU, inv))) < 1e-12   # EC without LF
    assert rt < 0 and rtv < 1e-12                                      # invariant (6): dissipative


def test_omp_threads_do_not_change_results(oracle_lib):
    p = orc.build_cns_problem(2, 6, 6, bc="periodic")
    co = orc.CnsOracle(p)
    a = co.rhsRK(p.Q, False)[0]
    orc.lib().oracle_set_threads(4)
    try:
        b = co.rhsRK(p.Q, False)[0]
    finally:
        orc.lib().oracle_set_threads(1)
    assert all(np.array_equal(x, y) for x, y in zip(a, b))


def test_shocktube_closures_c_vs_numpy_and_consistency(oracle_lib):
    """BCTYPE 4 = init_BC_funs of examples/CompressibleNS/dg2D_CNS_modalESDG.jl:161-217 (Dirichlet inflow, copy outflow,
    lam = 0 and sigma+ = sigma- on both, no penalty) on the quad element: the two restatements agree, and a uniform
    field equal to the inflow state is a steady state (the Dirichlet data is consistent with the interior)."""
    p = orc.build_cns_problem(3, 6, 4, bc="shocktube")
    assert p.BCTYPE == 4 and p.md.mapB.size == 2 * 4 * 4
    co = orc.CnsOracle(p, viscous_dissp=False)
    bc = rr.InflowBCFuns(p.md, p.inflow)
    a, t1, t2 = rr.rhsRK(p.Q, p.rd, p.md, p.ops, bc, p.Re, p.lam, p.mu, p.Pr, True, False)
    b, s1, s2 = co.rhsRK(p.Q)
    floor = noise_floor(lambda q: co.rhsRK(q, False)[0], p.Q)
    assert rel_l2(a, b) <= max(1e-12, 4 * floor), (rel_l2(a, b), floor)
    assert abs(t1 - s1) <= 1e-9 * max(1.0, abs(s1)) and abs(t2 - s2) <= 1e-9 * max(1.0, abs(s2))
    st = orc.becker_constants()
    Qc = [np.full_like(p.Q[0], v) for v in ph.primitive_to_conservative(st["rhoL"], st["uL"], 0.0, st["pL"])]
    assert max(np.abs(x).max() for x in co.rhsRK(Qc, False)[0]) < 1e-10
    # the Dirichlet data acts on the first column of elements (and, through the BR1 gradient of its neighbour, on the
    # second); nothing else moves
    p2 = orc.build_cns_problem(3, 6, 4, bc="shocktube")
    p2.inflow = (st["rhoL"] * 1.1, st["uL"], 0.0, st["pL"])
    d = [x - y for x, y in zip(orc.CnsOracle(p2, viscous_dissp=False).rhsRK(p.Q, False)[0], b)]
    cols = np.abs(np.stack(d)).max(axis=(0, 1)).reshape(4, 6)          # (Ky, Kx) element blocks
    assert cols[:, 0].min() > 1e-6 and cols[:, 2:].max() == 0.0


def test_truth_evaluator_is_the_same_statements_in_binary128(oracle_lib):
    """The binary128 build of oracle/oracle_rhs.c (`make liboracle_quad.so`, -DORACLE_QUAD) is what every GPU parity test is
    gated against.  Pin it on the CPU: it reports a 128-bit working type, agrees with the Float64 build to the Float64 build's
    own round-off on the vortex states (a few 1e-12 ... 1e-10: the conditioning DESIGN.md section 2 describes) and to
    ~1e-14 on the well-conditioned state, preserves the free stream far below Float64 round-off, and conserves entropy
    with the LF penalty off beyond what Float64 can show."""
    Lq = orc.lib_quad()
    assert Lq.oracle_real_bits() == 128 and orc.lib().oracle_real_bits() == 64      # storage bits of `real`
    p = orc.build_euler_problem(3, 6, 5)
    e64, e128 = orc.EulerOracle(p), orc.EulerOracle(p, quad=True)
    d_vortex = rel_l2(e64.rhs(p.Q)[0], e128.rhs(p.Q)[0])
    Qs = steep_state(p.md.xq, p.md.yq)
    d_steep = rel_l2(e64.rhs(Qs)[0], e128.rhs(Qs)[0])
    print(f"euler N=3 6x5: |f64 - f128| vortex {d_vortex:.2e}, steep {d_steep:.2e}")
    assert 1e-14 < d_vortex < 1e-9 and d_steep < 1e-12
    one = [np.full_like(q, v) for q, v in zip(p.Q, (1.3, 0.4, -0.3, 2.9))]
    r128 = e128.rhs(one)[0]
    assert max(np.abs(x).max() for x in r128) < 1e-12      # what is left is the round-off of the Float64 operators it is fed
    # entropy conservation (LF off): the Float64 build is limited by its own round-off, the binary128 build by the inputs'
    _, t64 = e64.rhs(p.Q, 0.0, True)
    _, t128 = e128.rhs(p.Q, 0.0, True)
    assert abs(t128) <= abs(t64) + 1e-13 and abs(t64) < 1e-10
    pc = orc.build_cns_problem(3, 4, 4)
    c64, c128 = orc.CnsOracle(pc), orc.CnsOracle(pc, quad=True)
    d_cns = rel_l2(c64.rhsRK(pc.Q, False)[0], c128.rhsRK(pc.Q, False)[0])
    print(f"cns N=3 4x4: |f64 - f128| {d_cns:.2e}")
    assert 1e-14 < d_cns < 1e-9


def test_viscous_operators_read_their_geometry_node_by_node(oracle_lib):
    """dg_grad! / dg_div! scale nodal coefficients by rows 1:Np of the metric arrays and by J[i,e] node by node
    (dg2D_CNS_cavity_optimized.jl:549-611); on an affine element those arrays are a constant plus the set-up's round-off.
    With adiabatic no-slip walls the lifted wall jump dominates the momentum rows of the boundary elements and the Float64
    evaluation is almost exact there (e_orc ~ 1e-15), so that round-off is visible: the same oracle fed with ELEMENT MEANS of
    those arrays -- what the device kernels hold, one record per element -- moves `rhs_viscous!` by tens of e_orc in exactly
    those rows, and by nothing measurable in the interior.  This pins why kt2_sigma repeats gradient and volume divergence of
    the elements with a boundary node in the nodal basis with the per-node arrays (MeshDev::wgeo; tests/test_gpu_parity.py:
    VISC_FACTOR; tools/cavity_visc_attribution.py)."""
    import copy

    from common import as_oracle_problem, product_cavity_problem
    N, Kx, Ky = 4, 8, 8
    rd, md, ops, Q = product_cavity_problem(N, Kx, Ky)
    phys = dict(Re=1000.0, mu=1e-3, lam=-2e-3 / 3, Pr=.71, BCTYPE=1)
    p = as_oracle_problem(rd, md, ops, Q, **phys)
    tv = orc.CnsOracle(p, quad=True).rhs_viscous(Q)[0]
    ov = orc.CnsOracle(p).rhs_viscous(Q)[0]
    Np = rd.Pq.shape[0]
    md2 = copy.copy(md)
    for nm in ("J", "rxJ", "sxJ", "ryJ", "syJ"):
        a = getattr(md, nm).copy()
        if nm == "J":
            a[:] = a.mean(axis=0)
        else:
            a[:Np] = a[:Np].mean(axis=0)
        setattr(md2, nm, a)
    mv = orc.CnsOracle(as_oracle_problem(rd, md2, ops, Q, **phys)).rhs_viscous(Q)[0]
    K = Kx * Ky
    ex, ey = np.arange(K) % Kx, np.arange(K) // Kx
    bnd = (ex == 0) | (ex == Kx - 1) | (ey == 0) | (ey == Ky - 1)
    ring = ((ex == 1) | (ex == Kx - 2) | (ey == 1) | (ey == Ky - 2)) & ~bnd
    inner = ~(bnd | ring)
    err = lambda a, f, m: np.linalg.norm((a[f] - tv[f])[:, m]) / np.linalg.norm(tv[f][:, m])
    for f in (1, 2):                                   # the momentum rows
        assert err(ov, f, bnd) < 2e-14                 # the reference's own statements: almost exact on the boundary elements
        assert err(mv, f, bnd) > 10 * err(ov, f, bnd)  # element means: 69 x / 25 x of that
        assert err(mv, f, inner) < 1.5 * err(ov, f, inner)   # invisible in the interior (1.1 x)
    assert rel_l2(mv[1:], tv[1:]) > 1.5 * rel_l2(ov[1:], tv[1:])    # and enough to leave the 2 x gate as a whole (1.9e-13 vs 9.5e-14)
