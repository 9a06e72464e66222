"""Error functionals of the drivers evaluated on the device (esdg_error_*) against the
oracle's numpy restatement (oracle/ref_errors.py).

Tolerance: these functionals square small differences Q - Qexact, so a relative round-off eps in Qexact (device exp/pow/
log vs libm, a few ulp) shows up as eps*|Q|/|Q - Qexact| in the result: with errors >= 1e-6 that is <= 1e-9 relative."""
import numpy as np
import pytest

from common import cavity_state, product_cavity_problem, product_cns_problem, product_euler_problem, product_shocktube_problem

pytestmark = pytest.mark.gpu
RTOL = 1e-9


@pytest.fixture(scope="module")
def eng_mod():
    from esdg_cns_amd import engine
    return engine


def _close(a, b, rtol=RTOL):
    return abs(a - b) <= rtol * max(abs(a), abs(b))


@pytest.mark.parametrize("N,Kx,Ky", [(3, 8, 6), (4, 6, 6)])
def test_vortex_l2_error_collocated_and_modal(eng_mod, N, Kx, Ky):
    """L2err block of dg2D_euler_quad.jl:214-233 for a state at the Gauss nodes (the Euler-quad driver, projection
    folded in) and for LGL nodal values (the CNS drivers' layout)."""
    from esdg_cns_amd import setup_dg as sd
    from oracle import oracle as orc
    from oracle import ref_errors as re
    from oracle import ref_physics as rp
    Vq2, wq2 = sd.error_quadrature(N)
    T = 0.3
    # collocated: vortex sampled at a slightly different time, so the error is O(1e-1)
    p = orc.build_euler_problem(N, Kx, Ky)
    rd, md, ops, _ = product_euler_problem(N, Kx, Ky)
    Q = [np.asfortranarray(q) for q in rp.primitive_to_conservative(*rp.vortex(p.md.xq, p.md.yq, T + 0.05))]
    eng = eng_mod.RhsEngine(rd, md, ops, eng_mod.EULER_COLLOCATED)
    eng.setup_errors(rd, md, Vq2, wq2)
    got, sums = eng.l2_error(eng.upload(Q), T)
    ref = re.vortex_l2_error(Q, p.rd, p.md, T)
    print(f"collocated N={N}: L2err device {got:.15e} oracle {ref:.15e}")
    assert _close(got, ref) and _close(np.sqrt(sum(sums)), got, 1e-15)
    # the interpolant of the exact solution itself: error O(1e-3..1e-2), still above the conditioning bound
    Q0 = [np.asfortranarray(q) for q in rp.primitive_to_conservative(*rp.vortex(p.md.xq, p.md.yq, T))]
    got0, ref0 = eng.l2_error(eng.upload(Q0), T)[0], re.vortex_l2_error(Q0, p.rd, p.md, T)
    assert got0 < got and _close(got0, ref0, 1e-8)
    # modal / LGL nodal layout
    pc = orc.build_cns_problem(N, Kx, Ky, bc="periodic")
    rd, md, ops, _ = product_cns_problem(N, Kx, Ky)
    Qn = [np.asfortranarray(q) for q in rp.primitive_to_conservative(*rp.vortex(pc.md.x, pc.md.y, T + 0.05))]
    engm = eng_mod.RhsEngine(rd, md, ops, eng_mod.EULER_MODAL)
    engm.setup_errors(rd, md, Vq2, wq2)
    gotm = engm.l2_error(engm.upload(Qn), T)[0]
    refm = re.vortex_l2_error(Qn, pc.rd, pc.md, T, project=False)
    print(f"modal N={N}: L2err device {gotm:.15e} oracle {refm:.15e}")
    assert _close(gotm, refm)


def test_becker_shock_errors(eng_mod):
    """Exact travelling shock by bisection on the device (dg2D_CNS_modalESDG.jl:545-579) inside the L1/Linf errors of
    :745-771 and inside the L2 functional."""
    from esdg_cns_amd import setup_dg as sd
    from oracle import oracle as orc
    from oracle import ref_errors as re
    N, Kx, Ky = 2, 24, 4
    par = re.becker_par()
    p = orc.build_cns_problem(N, Kx, Ky, bc="shocktube", BCTYPE=4)
    rd, md, ops, _ = product_shocktube_problem(N, Kx, Ky)
    st = orc.becker_constants()
    eng = eng_mod.RhsEngine(rd, md, ops, eng_mod.CNS_MODAL, BCTYPE=4, viscous_dissp=False, mu=st["mu"], lam=st["lam"], Pr=st["Pr"],
                            inflow=(st["rhoL"], st["uL"], 0.0, st["pL"]))
    Vq2, wq2 = sd.error_quadrature(N)
    eng.setup_errors(rd, md, Vq2, wq2)
    T = 0.2
    Q = [np.asfortranarray(q) for q in re.exact_sol_viscous_shocktube(p.md.x, T - 0.02, par)]   # lags by 0.02
    L1, Linf, raw = eng.nodal_error(eng.upload(Q), T, exact=1, par=par)
    rL1, rLinf = re.shocktube_errors(Q, p.md, T, par)
    print(f"Becker: L1 {L1:.12e}/{rL1:.12e}  Linf {Linf:.12e}/{rLinf:.12e}")
    assert rL1 > 1e-4 and _close(L1, rL1) and _close(Linf, rLinf)
    assert _close(raw[1], np.abs(Q[0]).sum(), 1e-13) and raw[3] == np.abs(Q[0]).max()
    # exact nodal values at the same time: the device bisection reproduces the host's to a few ulp
    Qe = [np.asfortranarray(q) for q in re.exact_sol_viscous_shocktube(p.md.x, T, par)]
    L1e, Linfe, _ = eng.nodal_error(eng.upload(Qe), T, exact=1, par=par)
    assert L1e < 1e-13 and Linfe < 1e-12, (L1e, Linfe)
    # L2 functional against the Becker solution (the shock-tube example's own error print)
    got = eng.l2_error(eng.upload(Q), T, exact=1, par=par)[0]
    xq2 = Vq2 @ p.md.x
    ex = re.exact_sol_viscous_shocktube(xq2, T, par)
    wJ = wq2[:, None] * (Vq2 @ p.md.J)
    ref = np.sqrt(sum(np.sum(wJ * (Vq2 @ q - e) ** 2) for q, e in zip(Q, ex)))
    assert _close(got, ref), (got, ref)
    with pytest.raises(Exception):
        eng.l2_error(eng.upload(Q), T, exact=1, par=None)


def test_boundary_velocity_error(eng_mod):
    """dg2D_CNS_convergence_test.jl:1055-1080 with the lid profile (1+cos(pi x))/2 of :76."""
    from oracle import oracle as orc
    from oracle import ref_errors as re
    N, K1D = 3, 6
    vl = lambda x: (1 + np.cos(np.pi * x)) / 2
    p = orc.build_cns_problem(N, K1D, K1D, bc="cavity", BCTYPE=1)
    rd, md, ops, Q = product_cavity_problem(N, K1D, K1D)
    eng = eng_mod.RhsEngine(rd, md, ops, eng_mod.CNS_MODAL, BCTYPE=1, vlid=vl)
    eng.setup_errors(rd, md, boundary=True)
    ex, full, sums = eng.boundary_velocity_error(eng.upload(Q), 2.0 / K1D)
    rex, rfull, rsums = re.boundary_velocity_error(p.Q, p.rd, p.md, K1D, vl)
    print(f"boundary velocity error: executed {ex:.12e}/{rex:.12e} written {full:.12e}/{rfull:.12e}")
    assert _close(ex, rex, 1e-12) and _close(full, rfull, 1e-12)
    assert all(_close(a, b, 1e-12) for a, b in zip(sums, rsums)) and rsums[2] > 1e-3
    # without the L2 quadrature / without walls the other entry points refuse
    with pytest.raises(Exception):
        eng.l2_error(eng.upload(Q), 0.0)
    rdp, mdp, opsp, Qp = product_cns_problem(N, 4, 4)
    engp = eng_mod.RhsEngine(rdp, mdp, opsp, eng_mod.CNS_MODAL)
    with pytest.raises(Exception):
        engp.boundary_velocity_error(engp.upload(Qp), 0.5)
    engp.setup_errors(rdp, mdp, boundary=True)
    with pytest.raises(Exception):
        engp.boundary_velocity_error(engp.upload(Qp), 0.5)
