"""The oracle's restatement of the drivers' error functionals (oracle/ref_errors.py), pinned by mathematical properties:
the reference has no tests or fixtures for these blocks."""
import math

import numpy as np

from oracle import oracle as orc
from oracle import ref_errors as re
from oracle import ref_physics as rp


def test_becker_profile_solves_its_equation_and_meets_the_end_states():
    par = re.becker_par()
    v_0, v_1, v_01, m_0, L_k, v_inf = par
    st = orc.becker_constants()
    g = 1.4
    for x in (-0.05, -0.01, 0.0, 0.02, 0.05):     # inside the shock layer (farther out v saturates at v_0 / v_1)
        v = re.bisection_solve_velocity(x, par)
        f = -x + 2 * L_k / (g + 1) * (v_0 / (v_0 - v_1) * math.log((v_0 - v) / (v_0 - v_01))
                                      - v_1 / (v_0 - v_1) * math.log((v - v_1) / (v_01 - v_1)))
        assert v_1 < v < v_0 and abs(f) < 1e-12
    assert abs(re.bisection_solve_velocity(0.0, par) - v_01) < 1e-13          # the profile is centred on v_01
    # far upstream / downstream: the left and right states of :46-57 (the inflow state the closures impose)
    rho, rhou, rhov, E = re.exact_sol_viscous_shocktube(np.array([-0.5, 1.0]), 0.0, par)
    assert abs(rho[0] - st["rhoL"]) < 1e-9 and abs(rhou[0] / rho[0] - st["uL"]) < 1e-9
    assert abs(rho[1] - st["rhoR"]) < 1e-9 and abs(rhou[1] / rho[1] - st["uR"]) < 1e-9
    pL = 0.4 * (E[0] - .5 * rhou[0] ** 2 / rho[0])
    assert abs(pL - st["pL"]) < 1e-9 and np.all(rhov == 0)
    # travelling wave: the profile at time t is the profile at 0 shifted by v_inf*t
    a = re.exact_sol_viscous_shocktube(np.array([0.1]), 0.3, par)
    b = re.exact_sol_viscous_shocktube(np.array([0.1 - v_inf * 0.3]), 0.0, par)
    assert all(np.array_equal(x, y) for x, y in zip(a, b))


def test_vortex_l2_error_of_the_interpolant_converges_at_order_N_plus_1():
    N = 3
    errs = []
    for Kx in (12, 24):
        p = orc.build_euler_problem(N, Kx, Kx)
        Q = rp.primitive_to_conservative(*rp.vortex(p.md.xq, p.md.yq, 0.25))
        errs.append(re.vortex_l2_error(Q, p.rd, p.md, 0.25))
    rate = math.log2(errs[0] / errs[1])
    assert errs[1] < errs[0] and rate > N, (errs, rate)
    # shifted in time by dt the error is ~ |dQ/dt| dt: the functional sees the exact solution's time argument
    assert re.vortex_l2_error(Q, p.rd, p.md, 0.35) > 10 * errs[1]


def test_shocktube_errors_of_exact_nodal_values_vanish_and_scale():
    par = re.becker_par()
    p = orc.build_cns_problem(2, 6, 4, bc="shocktube", BCTYPE=4)
    Q = list(re.exact_sol_viscous_shocktube(p.md.x, 0.1, par))
    L1, Linf = re.shocktube_errors(Q, p.md, 0.1, par)
    assert L1 == 0.0 and Linf == 0.0
    Q2 = [Q[0] * 1.01, Q[1], Q[2], Q[3]]
    L1, Linf = re.shocktube_errors(Q2, p.md, 0.1, par)
    assert abs(L1 - 0.01 / 1.01) < 1e-12 and abs(Linf - 0.01 / 1.01) < 1e-12


def test_boundary_velocity_error_sums():
    K1D, N = 4, 2
    p = orc.build_cns_problem(N, K1D, K1D, bc="cavity", BCTYPE=1)
    rho = 1.0 + 0.1 * p.md.x
    Q = [rho, rho * 0.3, rho * -0.2, 2.0 + 0 * rho]          # u = (0.3, -0.2) everywhere
    ex, full, (t2, tw, tl) = re.boundary_velocity_error(Q, p.rd, p.md, K1D, vlid_fun=lambda x: 0.3 + 0 * x)
    # Jf*sum(wf) per face = (2/K1D)*2; 4*K1D boundary faces, K1D of them on the lid
    assert abs(t2 - 0.04 * 16) < 1e-13 and abs(tw - 0.09 * 12) < 1e-13 and abs(tl) < 1e-26
    assert ex == math.sqrt(t2) and full == math.sqrt(t2 + tw + tl)
