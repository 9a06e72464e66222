"""ORACLE (test infrastructure, NOT product code) -- numpy restatement of the error functionals the reference's
drivers print.  Parity unpinned by the reference's own tests (it has none for these
blocks); pinned here by mathematical properties (tests/test_errors_cpu.py): the bisection root satisfies the Becker
profile equation, the profile tends to the left/right states, and the L2 error of an interpolant converges at N+1.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this.

  vortex_l2_error             <- examples/dg2D_euler_quad.jl:214-233
  becker_par / bisection_solve_velocity / exact_sol_viscous_shocktube
                              <- examples/CompressibleNS/dg2D_CNS_modalESDG.jl:31-61, 545-579
  shocktube_errors            <- examples/CompressibleNS/dg2D_CNS_modalESDG.jl:745-771
  boundary_velocity_error     <- examples/CompressibleNS/dg2D_CNS_convergence_test.jl:1055-1080
"""
import numpy as np

from . import ref_physics as rp
from . import ref_setup as rs

GAMMA = 1.4


def vortex_l2_error(Q, rd, md, T, project=True):
    """dg2D_euler_quad.jl:214-233.  Q: state at the Gauss nodes (project=True applies "Q = Pq*Q" first, :215) or the
    LGL nodal values; error with the (N+2) Gauss rule."""
    if project:
        Q = [rd.Pq @ q for q in Q]
    rq2, sq2, wq2 = rs.quad_nodes_2D(rd.N + 2)
    Vq2 = rs.rdiv(rs.vandermonde_2D(rd.N, rq2, sq2), rd.VDM)
    wJq2 = np.diag(wq2) @ (Vq2 @ md.J)
    xq2, yq2 = Vq2 @ md.x, Vq2 @ md.y
    Qq = [Vq2 @ q for q in Q]
    Qex = rp.primitive_to_conservative(*rp.vortex(xq2, yq2, T))
    L2err = 0.0
    for fld in range(len(Q)):
        L2err += np.sum(wJq2 * (Qq[fld] - Qex[fld]) ** 2)
    return np.sqrt(L2err)


def becker_par(M_0=3.0, mu=0.01, Pr=3 / 4, v_inf=0.2, m_0=1.0, v_0=1.0):
    """Constants of dg2D_CNS_modalESDG.jl:31-46 -> (v_0, v_1, v_01, m_0, L_k, v_inf), L_k = kappa/m_0/cv (:550)."""
    g = GAMMA
    cp, cv = g / (g - 1), 1 / (g - 1)
    kappa = mu * cp / Pr
    v_1 = (g - 1 + 2 / M_0 ** 2) / (g + 1)
    v_01 = np.sqrt(v_0 * v_1)
    return (v_0, v_1, v_01, m_0, kappa / m_0 / cv, v_inf)


def bisection_solve_velocity(x, par, max_iter=100, tol=1e-14):
    """dg2D_CNS_modalESDG.jl:545-569, one scalar x at a time exactly as written."""
    v_0, v_1, v_01, m_0, L_k, _ = par
    g = GAMMA

    def f(v):
        with np.errstate(divide="ignore"):
            return -x + 2 * L_k / (g + 1) * (v_0 / (v_0 - v_1) * np.log((v_0 - v) / (v_0 - v_01))
                                             - v_1 / (v_0 - v_1) * np.log((v - v_1) / (v_01 - v_1)))
    v_L, v_R = v_1, v_0
    v_new = (v_L + v_R) / 2
    for _ in range(max_iter):
        v_new = (v_L + v_R) / 2
        if abs(f(v_new)) < tol:
            return v_new
        elif np.sign(f(v_L)) == np.sign(f(v_new)):
            v_L = v_new
        else:
            v_R = v_new
    return v_new


def exact_sol_viscous_shocktube(x, t, par):
    """dg2D_CNS_modalESDG.jl:573-579 broadcast over an array x -> (rho, rho u, rho v, E)."""
    v_0, v_1, v_01, m_0, L_k, v_inf = par
    g = GAMMA
    xs = np.asarray(x, dtype=np.float64)
    u = np.array([bisection_solve_velocity(xi - v_inf * t, par) for xi in xs.ravel()]).reshape(xs.shape)
    rho = m_0 / u
    e = 1 / (2 * g) * ((g + 1) / (g - 1) * v_01 ** 2 - u ** 2)
    return rho, rho * (v_inf + u), np.zeros_like(xs), rho * (e + 1 / 2 * (v_inf + u) ** 2)


def shocktube_errors(Q, md, T, par):
    """dg2D_CNS_modalESDG.jl:745-771 -> (L1err, Linferr); the rho*v terms are commented out there."""
    ex = exact_sol_viscous_shocktube(md.x, T, par)
    rho, rhou, _, E = Q
    Linferr = (np.max(np.abs(ex[0] - rho)) / np.max(np.abs(rho)) + np.max(np.abs(ex[1] - rhou)) / np.max(np.abs(rhou))
               + np.max(np.abs(ex[3] - E)) / np.max(np.abs(E)))
    J = md.J.flatten(order="F")[0]      # the script takes the first entry of J: it assumes a uniform mesh (:767)
    L1err = (np.sum(J * np.abs(ex[0] - rho)) / np.sum(J * np.abs(rho)) + np.sum(J * np.abs(ex[1] - rhou)) / np.sum(J * np.abs(rhou))
             + np.sum(J * np.abs(ex[3] - E)) / np.sum(J * np.abs(E)))
    return L1err, Linferr


def boundary_velocity_error(Q, rd, md, K1D, vlid_fun=lambda x: (1 + np.cos(np.pi * x)) / 2):
    """dg2D_CNS_convergence_test.jl:1055-1080 -> (err as executed, err as written, the three sums)."""
    mapB = np.asarray(md.mapB, dtype=np.int64) - 1
    flat = lambda a: a.flatten(order="F")
    xb, yb = flat(md.xf)[mapB], flat(md.yf)[mapB]
    lid = mapB[np.abs(yb - 1) < 1e-12]
    wall = mapB[np.abs(yb - 1) >= 1e-12]
    boundary = np.concatenate([lid, wall])
    vlid = vlid_fun(flat(md.xf)[lid])
    u_1 = flat(rd.Vf @ (Q[1] / Q[0]))
    u_2 = flat(rd.Vf @ (Q[2] / Q[0]))
    Jf = 2.0 / K1D
    WF = flat(np.repeat(rd.wf[:, None], md.K, axis=1))
    # :1075-1077.  Julia ends the statement `err = sum(Jf*WF[boundary].*u_2_boundary.^2)` at the line end; the two
    # following lines begin with a unary plus and are evaluated and dropped.  Both readings are returned.
    t_u2 = np.sum(Jf * WF[boundary] * u_2[boundary] ** 2)
    t_wall = np.sum(Jf * WF[wall] * u_1[wall] ** 2)
    t_lid = np.sum(Jf * WF[lid] * (u_1[lid] - vlid) ** 2)
    return np.sqrt(t_u2), np.sqrt(t_u2 + t_wall + t_lid), (t_u2, t_wall, t_lid)
