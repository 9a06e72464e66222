#!/usr/bin/env python3
"""The Becker viscous shock-tube driver examples/CompressibleNS/dg2D_CNS_modalESDG.jl on the MI355X engine with the
reference *quad* element (the script meshes the same box with triangles): exact travelling viscous shock as initial
condition (bisection, :545-578), Dirichlet inflow / copy outflow closures (BCTYPE 4, :161-217), periodic in y,
adaptive DOPRI45 (:655-720), error against the exact solution at the final time.

  python examples/dg2D_CNS_shocktube_quad.py [N] [K1D] [T]
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from esdg_cns_amd import engine, setup_dg as sd, timestep  # noqa: E402

# "Becker viscous shocktube" constants, dg2D_CNS_modalESDG.jl:31-61
G = 1.4
M_0, MU, PR = 3.0, 0.01, 3 / 4
LAM = 2 / 3 * MU
CP, CV = G / (G - 1), 1 / (G - 1)
KAPPA = MU * CP / PR
V_INF, M0, V0 = 0.2, 1.0, 1.0
V1 = (G - 1 + 2 / M_0 ** 2) / (G + 1)
V01 = np.sqrt(V0 * V1)
UL, RHOL = V0 + V_INF, M0 / V0
EL_ = 1 / (2 * G) * ((G + 1) / (G - 1) * V01 ** 2 - V0 ** 2)
PL = (G - 1) * RHOL * EL_


def bisection_solve_velocity(x, max_iter=100, tol=1e-14):
    """:545-569, vectorised over x."""
    L_k = KAPPA / M0 / CV
    f = lambda v: -x + 2 * L_k / (G + 1) * (V0 / (V0 - V1) * np.log((V0 - v) / (V0 - V01)) - V1 / (V0 - V1) * np.log((v - V1) / (V01 - V1)))
    vL, vR = np.full_like(x, V1), np.full_like(x, V0)
    v = .5 * (vL + vR)
    for _ in range(max_iter):
        v = .5 * (vL + vR)
        with np.errstate(divide="ignore", invalid="ignore"):
            fv, fL = f(v), f(vL)
        done = np.abs(fv) < tol
        left = (~done) & (np.sign(fL) == np.sign(fv))
        vL = np.where(left, v, vL)
        vR = np.where((~done) & ~left, v, vR)
    return v


def exact_sol_viscous_shocktube(x, t):
    """:574-579 -> (rho, rho u, rho v, E)."""
    u = bisection_solve_velocity(x - V_INF * t)
    rho = M0 / u
    e = 1 / (2 * G) * ((G + 1) / (G - 1) * V01 ** 2 - u ** 2)
    return rho, rho * (V_INF + u), np.zeros_like(x), rho * (e + .5 * (V_INF + u) ** 2)


def run(N=2, K1D=32, T=0.2, CFL=0.05, Ky=None, verbose=True):
    Kx = int(K1D / 2 * 3)
    Ky = K1D if Ky is None else Ky
    VX, VY, EToV = sd.uniform_quad_mesh(Kx, Ky)
    VX, VY = VX / 4 * 3 + 1 / 4, (VY + 1) / 2                         # :63-65
    rd = sd.init_reference_quad(N)
    md = sd.init_mesh((VX, VY), EToV, rd)
    mapB = md.mapB.copy()
    sd.make_periodic(md, rd)                                          # :72-78
    xb = md.xf.flatten(order="F")[mapB - 1]
    md.mapB = mapB[(np.abs(xb + .5) < 1e-12) | (np.abs(xb - 1.0) < 1e-12)]   # leftwall / rightwall :165-166
    ops = sd.cns_ops(rd)
    sd.interp_geofacs_to_hybrid(md, ops["Vh"])
    Q = exact_sol_viscous_shocktube(md.x, 0.0)                        # :581-582
    CN = (N + 1) * (N + 2) / 2
    dt0 = CFL * (2 / K1D) / CN                                        # :64-69
    eng = engine.RhsEngine(rd, md, ops, engine.CNS_MODAL, BCTYPE=4, inviscid_dissp=True, viscous_dissp=False, Re=10000.0,
                           mu=MU, lam=LAM, Pr=PR, inflow=(RHOL, UL, 0.0, PL))
    Qd = eng.upload(Q)
    integ = timestep.Dopri45(eng, Qd, dt0, swap=True)       # (integ.Q is the state; an accepted step swaps buffers)
    while integ.t < T:
        ok, err = integ.step()
        if verbose and integ.i % 20 == 0:
            print(f"i = {integ.i}, t = {integ.t}, dt = {integ.dt}, errEst = {err}")
    # errors against the exact solution at the final time, on the device: L1err / Linferr of :745-771 and an L2 error
    # with the (N+2) Gauss rule as in the Euler driver
    par = (V0, V1, V01, M0, KAPPA / M0 / CV, V_INF)
    Vq2, wq2 = sd.error_quadrature(N)
    eng.setup_errors(rd, md, Vq2, wq2)
    L1, Linf, _ = eng.nodal_error(integ.Q, integ.t, exact=1, par=par)
    L2, _ = eng.l2_error(integ.Q, integ.t, exact=1, par=par)
    if verbose:
        print(f"N = {N}, K = {md.K}\nL1 error is {L1}\nLinf error is {Linf}")
        print(f"t = {integ.t}: L2 error {L2:.3e} ({integ.i} attempted steps, {integ.n_rhs} RHS evaluations)")
    return L2, Linf, integ


if __name__ == "__main__":
    a = sys.argv[1:]
    run(int(a[0]) if a else 2, int(a[1]) if len(a) > 1 else 32, float(a[2]) if len(a) > 2 else 0.2)
