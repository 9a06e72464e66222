#!/usr/bin/env python3
"""The time loop of examples/CompressibleNS/dg2D_CNS_cavity_optimized.jl (adaptive DOPRI45 around rhsRK!) on the
MI355X engine with the reference *quad* element: lid-driven cavity (walls, BCTYPE 1/2/3) or the periodic vortex box.

  python examples/dg2D_CNS_quad.py [cavity|periodic] [N] [K1D] [T] [BCTYPE]
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from esdg_cns_amd import engine, physics as ph, setup_dg as sd, timestep  # noqa: E402


def run(case="cavity", N=3, K1D=16, T=0.1, BCTYPE=2, Re=1000.0, CFL=0.5, verbose=True):
    mu, lam, Pr = 1 / Re, -2 / 3 / Re, .71                           # cavity_optimized.jl:33-36
    VX, VY, EToV = sd.uniform_quad_mesh(K1D, K1D)
    if case == "periodic":
        VX, VY = 15 * (1 + VX) / 2, 5 * VY
    rd = sd.init_reference_quad(N)
    md = sd.init_mesh((VX, VY), EToV, rd)
    if case == "periodic":
        sd.make_periodic(md, rd)
    ops = sd.cns_ops(rd)                                             # :62-90
    sd.interp_geofacs_to_hybrid(md, ops["Vh"])
    x, y = md.x, md.y
    if case == "periodic":
        rho, u, v, p = ph.vortex(x, y, 0)
    else:                                                            # :855-861: fluid at rest, Ma = .3
        rho, u, v = np.ones_like(x), np.zeros_like(x), np.zeros_like(x)
        p = (1 / (.3 ** 2 * ph.GAMMA)) * np.ones_like(x)
    Q = ph.primitive_to_conservative(rho, u, v, p)
    CN = (N + 1) * (N + 2) / 2
    dt0 = CFL * (2 / K1D) / CN                                       # :39-45
    eng = engine.RhsEngine(rd, md, ops, engine.CNS_MODAL, Re=Re, mu=mu, lam=lam, Pr=Pr, BCTYPE=BCTYPE,
                           inviscid_dissp=True, viscous_dissp=True)
    Qd = eng.upload(Q)
    integ = timestep.Dopri45(eng, Qd, dt0, swap=True)       # (integ.Q is the state; an accepted step swaps buffers)
    while integ.t < T:
        ok, err = integ.step()
        if verbose and integ.i % 5 == 0:
            print(f"i = {integ.i}, t = {integ.t}, dt = {integ.dt}, errEst = {err}")
    if verbose:
        print(f"done: t = {integ.t}, {integ.i} attempted steps, {integ.n_rhs} RHS evaluations")
    return eng.download(integ.Q), integ


if __name__ == "__main__":
    a = sys.argv[1:]
    run(a[0] if a else "cavity", int(a[1]) if len(a) > 1 else 3, int(a[2]) if len(a) > 2 else 16,
        float(a[3]) if len(a) > 3 else 0.1, int(a[4]) if len(a) > 4 else 2)
