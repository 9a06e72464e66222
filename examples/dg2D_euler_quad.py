#!/usr/bin/env python3
"""Transliteration of the reference driver examples/dg2D_euler_quad.jl onto the MI355X engine: same set-up calls,
same LSRK45 loop, same L2-error functional; the inline `rhs` of the script is replaced by the device RHS.

  python examples/dg2D_euler_quad.py [N] [K1D] [T]
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from esdg_cns_amd import engine, physics as ph, setup_dg as sd, timestep  # noqa: E402


def run(N=2, K1D=12, T=1.0, CFL=2.0, verbose=True):
    # "Mesh related variables" (dg2D_euler_quad.jl:26-31)
    Kx, Ky = int(4 / 3 * K1D), K1D
    VX, VY, EToV = sd.uniform_quad_mesh(Kx, Ky)
    VX, VY = 15 * (1 + VX) / 2, 5 * VY
    rd = sd.init_reference_quad(N, sd.gauss_quad(0, 0, N))          # :35
    md = sd.init_mesh((VX, VY), EToV, rd)                            # :36
    sd.make_periodic(md, rd)                                         # :38-44
    ops = sd.euler_quad_ops(rd)                                      # :47-78
    rho, u, v, p = ph.vortex(md.xq, md.yq, 0)                        # :81-83
    Q = ph.primitive_to_conservative(rho, u, v, p)
    sd.interp_geofacs_to_hybrid(md, ops["Vh"])                       # :86-88
    # "Time integration" (:93-99)
    CN = (N + 1) * (N + 2) / 2
    h = 2 / K1D
    dt = CFL * h / CN
    Nsteps = int(np.ceil(T / dt))
    dt = T / Nsteps
    eng = engine.RhsEngine(rd, md, ops, engine.EULER_COLLOCATED)
    Qd = eng.upload(Q)
    rhstest = timestep.lsrk45_run(eng, Qd, dt, Nsteps, rhstest_every=10)   # :196-212
    if verbose:
        print(f"Time step: {Nsteps} out of {Nsteps} with rhstest = {rhstest}")
    # "project solution back to GLL nodes" and error with an N+2 Gauss rule (:214-233), evaluated on the device
    Vq2, wq2 = sd.error_quadrature(N)
    eng.setup_errors(rd, md, Vq2, wq2)
    L2err, _ = eng.l2_error(Qd, T)
    if verbose:
        print(f"L2err at final time T = {T} is {L2err}\n")
    return L2err, rhstest


if __name__ == "__main__":
    a = sys.argv[1:]
    run(int(a[0]) if a else 2, int(a[1]) if len(a) > 1 else 12, float(a[2]) if len(a) > 2 else 1.0)
