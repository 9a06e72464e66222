#!/usr/bin/env python3
"""Transliteration of the reference driver examples/dg3D_euler_hex.jl onto the MI355X engine.

The script evaluates `rhs` once on a random state and prints the entropy-production diagnostic (`@show rhstest`,
:224-226, "for testing EC"); its LSRK45 loop and L2-error block are commented out (:228-262).  Both are provided:

  python examples/dg3D_euler_hex.py [N] [K1D]            # one RHS + rhstest on the script's random state
  python examples/dg3D_euler_hex.py [N] [K1D] wave [T]   # the commented-out time loop on a density wave

The density wave rho = 2 + .5 sin(pi (y - t)), (u,v,w) = (0,1,0), p = 1 is an exact solution (the script's commented
`rhoex` uses x - t with v = 1, which is not).  LF factor 0 as in the script (:193).
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from esdg_cns_amd import engine, physics as ph, setup_dg as sd, timestep  # noqa: E402


def setup(N, K1D):
    VX, VY, VZ, EToV = sd.uniform_hex_mesh(K1D, K1D, K1D)            # :26
    rd = sd.init_reference_hex(N, sd.gauss_quad(0, 0, N))             # :31
    md = sd.init_mesh_3d((VX, VY, VZ), EToV, rd)                      # :32
    ops = sd.hex_ops(rd)                                              # :34-56, 92-98
    sd.make_periodic_3d(md, rd)                                       # :58-65
    sd.hex_driver_geometry(md, rd)                                    # :75-98 with a = 0
    return rd, md, ops


def run_rhstest(N=2, K1D=8, seed=0, verbose=True):
    rd, md, ops = setup(N, K1D)
    rng = np.random.default_rng(seed)
    shp = md.xq.shape
    rho = 2 + .1 * rng.random(shp)                                    # :102-108
    u, v, w = np.zeros(shp), np.ones(shp), np.zeros(shp)
    p = np.ones(shp) + .1 * rng.random(shp)
    Q = ph.primitive_to_conservative_3d(rho, u, v, w, p)
    rhsQ, rhstest = engine.rhs_hex(Q, md, ops, None, True, rd=rd)     # :224-225
    if verbose:
        print(f"rhstest = {rhstest}")
    return rhstest


def run_wave(N=2, K1D=8, T=1 / 3, CFL=.5, verbose=True):
    rd, md, ops = setup(N, K1D)
    rhoex = lambda x, y, z, t: 2 + .5 * np.sin(np.pi * (y - t))
    shp = md.xq.shape
    Q = ph.primitive_to_conservative_3d(rhoex(md.xq, md.yq, md.zq, 0), np.zeros(shp), np.ones(shp), np.zeros(shp), np.ones(shp))
    CN = (N + 1) * (N + 2) * 3 / 2                                    # :113-117
    dt = CFL * 2 / (CN * K1D)
    Nsteps = int(np.ceil(T / dt))
    dt = T / Nsteps
    eng = engine.RhsEngine(rd, md, ops, engine.EULER_HEX_COLLOCATED, lf_scale=0.0)
    Qd = eng.upload(Q)
    rhstest = timestep.lsrk45_run(eng, Qd, dt, Nsteps, rhstest_every=10)   # :230-244
    rho = eng.download(Qd)[0]
    L2err = np.sqrt(np.sum(np.abs(md.wJq) * (rho - rhoex(md.xq, md.yq, md.zq, T)) ** 2))   # :255-256 (abs: J < 0)
    if verbose:
        print(f"Time step: {Nsteps} out of {Nsteps} with rhstest = {rhstest}")
        print(f"L2err = {L2err}")
    return L2err, rhstest


if __name__ == "__main__":
    a = sys.argv[1:]
    N = int(a[0]) if a else 2
    K1D = int(a[1]) if len(a) > 1 else 8
    if len(a) > 2 and a[2] == "wave":
        run_wave(N, K1D, float(a[3]) if len(a) > 3 else 1 / 3)
    else:
        run_rhstest(N, K1D)
