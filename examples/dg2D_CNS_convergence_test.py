#!/usr/bin/env python3
"""The parameter sweep of examples/CompressibleNS/dg2D_CNS_convergence_test.jl on the MI355X engine with the reference
*quad* element (the script itself sweeps triangles, which this build does not cover): lid-driven cavity with the smooth
lid profile vlid = (1+cos(pi*x))/2 (:72-76), fluid at rest at Ma = .3 (:924-930), adaptive DOPRI45 to T (:964-1053), then
the boundary-velocity error (:1055-1080) evaluated on the device.

  python examples/dg2D_CNS_convergence_test.py [T] [K1D,K1D,...] [N,N,...]

Every run prints the error both as Julia executes :1075-1078 (the u_2 term only; see include/esdg_hip.h) and as the
statement reads (all three terms).
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from esdg_cns_amd import engine, physics as ph, setup_dg as sd, timestep  # noqa: E402


def vlid(x):
    return (1 + np.cos(np.pi * x)) / 2                               # :76


def run_one(N, K1D, Re=100.0, T=1.0, CFL=0.01, BCTYPE=1, inviscid_dissp=True, viscous_dissp=True, verbose=False):
    mu, lam, Pr = 1 / Re, -2 / 3 / Re, .71                           # :866-870
    CN = (N + 1) * (N + 2) / 2
    dt = CFL * (2 / K1D) / CN                                        # :874-879
    Nsteps = int(np.ceil(T / dt))
    dt0 = T / Nsteps
    VX, VY, EToV = sd.uniform_quad_mesh(K1D, K1D)
    rd = sd.init_reference_quad(N)
    md = sd.init_mesh((VX, VY), EToV, rd)
    ops = sd.cns_ops(rd)                                             # :890-918
    sd.interp_geofacs_to_hybrid(md, ops["Vh"])
    x = md.x
    rho, u, v = np.ones_like(x), np.zeros_like(x), np.zeros_like(x)  # :924-930
    p = (1 / (.3 ** 2 * ph.GAMMA)) * np.ones_like(x)
    eng = engine.RhsEngine(rd, md, ops, engine.CNS_MODAL, Re=Re, mu=mu, lam=lam, Pr=Pr, BCTYPE=BCTYPE,
                           inviscid_dissp=inviscid_dissp, viscous_dissp=viscous_dissp, vlid=vlid)
    eng.setup_errors(rd, md, boundary=True)
    Qd = eng.upload(ph.primitive_to_conservative(rho, u, v, p))
    integ = timestep.Dopri45(eng, Qd, dt0, swap=True)       # (integ.Q is the state; an accepted step swaps buffers)
    while integ.t < T:
        ok, err = integ.step()
        if verbose and integ.i % 5 == 0:                             # interval = 5, :977
            print(f"i = {integ.i}, t = {integ.t}, dt = {integ.dt}, errEst = {err}")
    executed, written, _ = eng.boundary_velocity_error(integ.Q, 2.0 / K1D)  # Jf = 2.0/K1D, :1073
    return executed, written, integ


def main(T=1.0, K1D_arr=(4, 8, 16), N_arr=(1, 2, 3, 4), Re_arr=(100.0,)):
    err_arr = np.zeros((len(K1D_arr), len(N_arr), len(Re_arr), 2))
    for n_idx, N in enumerate(N_arr):
        for k_idx, K1D in enumerate(K1D_arr):
            for r_idx, Re in enumerate(Re_arr):
                print(f"========= K:{K1D} N:{N}")
                ex, wr, integ = run_one(N, K1D, Re=Re, T=T)
                err_arr[k_idx, n_idx, r_idx] = ex, wr
                print(f"err = {ex}   (all three terms: {wr}; {integ.i} attempted steps, {integ.n_rhs} RHS evaluations)")
    np.savetxt("err_arr.txt", err_arr.reshape(-1, 2))                # :1086-1088
    return err_arr


if __name__ == "__main__":
    a = sys.argv[1:]
    main(float(a[0]) if a else 1.0,
         tuple(int(k) for k in a[1].split(",")) if len(a) > 1 else (4, 8, 16),
         tuple(int(n) for n in a[2].split(",")) if len(a) > 2 else (1, 2, 3, 4))
