"""Generates the committed golden fixtures from the ORACLE (the reference is Julia and cannot run in
this pipeline; it stores no expected values of its own for the RHS -- SURVEY.md section 4):

  tests/golden/maps_2x2_N1.npz   mapM/mapP/mapB/FToF of init_mesh + periodic patch, hand-checked below
  tests/golden/maps_3x2_N2.npz
  tests/golden/rhs_euler_N2_3x3.npz, rhs_cns_N2_3x3.npz   state + oracle RHS (C restatement)

  python tests/golden/make_fixtures.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from oracle import oracle as orc          # noqa: E402
from oracle import ref_setup as rs        # noqa: E402

for (Kx, Ky, N) in ((2, 2, 1), (3, 2, 2)):
    VX, VY, EToV = rs.uniform_quad_mesh(Kx, Ky)
    rd = rs.init_reference_quad(N)
    md = rs.init_mesh_2D(VX, VY, EToV, rd)
    mapP0 = md.mapP.copy()
    rs.make_periodic_2D(md, rd, VX, VY)
    np.savez(os.path.join(HERE, f"maps_{Kx}x{Ky}_N{N}.npz"), EToV=EToV, FToF=md.FToF, mapM=md.mapM, mapP_walls=mapP0,
             mapP_periodic=md.mapP, mapB=md.mapB)

p = orc.build_euler_problem(2, 3, 3)
out, rt = orc.EulerOracle(p).rhs(p.Q, .5, True)
np.savez(os.path.join(HERE, "rhs_euler_N2_3x3.npz"), Q=np.stack(p.Q), rhs=np.stack(out), rhstest=rt)
p = orc.build_cns_problem(2, 3, 3, bc="periodic")
out, rt, rtv = orc.CnsOracle(p).rhsRK(p.Q)
np.savez(os.path.join(HERE, "rhs_cns_N2_3x3.npz"), Q=np.stack(p.Q), rhs=np.stack(out), rhstest=rt, rhstest_visc=rtv)
print("fixtures written")

# ---- hexahedra (examples/dg3D_euler_hex.jl) ---------------------------------------------------------------
VX, VY, VZ, EToV = rs.uniform_hex_mesh(2, 2, 2)
rd = rs.init_reference_hex(1)
md = rs.init_mesh_3D(VX, VY, VZ, EToV, rd)
mapP0 = md.mapP.copy()
rs.make_periodic_3D(md, rd)
np.savez(os.path.join(HERE, "maps_hex_2x2x2_N1.npz"), EToV=EToV, FToF=md.FToF, mapM=md.mapM, mapP_walls=mapP0,
         mapP_periodic=md.mapP, mapB=md.mapB)
p = orc.build_hex_problem(2, 2, 2, 2)
for lf, tag in ((0.0, "lf0"), (0.25, "lf025")):
    out, rt = orc.HexOracle(p, lf).rhs(p.Q, True)
    np.savez(os.path.join(HERE, f"rhs_hex_N2_2x2x2_{tag}.npz"), Q=np.stack(p.Q), rhs=np.stack(out), rhstest=rt)
print("hex fixtures written")
