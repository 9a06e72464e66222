"""rhs_viscous! on the cavity, CPU only: the Float64 oracle with the per-node J / metric arrays of the viscous operators
(dg_grad!, dg_div!: rows 1:Np of rxJ ..., J at every node, dg2D_CNS_cavity_optimized.jl:549-611) replaced by their element
means, against the binary128 truth on the raw arrays -- the attribution of the GPU's excess in `rhs_viscous!` alone on wall
meshes (the kernels hold one geometry record per element).   python tools/cavity_visc_attribution.py N Kx Ky"""
import copy
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from common import as_oracle_problem, product_cavity_problem
from oracle import oracle as orc
N,Kx,Ky = (int(a) for a in sys.argv[1:4])
PHYS = dict(Re=1000.0, mu=1e-3, lam=-2e-3 / 3, Pr=.71, BCTYPE=1)
orc.lib_quad().oracle_set_threads(orc.lib_quad().oracle_get_max_threads())
rd, md, ops, Q = product_cavity_problem(N, Kx, Ky)
p = as_oracle_problem(rd, md, ops, Q, **PHYS)
q = orc.CnsOracle(p, quad=True)
tv = q.rhs_viscous(Q)[0]
tt = q.rhsRK(Q, False)[0]
o = orc.CnsOracle(p)
rel = lambda a, t: " ".join("%.2e" % (np.linalg.norm(x - y) / np.linalg.norm(y)) for x, y in zip(a[1:], t[1:]))
print("oracle raw arrays          : viscous", rel(o.rhs_viscous(Q)[0], tv), "| rhsRK", rel(o.rhsRK(Q, False)[0], tt))
Np = rd.Pq.shape[0]
for which in (("J",), ("rxJ","sxJ","ryJ","syJ"), ("J","rxJ","sxJ","ryJ","syJ")):
    md2 = copy.copy(md)
    for nm in which:
        a = getattr(md, nm).copy()
        if nm == "J":
            a[:] = a.mean(axis=0)
        else:
            a[:Np] = a[:Np].mean(axis=0)      # the viscous operators use rows 1:Np (dg_grad! :552); flux differencing row 1
        setattr(md2, nm, a)
    p2 = as_oracle_problem(rd, md2, ops, Q, **PHYS)
    o2 = orc.CnsOracle(p2)
    print(f"oracle, element means of {'+'.join(which):22s}: viscous", rel(o2.rhs_viscous(Q)[0], tv),
          "| rhsRK", rel(o2.rhsRK(Q, False)[0], tt) if which == ("J",) else "n/a (flux differencing reads row 1 of the metric arrays)")
