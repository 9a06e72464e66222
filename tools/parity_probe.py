"""Viscous part (rhs_viscous!) on the periodic vortex box: tensor kernels (neighbour entropy variables rebuilt from the
trace state) vs generic kernels (interpolated entropy variables carried in A_v), each against the binary128 truth."""
import os
os.environ.setdefault("ESDG_HIP_LIB", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "esdg_cns_amd", "libesdg_hip_ab.so"))   # the A/B build reads the ESDG_* switches; the shipped library reads none
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from esdg_cns_amd import engine  # noqa: E402
from oracle import oracle as orc  # noqa: E402

orc.lib_quad().oracle_set_threads(orc.lib_quad().oracle_get_max_threads())
pf = lambda a, t: " ".join("%.1e" % (np.linalg.norm(x - y) / max(np.linalg.norm(y), 1e-300)) for x, y in zip(a, t))
for N, Kx, Ky in ((4, 12, 8), (3, 10, 10)):
    p = orc.build_cns_problem(N, Kx, Ky)
    o, q = orc.CnsOracle(p), orc.CnsOracle(p, quad=True)
    tt, tv = q.rhsRK(p.Q, False)[0], q.rhs_viscous(p.Q)[0]
    print(f"N={N} oracle   total {pf(o.rhsRK(p.Q, False)[0], tt)} | viscous {pf(o.rhs_viscous(p.Q)[0][1:], tv[1:])}")
    for tag in ("tensor", "generic"):
        if tag == "generic":
            os.environ["ESDG_FORCE_GENERIC"] = "1"
        else:
            os.environ.pop("ESDG_FORCE_GENERIC", None)
        eng = engine.RhsEngine(p.rd, p.md, p.ops, engine.CNS_MODAL, Re=p.Re, mu=p.mu, lam=p.lam, Pr=p.Pr)
        inv = engine.RhsEngine(p.rd, p.md, p.ops, engine.EULER_MODAL)
        Qd = eng.upload(p.Q)
        tot = eng.download(eng.rhs(Qd))
        vis = [a - b for a, b in zip(tot, inv.download(inv.rhs(Qd)))]
        print(f"N={N} {tag:8s} total {pf(tot, tt)} | viscous (total - inviscid) {pf(vis[1:], tv[1:])}")
