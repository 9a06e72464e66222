"""rhs_viscous! ALONE (esdg_set_parts(2)) on the lid-driven cavity: where does the GPU's error against the binary128 truth sit?
Per field: relative L2 errors of the GPU and of the Float64 oracle, the share of the squared GPU error carried by elements that
touch a wall / the lid / a corner, and the worst elements.  Kernel sets: v2 (default: nodal-basis viscous operators in the elements with a boundary node), v2 with one geometry record
per element everywhere (ESDG_WALL_GEOMETRY=element).  (The round-1 kernels it also probed until round 4 are gone.)
  python tools/cavity_visc_probe.py [N Kx Ky [BCTYPE [nopen] [vlid0]]]     nopen: viscous_dissp = false; vlid0: lid velocity 0"""
import os
os.environ.setdefault("ESDG_HIP_LIB", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "esdg_cns_amd", "libesdg_hip_ab.so"))   # the A/B build reads the ESDG_* switches; the shipped library reads none
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from common import as_oracle_problem, product_cavity_problem  # noqa: E402
from esdg_cns_amd import engine  # noqa: E402
from oracle import oracle as orc  # noqa: E402

N, Kx, Ky = (int(a) for a in sys.argv[1:4]) if len(sys.argv) >= 4 else (4, 8, 8)
BCTYPE = int(sys.argv[4]) if len(sys.argv) > 4 else 1
PEN = "nopen" not in sys.argv
VL0 = "vlid0" in sys.argv
orc.lib_quad().oracle_set_threads(orc.lib_quad().oracle_get_max_threads())
PHYS = dict(Re=1000.0, mu=1e-3, lam=-2e-3 / 3, Pr=.71, BCTYPE=BCTYPE)
rd, md, ops, Q = product_cavity_problem(N, Kx, Ky)
p = as_oracle_problem(rd, md, ops, Q, **PHYS)
if VL0:
    p.vlid = lambda x: 0.0 * x
o, q = orc.CnsOracle(p, viscous_dissp=PEN), orc.CnsOracle(p, viscous_dissp=PEN, quad=True)
ov, tv = o.rhs_viscous(Q)[0], q.rhs_viscous(Q)[0]
K = Kx * Ky
ex, ey = np.arange(K) % Kx, np.arange(K) // Kx      # uniform_quad_mesh numbers elements x-fastest
wall = (ex == 0) | (ex == Kx - 1) | (ey == 0)
lid = ey == Ky - 1
kinds = dict(interior=~(wall | lid), wall=wall & ~lid, lid=lid)
print(f"cavity N={N} {Kx}x{Ky} BCTYPE={BCTYPE} penalty={PEN} vlid={'0' if VL0 else '1'}: rhs_viscous! alone, fields 2..4")
print("oracle  :", " ".join("%.2e" % (np.linalg.norm(a - t) / np.linalg.norm(t)) for a, t in zip(ov[1:], tv[1:])))
for tag, env in (("v2", {}), ("v2, one record per element", {"ESDG_WALL_GEOMETRY": "element"})):
    for k in ("ESDG_FORCE_GENERIC", "ESDG_WALL_GEOMETRY"):
        os.environ.pop(k, None)
    os.environ.update(env)
    eng = engine.RhsEngine(rd, md, ops, engine.CNS_MODAL, Re=p.Re, mu=p.mu, lam=p.lam, Pr=p.Pr, BCTYPE=BCTYPE, viscous_dissp=PEN,
                           **(dict(vlid=lambda x: 0.0 * x) if VL0 else {}))
    eng.set_parts(2)
    gv = eng.download(eng.rhs(eng.upload(Q)))
    print(f"{tag:26s}:", " ".join("%.2e" % (np.linalg.norm(a - t) / np.linalg.norm(t)) for a, t in zip(gv[1:], tv[1:])),
          "| gpu - oracle:", " ".join("%.2e" % (np.linalg.norm(a - b) / np.linalg.norm(t)) for a, b, t in zip(gv[1:], ov[1:], tv[1:])))
    for f in range(1, 4):
        eg = ((gv[f] - tv[f]) ** 2).sum(axis=0)          # per element
        eo = ((ov[f] - tv[f]) ** 2).sum(axis=0)
        shares = " ".join(f"{k} {eg[m].sum() / eg.sum():.2f} (oracle {eo[m].sum() / eo.sum():.2f})" for k, m in kinds.items())
        worst = np.argsort(eg)[::-1][:4]
        print(f"   field {f + 1}: share of squared error: {shares}; worst elements (x,y: gpu/oracle error ratio) " +
              " ".join(f"({ex[w]},{ey[w]}: {np.sqrt(eg[w] / max(eo[w], 1e-300)):.1f})" for w in worst))
        if tag == "v2" and f == 3:
            w = worst[0]
            d = gv[f][:, w] - tv[f][:, w]
            print("   worst element, field 4, node errors (gpu - truth):", np.array2string(d.reshape(N + 1, N + 1), precision=1, suppress_small=False, max_line_width=200))
            d = ov[f][:, w] - tv[f][:, w]
            print("   same element, oracle - truth:", np.array2string(d.reshape(N + 1, N + 1), precision=1, max_line_width=200))
