"""Where does the cavity state's e_gpu / e_orc of 1.6-2.2 at N >= 8 come from?  (Round 5.)
Not from the kernels' arithmetic: builds with IEEE divisions and the library logarithm (-DESDG_IEEE_DIV -DESDG_LIBM_LOG), with corrected
quotients in the primitives and in the flux's pressure term give the same figures to two digits.  It is the REPRESENTATION of the
operators.  The binary128 truth evaluates the dense arrays the driver holds exactly, their set-up round-off included (VhP = Vh * Pq:
entries that are zero or one mathematically carry 1e-16 ... 2e-15 of noise); the kernels apply the same operators from 1D tables
(sum factorisation; identity at the volume nodes, N1 weights per face node).  On this low-Mach state the mass and momentum rows
cancel to a few per cent of their terms, and a perturbation of ONE ULP per entry of VhP moves them by as much as the Float64
reference's whole rounding error (last lines of the output) -- growing with N like the conditioning of the LGL -> Gauss change of basis.
The probe reports, per field: the kernels and the oracle against the truth; the same with VhP[1:Nq, :] set to the exact identity;
and the truth's own sensitivity to VhP * (1 + 1.1e-16 xi), xi uniform in [-1, 1).
    python tools/lowmach_probe.py [N Kx Ky]"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from common import as_oracle_problem, cavity_state, product_cavity_problem, product_cns_problem
from oracle import oracle as orc
from esdg_cns_amd import engine

PHYS = dict(Re=1000.0, mu=1e-3, lam=-2e-3 / 3, Pr=.71, BCTYPE=1)
N, Kx, Ky = (int(a) for a in sys.argv[1:4]) if len(sys.argv) > 3 else (9, 3, 2)


def per_field(a, b):
    return [float(np.linalg.norm(np.asarray(x) - np.asarray(y, float)) / np.linalg.norm(np.asarray(y, float))) for x, y in zip(a, b)]


def fmt(v):
    return "max %.2e  [" % max(v) + " ".join("%.1e" % x for x in v) + "]"


for mesh in ("periodic", "walls"):
    if mesh == "periodic":
        rd, md, ops, _ = product_cns_problem(N, Kx, Ky)
        Q = cavity_state(2 * md.x / 15 - 1, md.y / 5)     # (the vortex box is [0,15] x [-5,5])
        kw = {}
    else:
        rd, md, ops, Q = product_cavity_problem(N, Kx, Ky)
        kw = {"BCTYPE": 1}
    p = as_oracle_problem(rd, md, ops, Q, **PHYS)
    VhP = np.array(ops["VhP"], copy=True)
    Nq = VhP.shape[1]
    dev = np.abs(VhP[:Nq] - np.eye(Nq)).max()
    VhP[:Nq] = np.eye(Nq)
    pI = as_oracle_problem(rd, md, dict(ops, VhP=np.asfortranarray(VhP)), Q, **PHYS)
    eng = engine.RhsEngine(rd, md, ops, engine.CNS_MODAL, Re=p.Re, mu=p.mu, lam=p.lam, Pr=p.Pr, **kw)
    g = eng.download(eng.rhs(eng.upload(Q)))
    print(f"{mesh} N={N} {Kx}x{Ky} rhsRK!   max |VhP[1:Nq,:] - I| = {dev:.2e}")
    for name, pp in (("arrays as the driver holds them", p), ("VhP[1:Nq,:] = I exactly", pI)):
        r, t = orc.CnsOracle(pp).rhsRK(Q, False)[0], orc.CnsOracle(pp, quad=True).rhsRK(Q, False)[0]
        eg, eo = per_field(g, t), per_field(r, t)
        print(f"  {name:32s} e_gpu {fmt(eg)}\n  {'':32s} e_orc {fmt(eo)}   ratio of the maxima {max(eg) / max(eo):.2f}")
    t = orc.CnsOracle(p, quad=True).rhsRK(Q, False)[0]
    A = np.array(ops["VhP"])
    pU = as_oracle_problem(rd, md, dict(ops, VhP=np.asfortranarray(A * (1 + 1.1e-16 * (2 * np.random.default_rng(1).random(A.shape) - 1)))), Q, **PHYS)
    print(f"  {'truth(VhP perturbed by one ulp) vs truth':43s} {fmt(per_field(orc.CnsOracle(pU, quad=True).rhsRK(Q, False)[0], t))}")
