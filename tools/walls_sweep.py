"""Wall instantiation of kt3_rhs against kt2_rhs (ESDG_V2=rhs) on the lid-driven cavity, degrees 1..6, BCTYPE 1..3, and on the
shock-tube closures (BCTYPE 4) where the test helpers provide them: relative difference of the two results (two mappings of the
same formulas: round-off apart).   python tools/walls_sweep.py"""
import os, sys
os.environ.setdefault("ESDG_HIP_LIB", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "esdg_cns_amd", "libesdg_hip_ab.so"))   # the A/B build reads the ESDG_* switches; the shipped library reads none
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch
from common import product_cavity_problem
from esdg_cns_amd import engine as E

worst = 0.0
for N in range(1, 7):
    for BCTYPE in (1, 2, 3):
        rd, md, ops, Q = product_cavity_problem(N, 11, 7)
        outs = []
        for env in (None, "rhs"):
            if env: os.environ["ESDG_V2"] = env
            else: os.environ.pop("ESDG_V2", None)
            eng = E.RhsEngine(rd, md, ops, E.CNS_MODAL, BCTYPE=BCTYPE)
            outs.append(np.stack(eng.download(eng.rhs(eng.upload(Q)))))
            for parts in (1, 2):                       # rhs_inviscid! / rhs_viscous! alone
                E.check(eng.L.esdg_set_parts(eng.ctx, parts))
                outs.append(np.stack(eng.download(eng.rhs(eng.upload(Q)))))
            E.check(eng.L.esdg_set_parts(eng.ctx, 3))
            del eng
        os.environ.pop("ESDG_V2", None)
        d = [float(np.linalg.norm(outs[i] - outs[i + 3]) / np.linalg.norm(outs[i + 3])) for i in range(3)]
        worst = max(worst, *d)
        print(f"cavity N={N} BCTYPE={BCTYPE} 11x7: |kt3 - kt2| / |kt2|  rhsRK! {d[0]:.2e}  inviscid {d[1]:.2e}  viscous {d[2]:.2e}", flush=True)
print(f"worst {worst:.2e}")
assert worst < 1e-11
