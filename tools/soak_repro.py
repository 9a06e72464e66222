#!/usr/bin/env python3
"""Run-to-run reproducibility soak on the GPU: python tools/soak_repro.py (from the repo root)."""
import sys, os
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
import torch, numpy as np
from common import product_cns_problem, product_euler_problem
from esdg_cns_amd import engine as E
# (the Euler cases and CNS N=3 run the instantiations of kt2_rhs that use their accumulator plane sets twice, late round 3)
for form, N, Kx in (("cns", 4, 512), ("cns", 5, 256), ("cns", 3, 512), ("euler", 4, 512), ("euler", 3, 509), ("euler", 4, 203)):
    rd, md, ops, Q = (product_cns_problem if form == "cns" else product_euler_problem)(N, Kx, Kx)
    eng = E.RhsEngine(rd, md, ops, E.CNS_MODAL if form == "cns" else E.EULER_COLLOCATED)
    Qd = eng.upload(Q)
    ref = eng.rhs(Qd).clone()
    bad = 0
    for i in range(300):
        if not torch.equal(eng.rhs(Qd), ref):
            bad += 1
    print(f"{form} N={N} {Kx}x{Kx}: 300 repeated evaluations, {bad} differ from the first", flush=True)
    assert bad == 0
    del eng
