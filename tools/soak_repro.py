#!/usr/bin/env python3
"""Run-to-run reproducibility soak on the GPU: python tools/soak_repro.py (from the repo root)."""
import sys, os
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
import torch, numpy as np
from common import product_cns_problem
from esdg_cns_amd import engine as E
for N, Kx in ((4, 512), (5, 256), (3, 512)):
    rd, md, ops, Q = product_cns_problem(N, Kx, Kx)
    eng = E.RhsEngine(rd, md, ops, E.CNS_MODAL)
    Qd = eng.upload(Q)
    ref = eng.rhs(Qd).clone()
    bad = 0
    for i in range(300):
        if not torch.equal(eng.rhs(Qd), ref):
            bad += 1
    print(f"N={N} {Kx}x{Kx}: 300 repeated evaluations, {bad} differ from the first", flush=True)
    assert bad == 0
    del eng
