"""Repeated evaluations on lid-driven-cavity meshes compared bit for bit with the first (the wall instantiations reuse LDS planes
inside group-uniform branches: a missing barrier would show as a result that depends on wave timing).
  python tools/soak_walls.py [repeats]"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from common import product_cavity_problem  # noqa: E402
from esdg_cns_amd import engine as E  # noqa: E402

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 300
for N, Kx, BCTYPE in ((4, 64, 1), (4, 256, 1), (3, 48, 2), (5, 24, 3), (2, 40, 1), (6, 16, 1), (8, 12, 1), (9, 12, 1), (11, 8, 3)):   # (N = 9, 11: kt3_rhs wall form, one wave per SIMD)
    rd, md, ops, Q = product_cavity_problem(N, Kx, Kx)
    eng = E.RhsEngine(rd, md, ops, E.CNS_MODAL, BCTYPE=BCTYPE)
    Qd = eng.upload(Q)
    first = eng.rhs(Qd).clone()
    bad = 0
    for _ in range(reps):
        bad += int(not torch.equal(eng.rhs(Qd), first))
    print(f"cavity N={N} {Kx}x{Kx} BCTYPE={BCTYPE}: {reps} repeated evaluations, {bad} differ from the first")
