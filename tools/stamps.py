"""Per-phase wave cycles of a -DESDG_T2_STAMP build (esdg_cns_amd/variants/stamp.so): shares of one kernel's iteration.
   ESDG_HIP_LIB=$PWD/esdg_cns_amd/variants/stamp.so python tools/stamps.py"""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
import torch  # noqa: E402
from esdg_cns_amd import _lib, engine  # noqa: E402

rd, md, ops, Q = bench.build_problem(4, 512, 512, 0, 512 * 512, "cns")
eng = engine.RhsEngine(rd, md, ops, engine.CNS_MODAL)
L = C.CDLL(_lib.LIB_PATH)
Qd, out = eng.upload(Q), eng.new_state()
for _ in range(50):
    eng.rhs_into(Qd, out)
torch.cuda.synchronize()
buf = (C.c_ulonglong * 16)()
L.esdg_debug_stamps(buf, 1)
n = 20
for _ in range(n):
    eng.rhs_into(Qd, out)
torch.cuda.synchronize()
L.esdg_debug_stamps(buf, 1)
v = np.array(list(buf), dtype=float)
tot = v.sum()
print("cycles per iteration (wave 0 of each workgroup), by stamp:", " ".join(f"{i}:{x / tot * 100:.1f}%" for i, x in enumerate(v) if x))
print("sum per launch (cycles x workgroups):", tot / n)
