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
n = 20
q, o = C.c_void_p(Qd.data_ptr()), C.c_void_p(out.data_ptr())
for ph in range(eng.nphases):      # one phase (= one kernel) at a time: the stamp slots are shared by the kernels
    L.esdg_debug_stamps(buf, 1)
    for _ in range(n):
        _lib.check(eng.L.esdg_rhs_phase(eng.ctx, ph, q, o, None))
    torch.cuda.synchronize()
    L.esdg_debug_stamps(buf, 1)
    v = np.array(list(buf), dtype=float)
    tot = v.sum()
    if tot:
        print(f"phase {ph}: wave cycles by stamp:", " ".join(f"{i}:{x / tot * 100:.1f}%" for i, x in enumerate(v) if x),
              f"| sum per launch {tot / n:.3e}")
