#!/usr/bin/env python3
"""Host-side cost of driving one sharded RHS evaluation (the launches of RhsEngine._phases without the transport):
issue time per evaluation vs GPU time per evaluation, rank 0 of a 2-rank split of a 512x1024 mesh.  If the host issues
faster than the GPU executes, the Python driver is not the bottleneck of a multi-GPU run.
  python tools/host_overhead.py"""
import ctypes as C
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from common import product_cns_problem  # noqa: E402
from esdg_cns_amd import engine as E  # noqa: E402

N, Kx, Ky = 4, 512, 1024
offs = np.array([0, Kx * Ky // 2, Kx * Ky], dtype=np.int64)
rd, md, ops, Q = product_cns_problem(N, Kx, Ky, elem_range=(0, int(offs[1])))
eng = E.RhsEngine(rd, md, ops, E.CNS_MODAL, rank=0, nranks=2, rank_offsets=offs)
Qd, out = eng.upload(Q), eng.new_state()
L, ctx = eng.L, eng.ctx
q, o = C.c_void_p(Qd.data_ptr()), C.c_void_p(out.data_ptr())
lo, hi = eng.interior
K = eng.K


def one_rhs():
    s = eng._stream()
    for ph in range(eng.nphases):
        if ph > 0:
            E.check(L.esdg_rhs_phase_range(ctx, ph, lo, hi - lo, q, o, s))
        E.check(L.esdg_rhs_phase_range(ctx, ph, 0, lo, q, o, s))
        E.check(L.esdg_rhs_phase_range(ctx, ph, hi, K - hi, q, o, s))
        for x, (a, b, _) in enumerate(eng.xinfo):
            if a == ph:
                E.check(L.esdg_halo_pack(ctx, x, s))
        if ph == 0:
            E.check(L.esdg_rhs_phase_range(ctx, ph, lo, hi - lo, q, o, s))


for _ in range(200):
    one_rhs()
torch.cuda.synchronize()
# issue time: short bursts into an empty queue (a long burst measures back-pressure from the GPU, not the host)
n_burst, t_issue = 20, 0.0
for _ in range(10):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n_burst):
        one_rhs()
    t_issue += time.perf_counter() - t0
t_issue /= 10 * n_burst
torch.cuda.synchronize()
n = 300
t0 = time.perf_counter()
for _ in range(n):
    one_rhs()
torch.cuda.synchronize()
t_all = (time.perf_counter() - t0)
t_issue *= n
print(f"interior [{lo},{hi}) of {K}; per RHS: host issue {t_issue / n * 1e6:.0f} us, GPU {t_all / n * 1e6:.0f} us "
      f"({eng.nphases * 3 + 1} kernel launches + {len(eng.xinfo)} packs through ctypes)")
