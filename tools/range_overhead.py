"""GPU-time cost of splitting every phase into boundary + interior launches (the halo-overlap schedule) on one GPU."""
import ctypes as C, sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from esdg_cns_amd import engine
from esdg_cns_amd._lib import check
rd, md, ops, Q = bench.build_problem(4, 512, 512, 0, 512 * 512, "cns")
eng = engine.RhsEngine(rd, md, ops, engine.CNS_MODAL)
Qd = eng.upload(Q); out = eng.new_state()
L, ctx = eng.L, eng.ctx
q, o = C.c_void_p(Qd.data_ptr()), C.c_void_p(out.data_ptr())
K, lo, hi = eng.K, 512, eng.K - 512
def full():
    check(L.esdg_rhs(ctx, q, o, eng._stream()))
def split():
    for ph in range(3):
        s = eng._stream()
        if ph == 0:
            check(L.esdg_rhs_phase_range(ctx, ph, 0, lo, q, o, s)); check(L.esdg_rhs_phase_range(ctx, ph, hi, K - hi, q, o, s))
            check(L.esdg_rhs_phase_range(ctx, ph, lo, hi - lo, q, o, s))
        else:
            check(L.esdg_rhs_phase_range(ctx, ph, lo, hi - lo, q, o, s))
            check(L.esdg_rhs_phase_range(ctx, ph, 0, lo, q, o, s)); check(L.esdg_rhs_phase_range(ctx, ph, hi, K - hi, q, o, s))
for name, fn in (("full", full), ("split", split), ("full", full), ("split", split)):
    for _ in range(5): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(50): fn()
    torch.cuda.synchronize(); print(name, f"{(time.perf_counter() - t0) / 50 * 1e3:.4f} ms")
ref = out.clone(); full(); torch.cuda.synchronize(); print("split == full:", torch.equal(ref, out))
