"""Experiment: one CNS evaluation at cfg3 as a phase pipeline over strips of S element rows, so that
the traces a strip's phase p writes are read by its phase p+1 while they are still in the 256 MiB Infinity Cache, against
one launch per phase.  Uses only esdg_rhs_phase_range; the result must equal the plain evaluation bit for bit.
    python tools/strip_pipeline.py [S ...]          (ESDG_T2_RESERVE=0 is set: no slot reserve for ranged launches)"""
import ctypes as C, os, sys, time
os.environ.setdefault("ESDG_HIP_LIB", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "esdg_cns_amd", "libesdg_hip_ab.so"))   # the A/B build reads the ESDG_* switches; the shipped library reads none
os.environ.setdefault("ESDG_T2_RESERVE", "0")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from esdg_cns_amd import engine
from esdg_cns_amd._lib import check

Kx, Ky = int(os.environ.get("KX", "512")), int(os.environ.get("KY", "512"))   # (KX=2048 KY=256: rank 0's strip of cfg4 as a stand-alone periodic mesh)
rd, md, ops, Q = bench.build_problem(4, Kx, Ky, 0, Kx * Ky, "cns")
eng = engine.RhsEngine(rd, md, ops, engine.CNS_MODAL)
Qd = eng.upload(Q); out = eng.new_state()
L, ctx = eng.L, eng.ctx
q, o = C.c_void_p(Qd.data_ptr()), C.c_void_p(out.data_ptr())


def rows(ph, r0, r1):
    if r1 > r0:
        check(L.esdg_rhs_phase_range(ctx, ph, r0 * Kx, (r1 - r0) * Kx, q, o, eng._stream()))


def full():
    check(L.esdg_rhs(ctx, q, o, eng._stream()))


def pipeline(S):
    # rows are periodic in y: phase p of row r needs phase p-1 of rows r-1, r, r+1 (mod Ky)
    rows(0, Ky - 2, Ky)                       # prologue: the rows the wrap-around needs
    d0 = d1 = d2 = 0                          # leading rows done per phase
    k = 0
    while d2 < Ky:
        k += 1
        b0 = min(k * S + 2, Ky - 2); rows(0, d0, b0); d0 = max(d0, b0)
        if k == 1:
            rows(1, Ky - 1, Ky)
        b1 = min(k * S + 1, Ky - 1); rows(1, d1, b1); d1 = max(d1, b1)
        b2 = min(k * S, Ky); rows(2, d2, b2); d2 = b2


def timeit(fn, n=50):
    for _ in range(5): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


for _ in range(200): full()      # clock ramp
full(); torch.cuda.synchronize(); ref = out.clone()
sizes = [int(a) for a in sys.argv[1:]] or [16, 32, 64, 128]
print(f"mesh {Kx} x {Ky}")
print(f"full: {timeit(full):.4f} ms")
for S in sizes:
    out.zero_(); pipeline(S); torch.cuda.synchronize()
    print(f"pipeline S={S}: {timeit(lambda: pipeline(S)):.4f} ms  equal={torch.equal(ref, out)}  launches={3 * ((Ky + S - 1) // S) + 2}")
print(f"full: {timeit(full):.4f} ms")
