"""Experiment (round 3, late): phase 1 (kt2_sigma, memory side) and the last phase (kt2_rhs, VALU side) of one CNS evaluation at
cfg3 run CONCURRENTLY on two streams, strip by strip: stream A does phase 0 of the whole mesh, then phase 1 over strips of S
element rows; stream B does the last phase of strip k as soon as phase 1 of strips k and k + 1 (and of the wrap-around row) is
done.  Uses only esdg_rhs_phase_range; the result must equal the plain evaluation bit for bit.
    python tools/strip_concurrent.py [S ...]        (ESDG_T2_RESERVE=0; ESDG_T2_WG_PER_CU=n limits the persistent kt2_sigma grid)"""
import ctypes as C, os, sys, time
os.environ.setdefault("ESDG_HIP_LIB", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "esdg_cns_amd", "libesdg_hip_ab.so"))   # the A/B build reads the ESDG_* switches; the shipped library reads none
os.environ.setdefault("ESDG_T2_RESERVE", "0")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from esdg_cns_amd import engine
from esdg_cns_amd._lib import check

Kx = Ky = 512
rd, md, ops, Q = bench.build_problem(4, Kx, Ky, 0, Kx * Ky, "cns")
eng = engine.RhsEngine(rd, md, ops, engine.CNS_MODAL)
Qd = eng.upload(Q); out = eng.new_state()
L, ctx = eng.L, eng.ctx
q, o = C.c_void_p(Qd.data_ptr()), C.c_void_p(out.data_ptr())
sA, sB = torch.cuda.Stream(), torch.cuda.Stream()
hA, hB = C.c_void_p(sA.cuda_stream), C.c_void_p(sB.cuda_stream)


def rows(ph, r0, r1, h):
    if r1 > r0:
        check(L.esdg_rhs_phase_range(ctx, ph, r0 * Kx, (r1 - r0) * Kx, q, o, h))


def full():
    check(L.esdg_rhs(ctx, q, o, hA))


def concurrent(S):
    K = (Ky + S - 1) // S
    ev = [torch.cuda.Event() for _ in range(K)]
    sA.wait_stream(sB)                         # the previous evaluation's last phase has read B / SG
    rows(0, 0, Ky, hA)
    rows(1, Ky - 1, Ky, hA)                    # the row the wrap-around of strip 0 needs
    for k in range(K):
        rows(1, k * S, min((k + 1) * S, Ky - 1 if k == K - 1 else Ky), hA)
        ev[k].record(sA)
        if k >= 1:
            sB.wait_event(ev[k]); rows(2, (k - 1) * S, k * S, hB)
    sB.wait_event(ev[K - 1]); rows(2, (K - 1) * S, Ky, hB)


def timeit(fn, n=50):
    for _ in range(5): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


for _ in range(200): full()      # clock ramp
full(); torch.cuda.synchronize(); ref = out.clone()
sizes = [int(a) for a in sys.argv[1:]] or [32, 64, 128]
print(f"WG_PER_CU={os.environ.get('ESDG_T2_WG_PER_CU', 'default')}  full: {timeit(full):.4f} ms", flush=True)
for S in sizes:
    out.zero_(); torch.cuda.synchronize(); concurrent(S); torch.cuda.synchronize()
    print(f"concurrent S={S}: {timeit(lambda: concurrent(S)):.4f} ms  equal={torch.equal(ref, out)}", flush=True)
print(f"full: {timeit(full):.4f} ms")
