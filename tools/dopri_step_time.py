"""DOPRI45 attempt of the cavity driver's time loop (dg2D_CNS_cavity_optimized.jl:999-1037) at cfg3's size: ms per attempted step
against six right-hand sides, i.e. what the stage combinations and the error norm cost on top of the hot path.
    python tools/dopri_step_time.py [N Kx reps]"""
import os, sys, time
os.environ.setdefault("ESDG_HIP_LIB", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "esdg_cns_amd", "libesdg_hip_ab.so"))   # the A/B build reads the ESDG_* switches; the shipped library reads none
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
from esdg_cns_amd import engine as E, timestep as TS

a = sys.argv[1:4]
N, Kx, reps = int(a[0]) if len(a) > 0 else 4, int(a[1]) if len(a) > 1 else 512, int(a[2]) if len(a) > 2 else 20
rd, md, ops, Q = bench.build_problem(N, Kx, Kx, 0, Kx * Kx, "cns")
eng = E.RhsEngine(rd, md, ops, E.CNS_MODAL)
Qd, out = eng.upload(Q), eng.new_state()
for _ in range(200):
    eng.rhs_into(Qd, out)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(60):
    eng.rhs_into(Qd, out)
torch.cuda.synchronize()
rhs_ms = (time.perf_counter() - t0) / 60 * 1e3
dp = TS.Dopri45(eng, Qd, 1e-5, err_tol=1e-5, swap=True)
for _ in range(3):
    dp.step()
torch.cuda.synchronize()
t0 = time.perf_counter()
acc = 0
for _ in range(reps):
    ok, err = dp.step()
    acc += bool(ok)
torch.cuda.synchronize()
step_ms = (time.perf_counter() - t0) / reps * 1e3
sweep = Qd.numel() * 8 / 1e9
print(f"cns N={N} {Kx}x{Kx}: rhs {rhs_ms:.4f} ms, dopri45 attempt {step_ms:.4f} ms = 6 rhs + {step_ms - 6 * rhs_ms:.4f} ms "
      f"({(step_ms - 6 * rhs_ms) / step_ms * 100:.1f} % of the step; one state sweep = {sweep:.3f} GB = {sweep / 4.7:.4f} ms at 4.7 TB/s; "
      f"accepted {acc}/{reps}, last err {err:.3e}, dt {dp.dt:.3e})")
