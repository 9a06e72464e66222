"""Soak of the fused DOPRI45 attempt at cfg3's size: `n` attempts of the fused path and of the building blocks in lockstep (same step
sizes), states compared bit for bit every 50 attempts -- a race in the in-place stage update would show as a difference.
    python tools/dopri_soak.py [n]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
from esdg_cns_amd import engine as E, timestep as TS

n = int(sys.argv[1]) if len(sys.argv) > 1 else 300
rd, md, ops, Q = bench.build_problem(4, 512, 512, 0, 512 * 512, "cns")
eng = E.RhsEngine(rd, md, ops, E.CNS_MODAL)
a = TS.Dopri45(eng, eng.upload(bench.rough_state(Q)), 2e-4, err_tol=1e-5, swap=True)
b = TS.Dopri45(eng, eng.upload(bench.rough_state(Q)), 2e-4, err_tol=1e-5, pieces=True, swap=True)
acc = 0
for i in range(1, n + 1):
    dt, prev = a.dt, a.prev_err
    b.dt, b.prev_err = dt, prev
    ok, err = a.step()
    ok2, err2 = b.step()
    acc += bool(ok)
    assert ok == ok2 and abs(err - err2) <= 1e-11 * err2, (i, err, err2)
    if i % 50 == 0 or i == n:
        torch.cuda.synchronize()
        same = torch.equal(a.Q, b.Q) and all(torch.equal(x, y) for x, y in zip(a.k, b.k))
        print(f"attempt {i}: t = {a.t:.5e}, dt = {a.dt:.3e}, errEst = {err:.3e}, accepted {acc}, fused == building blocks: {same}", flush=True)
        assert same and torch.isfinite(a.Q).all()
print("OK")
