"""Soak of the fused DOPRI45 attempt at cfg3's size (or: euler = cfg2's size, hex = a 64 x 64 x 16 slab): `n` attempts of the fused path and of the building blocks in lockstep (same step
sizes), states compared bit for bit every 50 attempts -- a race in the in-place stage update would show as a difference.
    python tools/dopri_soak.py [n [cns|euler|hex|cavity [free]]]
`free` (late round 5): NO lockstep -- each run follows its own error estimates through the controller.  Since the norm's terms are
added in one order (esdg_kernels.hip: k_dopri_err / k_chunk_sum) the estimates, hence the step sizes, hence the states must stay
equal bit for bit without any help; every attempt's estimate and step size are compared."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
from esdg_cns_amd import engine as E, timestep as TS

n = int(sys.argv[1]) if len(sys.argv) > 1 else 300
free = len(sys.argv) > 3 and sys.argv[3] == "free"
form = sys.argv[2] if len(sys.argv) > 2 else "cns"      # cns | euler (cfg2's size) | hex (64 x 64 x 16 elements, N = 3)
if form == "hex":
    rd, md, ops, Q = bench.build_hex_problem(3, 64, 64, 16, 0, 64 * 64 * 16)
    eng = E.RhsEngine(rd, md, ops, E.EULER_HEX_COLLOCATED, lf_scale=0.0)   # (the reference's 0*.25; a non-zero factor is anti-dissipative on its J < 0 meshes and blows up by t = 0.02: docs/history.md section 9)
    Q0, dt0 = Q, 2e-4
elif form == "cavity":   # the reference driver's own case: lid-driven cavity, adiabatic no-slip walls, N = 4 on 128 x 128 (kt3_rhs's wall form)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from common import product_cavity_problem
    rd, md, ops, Q0 = product_cavity_problem(4, 128, 128)
    eng = E.RhsEngine(rd, md, ops, E.CNS_MODAL, BCTYPE=1)
    dt0 = 1e-4
elif form == "euler":
    rd, md, ops, Q = bench.build_problem(4, 256, 256, 0, 256 * 256, "euler")
    eng = E.RhsEngine(rd, md, ops, E.EULER_COLLOCATED)
    Q0, dt0 = bench.rough_state(Q), 2e-4
else:
    rd, md, ops, Q = bench.build_problem(4, 512, 512, 0, 512 * 512, "cns")
    eng = E.RhsEngine(rd, md, ops, E.CNS_MODAL)
    Q0, dt0 = bench.rough_state(Q), 2e-4
a = TS.Dopri45(eng, eng.upload(Q0), dt0, err_tol=1e-5, swap=True)
b = TS.Dopri45(eng, eng.upload(Q0), dt0, err_tol=1e-5, pieces=True, swap=True)
acc = 0
for i in range(1, n + 1):
    if not free:
        b.dt, b.prev_err = a.dt, a.prev_err
    ok, err = a.step()
    ok2, err2 = b.step()
    acc += bool(ok)
    assert ok == ok2 and err == err2 and a.dt == b.dt and a.t == b.t, (i, err, err2, a.dt, b.dt)
    if i % 50 == 0 or i == n:
        torch.cuda.synchronize()
        same = torch.equal(a.Q, b.Q) and all(torch.equal(x, y) for x, y in zip(a.k, b.k))
        print(f"attempt {i}: t = {a.t:.5e}, dt = {a.dt:.3e}, errEst = {err:.3e}, accepted {acc}, fused == building blocks: {same}" + (" (free-running)" if free else ""), flush=True)
        assert same and torch.isfinite(a.Q).all()
print("OK")
