"""Lid-driven cavity (walls, BCTYPE 1): ms per CNS right-hand side, current kernels vs ESDG_V1=walls (round-1 kernels on wall
meshes), and the difference of the two results.   python tools/cavity_ab.py [N Kx]"""
import os, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
if len(sys.argv) > 1 and sys.argv[1] == "--child":
    import numpy as np, torch
    from common import product_cavity_problem
    from esdg_cns_amd import engine as E
    N, Kx = int(sys.argv[2]), int(sys.argv[3])
    rd, md, ops, Q = product_cavity_problem(N, Kx, Kx)
    eng = E.RhsEngine(rd, md, ops, E.CNS_MODAL, BCTYPE=1)
    Qd, out = eng.upload(Q), eng.new_state()
    for _ in range(200):
        eng.rhs_into(Qd, out)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(100):
        eng.rhs_into(Qd, out)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) * 10
    np.save(sys.argv[4], out.cpu().numpy())
    print(f"{os.environ.get('ESDG_V1', 'v2')}: {ms:.4f} ms per RHS (cavity N={N} {Kx}x{Kx})")
else:
    import numpy as np
    N, Kx = (sys.argv[1:3] + ["4", "256"])[:2] if len(sys.argv) > 2 else ("4", "256")
    outs = []
    for v in (None, "walls"):
        env = dict(os.environ)
        if v: env["ESDG_V1"] = v
        f = f"/tmp/cav_{v}.npy"
        subprocess.run([sys.executable, __file__, "--child", N, Kx, f], env=env, check=True)
        outs.append(np.load(f))
    print("max rel difference of the two results: %.2e" % (np.abs(outs[0] - outs[1]).max() / np.abs(outs[1]).max()))
