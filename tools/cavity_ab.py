"""Lid-driven cavity (walls, BCTYPE 1): ms per CNS right-hand side of the current kernels (kt2_project, kt2_sigma, kt3_rhs with its
wall instantiation) and of the same with the v2 last phase (ESDG_V2=rhs: kt2_rhs), and the difference of the results.
(The round-1 kernels this tool also timed until round 4 -- ESDG_V1=walls -- are gone.)   python tools/cavity_ab.py [N Kx]"""
import os, subprocess, sys, time
os.environ.setdefault("ESDG_HIP_LIB", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "esdg_cns_amd", "libesdg_hip_ab.so"))   # the A/B build reads the ESDG_* switches; the shipped library reads none
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
    tag = "ESDG_V2=" + os.environ["ESDG_V2"] if "ESDG_V2" in os.environ else "default"
    print(f"{tag}: {ms:.4f} ms per RHS (cavity N={N} {Kx}x{Kx})")
else:
    import numpy as np
    N, Kx = (sys.argv[1:3] + ["4", "256"])[:2] if len(sys.argv) > 2 else ("4", "256")
    outs = []
    for k, v in ((None, None), ("ESDG_V2", "rhs"), (None, None), ("ESDG_V2", "rhs")):
        env = dict(os.environ)
        if k: env[k] = v
        f = f"/tmp/cav_{k}_{v}.npy"
        subprocess.run([sys.executable, __file__, "--child", N, Kx, f], env=env, check=True)
        outs.append(np.load(f))
    for i, name in ((1, "ESDG_V2=rhs"),):
        print("max rel difference of %s to the default: %.2e" % (name, np.abs(outs[0] - outs[i]).max() / np.abs(outs[0]).max()))
