"""Hex last-phase kernel by degree: the line-per-lane kh_rhs_l (ESDG_HEX_LINE=1) against kh_rhs (N <= 3) / the row-wise kh_rhs_g (N >= 4)
(ESDG_HEX_LINE=0), same box, same inputs.
    python tools/hex_degree_sweep.py [Kx Kz]   (ms per RHS, nodal DOF updates per second, relative difference of the two results)"""
import os, sys, time
os.environ.setdefault("ESDG_HIP_LIB", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "esdg_cns_amd", "libesdg_hip_ab.so"))   # the A/B build reads the ESDG_* switches; the shipped library reads none
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from esdg_cns_amd import engine

if len(sys.argv) > 1 and sys.argv[1] == "--child":   # one timing in a fresh process (environment switch already set)
    N, Kx, Kz, pn = (int(a) for a in sys.argv[2:6])
    rd, md, ops, Q = bench.build_hex_problem(N, Kx, Kx, Kz, 0, Kx * Kx * Kz, 0.0, bool(pn))
    eng = engine.RhsEngine(rd, md, ops, engine.EULER_HEX_COLLOCATED, lf_scale=0.0)
    Qd = eng.upload(Q); o = eng.new_state()
    for _ in range(150):
        eng.rhs_into(Qd, o)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(20):
        eng.rhs_into(Qd, o)
    torch.cuda.synchronize()
    print((time.perf_counter() - t0) / 20 * 1e3)
    sys.exit(0)
Kx = int(sys.argv[1]) if len(sys.argv) > 1 else 32
Kz = int(sys.argv[2]) if len(sys.argv) > 2 else 16


def ms(eng, Qd, out, n=20):
    for _ in range(150):                      # (clock ramp)
        eng.rhs_into(Qd, out)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n):
        eng.rhs_into(Qd, out)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


for per_node in (True, False):
    for N in range(1, 8):
        rd, md, ops, Q = bench.build_hex_problem(N, Kx, Kx, Kz, 0, Kx * Kx * Kz, 0.0, per_node)
        res = {}
        for tag, env in (("line", None), ("row", "1")):
            if env: os.environ["ESDG_HEX_LINE"] = "0"
            else: os.environ.pop("ESDG_HEX_LINE", None)
            # (the switch is read once per process: a fresh process per setting would be cleaner; the launcher caches it, so
            # the row-wise run comes from a child process below)
            if tag == "row":
                import subprocess
                out = subprocess.run([sys.executable, __file__, "--child", str(N), str(Kx), str(Kz), str(int(per_node))], capture_output=True, text=True,
                                     env=dict(os.environ, ESDG_HEX_LINE="0"))
                res[tag] = (float(out.stdout.strip().split()[-1]), None)
                continue
            os.environ["ESDG_HEX_LINE"] = "1"
            eng = engine.RhsEngine(rd, md, ops, engine.EULER_HEX_COLLOCATED, lf_scale=0.0)
            Qd = eng.upload(Q); o = eng.new_state()
            res[tag] = (ms(eng, Qd, o), o.clone())
            del eng
        os.environ.pop("ESDG_HEX_LINE", None)
        dof = Kx * Kx * Kz * (N + 1) ** 3
        line = res["line"][0]
        s = f"hex N={N} {Kx}x{Kx}x{Kz} {'per-node' if per_node else 'element'} geometry: {line:.4f} ms = {dof / line * 1e3:.3e} DOF/s"
        if "row" in res:
            s += f"   kh_rhs / kh_rhs_g {res['row'][0]:.4f} ms  ratio {line / res['row'][0]:.3f}"
        print(s, flush=True)
