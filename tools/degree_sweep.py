"""Last-phase kernel by degree: kt3_rhs (default) against kt2_rhs (ESDG_V2=rhs), CNS and collocated Euler, same box, same inputs.
    python tools/degree_sweep.py [Kx]      (prints ms per launch of the last phase and the relative difference of the two results)"""
import ctypes as C, os, sys
os.environ.setdefault("ESDG_HIP_LIB", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "esdg_cns_amd", "libesdg_hip_ab.so"))   # the A/B build reads the ESDG_* switches; the shipped library reads none
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import bench
from esdg_cns_amd import engine
from esdg_cns_amd._lib import check

Kx = int(sys.argv[1]) if len(sys.argv) > 1 else 256


def last_phase_ms(eng, Qd, out, n=30):
    q, o = C.c_void_p(Qd.data_ptr()), C.c_void_p(out.data_ptr())
    for _ in range(60):
        check(eng.L.esdg_rhs(eng.ctx, q, o, eng._stream()))
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ph = eng.nphases - 1
    torch.cuda.synchronize()
    e0.record(torch.cuda.current_stream())
    for _ in range(n):
        check(eng.L.esdg_rhs_phase(eng.ctx, ph, q, o, eng._stream()))
    e1.record(torch.cuda.current_stream())
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


for form in ("cns", "euler"):
    for N in range(1, 8):
        rd, md, ops, Q = bench.build_problem(N, Kx, Kx, 0, Kx * Kx, form)
        res = {}
        for tag, env in (("kt3", None), ("kt2", "rhs")):
            if env: os.environ["ESDG_V2"] = env
            else: os.environ.pop("ESDG_V2", None)
            eng = engine.RhsEngine(rd, md, ops, engine.CNS_MODAL if form == "cns" else engine.EULER_COLLOCATED)
            Qd = eng.upload(Q); out = eng.new_state()
            ms = last_phase_ms(eng, Qd, out)
            eng.rhs_into(Qd, out); torch.cuda.synchronize()
            res[tag] = (ms, out.clone())
            del eng
        os.environ.pop("ESDG_V2", None)
        a, b = res["kt3"][1], res["kt2"][1]
        rel = float((a - b).norm() / b.norm())
        print(f"{form} N={N} {Kx}x{Kx}: kt3_rhs {res['kt3'][0]:.4f} ms  kt2_rhs {res['kt2'][0]:.4f} ms  ratio {res['kt3'][0] / res['kt2'][0]:.3f}  |kt3-kt2|/|kt2| {rel:.2e}", flush=True)
