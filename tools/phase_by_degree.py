"""Every phase by degree (CNS on Kx x Kx quads): ms per launch, and the whole evaluation as nodal DOF updates per second and as
bytes-at-the-practical-rate (the design byte counts of DESIGN.md section 4 scaled with the degree: per element 3 x 32 Np state reads,
32 Np rhs write, A_U 32 Nfq written once and read four times, B 24 Nfq written once and read twice, SG 24 Np written and read).
    python tools/phase_by_degree.py [Kx]"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from esdg_cns_amd import engine
from esdg_cns_amd._lib import check

Kx = int(sys.argv[1]) if len(sys.argv) > 1 else 256
for N in range(1, 10):
    rd, md, ops, Q = bench.build_problem(N, Kx, Kx, 0, Kx * Kx, "cns")
    eng = engine.RhsEngine(rd, md, ops, engine.CNS_MODAL)
    Qd, out = eng.upload(Q), eng.new_state()
    q, o = C.c_void_p(Qd.data_ptr()), C.c_void_p(out.data_ptr())
    for _ in range(100):
        check(eng.L.esdg_rhs(eng.ctx, q, o, eng._stream()))
    ms = []
    for ph in range(eng.nphases):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        e0.record(torch.cuda.current_stream())
        for _ in range(30):
            check(eng.L.esdg_rhs_phase(eng.ctx, ph, q, o, eng._stream()))
        e1.record(torch.cuda.current_stream())
        torch.cuda.synchronize()
        ms.append(e0.elapsed_time(e1) / 30)
    Np, Nfq, K = (N + 1) ** 2, 4 * (N + 1), Kx * Kx
    byts = K * (3 * 32 * Np + 32 * Np + 5 * 32 * Nfq + 3 * 24 * Nfq + 2 * 24 * Np)
    tot = sum(ms)
    print(f"cns N={N} {Kx}x{Kx}: phases {' / '.join('%.4f' % m for m in ms)} = {tot:.4f} ms; {K * Np / tot / 1e6:.2f} G DOF/s; "
          f"design bytes {byts / 1e6:.0f} MB = {byts / 4.7e9:.4f} ms at 4.7 TB/s -> {tot / (byts / 4.7e9):.2f} x", flush=True)
    del eng
