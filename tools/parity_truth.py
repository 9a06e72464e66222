"""e_gpu vs e_orc against the binary128 truth evaluator (oracle/liboracle_quad.so), per configuration.
   e_gpu = |gpu - truth| / |truth|,  e_orc = |oracle_f64 - truth| / |truth|  (max over fields, relative L2)
The engine is built from the SAME set-up objects and state arrays the oracle gets (identical inputs): differences
between two set-up implementations (~1e-13 in J and the normals on a 64x64 mesh) would otherwise show up amplified.
Run on the GPU box:  python tools/parity_truth.py [out.json]"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from common import (product_cavity_problem, product_cns_problem, product_euler_problem, rel_l2, steep_state)  # noqa: E402
from esdg_cns_amd import engine  # noqa: E402
from oracle import oracle as orc  # noqa: E402

orc.lib().oracle_set_threads(orc.lib().oracle_get_max_threads())
orc.lib_quad().oracle_set_threads(orc.lib_quad().oracle_get_max_threads())
rows = []


def gpu(eng, Q):
    return eng.download(eng.rhs(eng.upload(Q)))


def row(name, got, ref, truth):
    r = dict(case=name, e_gpu=rel_l2(got, truth), e_orc=rel_l2(ref, truth), gpu_vs_orc=rel_l2(got, ref))
    rows.append(r)
    print(f"{name:42s} e_gpu {r['e_gpu']:.2e}  e_orc {r['e_orc']:.2e}  gpu-orc {r['gpu_vs_orc']:.2e}", flush=True)


for N, Kx, Ky in [(3, 16, 16), (4, 12, 8), (4, 64, 64), (2, 9, 7), (5, 5, 4), (6, 4, 3)]:
    p = orc.build_euler_problem(N, Kx, Ky)
    o, q = orc.EulerOracle(p), orc.EulerOracle(p, quad=True)
    rd, md, ops, Q = p.rd, p.md, p.ops, p.Q
    eng = engine.RhsEngine(rd, md, ops, engine.EULER_COLLOCATED)
    row(f"euler N={N} {Kx}x{Ky} vortex", gpu(eng, Q), o.rhs(p.Q)[0], q.rhs(p.Q)[0])
    Qw = steep_state(md.xq, md.yq)
    row(f"euler N={N} {Kx}x{Ky} steep", gpu(eng, Qw), o.rhs(Qw)[0], q.rhs(Qw)[0])

for N, Kx, Ky in [(4, 8, 8), (4, 12, 8), (4, 64, 64), (3, 10, 10), (5, 4, 4)]:
    p = orc.build_cns_problem(N, Kx, Ky)
    o, q = orc.CnsOracle(p), orc.CnsOracle(p, quad=True)
    rd, md, ops, Q = p.rd, p.md, p.ops, p.Q
    eng = engine.RhsEngine(rd, md, ops, engine.CNS_MODAL, Re=p.Re, mu=p.mu, lam=p.lam, Pr=p.Pr)
    row(f"cns N={N} {Kx}x{Ky} vortex", gpu(eng, Q), o.rhsRK(p.Q, False)[0], q.rhsRK(p.Q, False)[0])
    Qw = steep_state(md.x, md.y)
    row(f"cns N={N} {Kx}x{Ky} steep", gpu(eng, Qw), o.rhsRK(Qw, False)[0], q.rhsRK(Qw, False)[0])

for BCTYPE in (1, 2, 3):
    N, Kx, Ky = 4, 8, 8
    p = orc.build_cns_problem(N, Kx, Ky, bc="cavity", BCTYPE=BCTYPE)
    o, q = orc.CnsOracle(p), orc.CnsOracle(p, quad=True)
    rd, md, ops, Q = p.rd, p.md, p.ops, p.Q
    eng = engine.RhsEngine(rd, md, ops, engine.CNS_MODAL, Re=p.Re, mu=p.mu, lam=p.lam, Pr=p.Pr, BCTYPE=BCTYPE)
    row(f"cavity BCTYPE={BCTYPE} N={N} {Kx}x{Ky}", gpu(eng, Q), o.rhsRK(p.Q, False)[0], q.rhsRK(p.Q, False)[0])
    rv, rq = o.rhs_viscous(p.Q)[0], q.rhs_viscous(p.Q)[0]
    inv = engine.RhsEngine(rd, md, ops, engine.EULER_MODAL, BCTYPE=BCTYPE)
    gv = [a - b for a, b in zip(gpu(eng, Q), gpu(inv, Q))]
    row(f"cavity BCTYPE={BCTYPE} viscous part (diff)", gv[1:], rv[1:], rq[1:])
    eng.set_parts(2)
    row(f"cavity BCTYPE={BCTYPE} viscous part (parts=2)", gpu(eng, Q)[1:], rv[1:], rq[1:])

for N, K3, lf in [(3, 4, 0.0), (3, 4, 0.25), (2, 5, 0.0)]:
    p = orc.build_hex_problem(N, K3)
    o, q = orc.HexOracle(p, lf), orc.HexOracle(p, lf, quad=True)
    eng = engine.RhsEngine(p.rd, p.md, p.ops, engine.EULER_HEX_COLLOCATED, lf_scale=lf)
    row(f"hex N={N} {K3}^3 lf={lf}", gpu(eng, p.Q), o.rhs(p.Q)[0], q.rhs(p.Q)[0])

if len(sys.argv) > 1:
    json.dump(rows, open(sys.argv[1], "w"), indent=1)
