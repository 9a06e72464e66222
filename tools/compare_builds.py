#!/usr/bin/env python3
"""Bit-for-bit comparison of two builds of libesdg_hip.so on the same inputs (GPU box).

    python tools/compare_builds.py esdg_cns_amd/variants/r04.so [other.so]     (second default: the in-tree library)

Each library runs in its own child process (ESDG_HIP_LIB) over a list of cases -- CNS / Euler, periodic / cavity walls, smooth and
rough states, several degrees -- and writes its right-hand sides to a scratch file; the parent compares the bits and prints,
per case, the number of differing entries and the largest relative difference.  Used for changes that claim to be bitwise
neutral (round 5: the EC flux from sums instead of averages, the smooth-wave flux selection of kt3_rhs)."""
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

CASES = [  # (name, formulation, N, Kx, Ky, walls, rough)
    ("cns_N4_96x64", "cns", 4, 96, 64, 0, 0),
    ("cns_N4_96x64_rough", "cns", 4, 96, 64, 0, 1),
    ("euler_N4_64x64", "euler", 4, 64, 64, 0, 0),
    ("euler_N3_33x17_rough", "euler", 3, 33, 17, 0, 1),
    ("cns_N2_40x24", "cns", 2, 40, 24, 0, 0),
    ("cns_N3_40x24_rough", "cns", 3, 40, 24, 0, 1),
    ("cns_N5_24x24", "cns", 5, 24, 24, 0, 0),
    ("cns_N6_16x16_rough", "cns", 6, 16, 16, 0, 1),
    ("cns_N7_16x12", "cns", 7, 16, 12, 0, 0),
    ("cavity_N4_32x32", "cns", 4, 32, 32, 1, 0),
    ("cavity_N3_24x24_bc3", "cns", 3, 24, 24, 3, 0),
]


def child(out):
    import bench
    from esdg_cns_amd import engine
    from esdg_cns_amd import setup_dg as sd
    res = {}
    for name, form, N, Kx, Ky, walls, rough in CASES:
        if walls:
            sys.path.insert(0, os.path.join(ROOT, "tests"))
            import common
            rd, md, ops, Q = common.product_cavity_problem(N, Kx, Ky)
            eng = engine.RhsEngine(rd, md, ops, engine.CNS_MODAL, device="cuda:0", BCTYPE=walls)
        else:
            rd, md, ops, Q = bench.build_problem(N, Kx, Ky, 0, Kx * Ky, form)
            if rough:
                Q = bench.rough_state(Q)
            eng = engine.RhsEngine(rd, md, ops, engine.CNS_MODAL if form == "cns" else engine.EULER_COLLOCATED, device="cuda:0")
        r = eng.download(eng.rhs(eng.upload(Q)))
        for f, a in enumerate(r):
            res["%s/%d" % (name, f)] = np.asarray(a)
    np.savez(out, **res)


def main():
    if len(sys.argv) >= 3 and sys.argv[1] == "--child":
        return child(sys.argv[2])
    libs = [os.path.abspath(p) for p in sys.argv[1:3]]
    if len(libs) < 1:
        raise SystemExit(__doc__)
    if len(libs) == 1:
        libs.append(os.path.join(ROOT, "esdg_cns_amd", "libesdg_hip.so"))
    outs = []
    for i, lib in enumerate(libs):
        if not os.path.exists(lib):
            raise SystemExit("missing library: " + lib)
        out = "/tmp/compare_builds_%d.npz" % i
        env = dict(os.environ, ESDG_HIP_LIB=lib)
        subprocess.run([sys.executable, os.path.abspath(__file__), "--child", out], check=True, env=env, cwd=ROOT)
        outs.append(np.load(out))
    a, b = outs
    print("A = %s\nB = %s" % tuple(libs))
    bad = 0
    for name, *_ in CASES:
        nd, rel = 0, 0.0
        for f in range(4):
            x, y = a["%s/%d" % (name, f)], b["%s/%d" % (name, f)]
            nd += int(np.count_nonzero(x.view(np.uint64) != y.view(np.uint64)))
            rel = max(rel, float(np.max(np.abs(x - y)) / max(np.max(np.abs(y)), 1e-300)))
        bad += nd
        print("%-26s differing entries %8d   max |A-B| / max|B| %.2e" % (name, nd, rel))
    print("TOTAL differing entries:", bad)
    return 0


if __name__ == "__main__":
    sys.exit(main())
