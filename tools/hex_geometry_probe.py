"""Hex path: e_gpu / e_orc against the binary128 truth with the element-constant geometry record (ESDG_HEX_GEOMETRY=element),
with the record plus 10-bit per-node differences (default on affine meshes whose driver passes per-node arrays) and with every
node's own metric terms and normals in full (ESDG_HEX_PER_NODE=1, the kernels of curved meshes), for K^3 boxes.
    python tools/hex_geometry_probe.py [K ...]"""
import os, subprocess, sys
os.environ.setdefault("ESDG_HIP_LIB", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "esdg_cns_amd", "libesdg_hip_ab.so"))   # the A/B build reads the ESDG_* switches; the shipped library reads none
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))


def one(K, mode):
    import numpy as np
    from common import rel_l2
    from esdg_cns_amd import engine
    from oracle import oracle as orc
    n = len(os.sched_getaffinity(0))
    orc.lib().oracle_set_threads(n); orc.lib_quad().oracle_set_threads(n)
    p = orc.build_hex_problem(3, K)
    o, q = orc.HexOracle(p, 0.0), orc.HexOracle(p, 0.0, quad=True)
    eng = engine.RhsEngine(p.rd, p.md, p.ops, engine.EULER_HEX_COLLOCATED, lf_scale=0.0)
    g = eng.download(eng.rhs(eng.upload(p.Q)))
    a, t = o.rhs(p.Q)[0], q.rhs(p.Q)[0]
    print(f"hex N=3 {K}^3 {mode:9s} e_gpu {rel_l2(g, t):.2e} e_orc {rel_l2(a, t):.2e} ratio {rel_l2(g, t) / rel_l2(a, t):.2f}", flush=True)


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "--one":
        one(int(sys.argv[2]), sys.argv[3]); sys.exit(0)
    for K in [int(a) for a in sys.argv[1:]] or [8, 16]:
        for mode, env in (("element", {"ESDG_HEX_GEOMETRY": "element"}), ("10-bit", {}), ("per-node", {"ESDG_HEX_PER_NODE": "1"})):
            e = dict(os.environ); e.update(env)
            subprocess.call([sys.executable, os.path.abspath(__file__), "--one", str(K), mode], env=e)
