"""How e_gpu and e_orc (against the binary128 truth) grow with refinement, and where the GPU's excess lives.
Round 3: at Euler N=4 256x256 the GPU was 4.5 x e_orc (64x64: 1.39, 12x8: 1.00).  For K = 32 ... 256 (vortex state):
e_gpu / e_orc for the production kernels and the generic pair-list kernels (the round-1 tensor kernels it also ran are gone)
(ESDG_FORCE_GENERIC=1); the split of both errors into the vortex core (r < 3) and the far field; the free-stream residual
of GPU and oracle.    python tools/parity_scaling.py [cns] [Kmax]"""
import os
os.environ.setdefault("ESDG_HIP_LIB", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "esdg_cns_amd", "libesdg_hip_ab.so"))   # the A/B build reads the ESDG_* switches; the shipped library reads none
import sys
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main(form, K, mode):
    import numpy as np
    from common import as_oracle_problem, product_cns_problem, product_euler_problem
    from esdg_cns_amd import engine, physics as ph
    from oracle import oracle as orc
    n = len(os.sched_getaffinity(0))
    orc.lib().oracle_set_threads(n); orc.lib_quad().oracle_set_threads(n)
    PHYS = dict(Re=1000.0, mu=1e-3, lam=-2e-3 / 3, Pr=.71, BCTYPE=1)
    if form == "euler":
        rd, md, ops, Q = product_euler_problem(4, K, K)
        p = as_oracle_problem(rd, md, ops, Q)
        o, q = orc.EulerOracle(p), orc.EulerOracle(p, quad=True)
        f64, truth = (lambda Q: o.rhs(Q)[0]), (lambda Q: q.rhs(Q)[0])
        eng = engine.RhsEngine(rd, md, p.ops, engine.EULER_COLLOCATED)
        x, y = md.xq, md.yq
    else:
        rd, md, ops, Q = product_cns_problem(4, K, K)
        p = as_oracle_problem(rd, md, ops, Q, **PHYS)
        o, q = orc.CnsOracle(p), orc.CnsOracle(p, quad=True)
        f64, truth = (lambda Q: o.rhsRK(Q, False)[0]), (lambda Q: q.rhsRK(Q, False)[0])
        eng = engine.RhsEngine(rd, md, ops, engine.CNS_MODAL, Re=p.Re, mu=p.mu, lam=p.lam, Pr=p.Pr)
        x, y = md.x, md.y
    g = eng.download(eng.rhs(eng.upload(Q)))
    a, t = f64(Q), truth(Q)
    core = ((x - 7.5) ** 2 + y ** 2 < 9.0).all(axis=0)     # elements wholly within r < 3 of the vortex centre
    def nrm(u, v, m=None):
        return max(np.linalg.norm((uu - vv)[:, m] if m is not None else uu - vv) / np.linalg.norm(vv) for uu, vv in zip(u, v))
    per_field = [np.linalg.norm(gg - tt) / np.linalg.norm(tt) for gg, tt in zip(g, t)]
    one = np.ones_like(Q[0])
    Qc = [np.asfortranarray(qq) for qq in ph.primitive_to_conservative(1.1 * one, .3 * one, -.2 * one, .9 * one)]
    fs_g = max(float(np.abs(r).max()) for r in eng.download(eng.rhs(eng.upload(Qc))))
    fs_o = max(float(np.abs(r).max()) for r in f64(Qc))
    print(f"{form} K={K:4d} {mode:8s} e_gpu {nrm(g, t):.2e} e_orc {nrm(a, t):.2e} ratio {nrm(g, t) / nrm(a, t):.2f} | core: gpu {nrm(g, t, core):.2e} orc {nrm(a, t, core):.2e}"
          f" | far: gpu {nrm(g, t, ~core):.2e} orc {nrm(a, t, ~core):.2e} | per field gpu {' '.join('%.1e' % v for v in per_field)}"
          f" | free stream max|rhs|: gpu {fs_g:.2e} orc {fs_o:.2e}", flush=True)


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "--one":
        main(sys.argv[2], int(sys.argv[3]), sys.argv[4])
        sys.exit(0)
    form = "cns" if "cns" in sys.argv[1:] else "euler"
    kmax = max([int(a) for a in sys.argv[1:] if a.isdigit()] or [256])
    for K in (32, 64, 128, 256):
        if K > kmax:
            break
        for mode, env in (("v2", {}), ("generic", {"ESDG_FORCE_GENERIC": "1"})):
            if mode == "generic" and K > 128:
                continue
            e = dict(os.environ); e.update(env)
            subprocess.call([sys.executable, os.path.abspath(__file__), "--one", form, str(K), mode], env=e)
