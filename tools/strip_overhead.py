"""What does the sharded, overlapped schedule cost on one GPU?  Rank 0's strip of the 8-rank BASELINE config 4 mesh
(2048 x 256 elements, N=4, CNS) evaluated (a) through the library's RCCL transport in loopback -- boundary strips, packs,
grouped ncclSend/ncclRecv on the comm stream, interior launches, exactly what an 8-GPU run executes per rank -- and (b) as
a stand-alone periodic strip in one launch per phase.  Prints ms per evaluation of both and the ratio (the per-GPU rate a
weak-scaled run can reach relative to an unsharded one, network aside)."""
import copy
import os
os.environ.setdefault("ESDG_HIP_LIB", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "esdg_cns_amd", "libesdg_hip_ab.so"))   # the A/B build reads the ESDG_* switches; the shipped library reads none
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: E402
from esdg_cns_amd import engine as E, setup_dg as sd  # noqa: E402
from test_gpu_engine import _strip_periodic_state  # noqa: E402

HEX = len(sys.argv) > 1 and sys.argv[1] == "hex"     # python tools/strip_overhead.py hex [Kx Kz_per_rank]: rank 0's slab of cfg5
if HEX:
    del sys.argv[1]
N, Kx, Kyr, nr = (3 if HEX else 4), int(sys.argv[1]) if len(sys.argv) > 1 else (128 if HEX else 2048), \
    int(sys.argv[2]) if len(sys.argv) > 2 else (16 if HEX else 256), 8


def build(Ky_total, e0, e1):
    VX, VY, EToV = sd.uniform_quad_mesh(Kx, Ky_total)
    VX = 15 * (1 + VX) / 2
    VY = 5 * VY * (Ky_total / Kx)
    rd = sd.init_reference_quad(N)
    md = sd.init_mesh((VX, VY), EToV, rd, elem_range=(e0, e1))
    sd.make_periodic(md, rd)
    md.mapB = np.zeros(0, dtype=np.int64)
    ops = sd.cns_ops(rd)
    sd.interp_geofacs_to_hybrid(md, ops["Vh"])
    return rd, md, ops


if HEX:
    from test_gpu_hex import _slab_periodic_state

    def build(Kz_total, e0, e1):
        VX, VY, VZ, EToV = sd.uniform_hex_mesh(Kx, Kx, Kz_total)
        VZ = VZ * (Kz_total / Kx)
        rd = sd.init_reference_hex(N, sd.gauss_quad(0, 0, N))
        md = sd.init_mesh_3d((VX, VY, VZ), EToV, rd, elem_range=(e0, e1))
        sd.make_periodic_3d(md, rd)
        ops = sd.hex_ops(rd)
        sd.hex_driver_geometry(md, rd, hybrid=False)
        return rd, md, ops

    Ks = Kx * Kx * Kyr
    form, kw = E.EULER_HEX_COLLOCATED, dict(lf_scale=0.0)
else:
    Ks = Kx * Kyr
    form, kw = E.CNS_MODAL, {}
rd, md, ops = build(Kyr * nr, 0, Ks)
offsets = np.array([Ks * r for r in range(nr + 1)], dtype=np.int64)
sh = E.RhsEngine(rd, md, ops, form, rank=0, nranks=nr, rank_offsets=offsets, **kw)
sh.attach_rccl(loopback=True)
_, md1s, _ = build(Kyr, 0, Ks)
md1 = copy.copy(md)
md1.mapP, md1.elem_offset, md1.Kglobal = md1s.mapP, 0, md.K
one = E.RhsEngine(rd, md1, ops, form, **kw)
Q = _slab_periodic_state(md.xq, md.yq, md.zq, 2.0 * Kyr / Kx) if HEX else _strip_periodic_state(md.x, md.y, 10.0 * Kyr / Kx)
Qd = sh.upload(Q)
out1, out2 = sh.new_state(), one.new_state()


QUICK = os.environ.get("STRIP_QUICK") == "1"      # short run for a rocprofv3 --kernel-trace timeline


def timeit(eng, out, n=60):
    n = 20 if QUICK else n
    for _ in range(100 if QUICK else 150):
        eng.rhs_into(Qd, out)
    torch.cuda.synchronize()
    best = []
    for _ in range(1 if QUICK else 5):
        t0 = time.perf_counter()
        for _ in range(n):
            eng.rhs_into(Qd, out)
        torch.cuda.synchronize()
        best.append((time.perf_counter() - t0) / n * 1e3)
    return float(np.median(best)), float(np.min(best))


a = timeit(sh, out1)
b = timeit(one, out2)
a2 = timeit(sh, out1)
assert torch.equal(out1, out2)
print(f"{'hex slab' if HEX else 'strip'} {Kx}x{Kyr} N={N}: sharded schedule over RCCL loopback {a[0]:.4f} ms (min {a[1]:.4f}; again {a2[0]:.4f}), "
      f"one launch per phase {b[0]:.4f} ms (min {b[1]:.4f}); ratio {a[0] / b[0]:.3f}; interior {sh.interior}")
