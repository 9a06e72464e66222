"""Diagnostic: per-section s_memtime deltas of kf_rhs (ESDG_DBG=8)."""
import ctypes as C, os, sys
os.environ["ESDG_DBG"] = "8"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from esdg_cns_amd import engine, _lib
rd, md, ops, Q = bench.build_problem(4, 512, 512, 0, 512 * 512, "cns")
eng = engine.RhsEngine(rd, md, ops, engine.CNS_MODAL)
Qd = eng.upload(Q); out = eng.new_state()
for _ in range(3): eng.rhs_into(Qd, out)
torch.cuda.synchronize()
L = _lib.lib()
L.esdg_debug_stamps.restype = C.c_int
L.esdg_debug_stamps.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
buf = np.zeros(4096 * 16, dtype=np.uint64)
_lib.check(L.esdg_debug_stamps(eng.ctx, buf.ctypes.data_as(C.c_void_p), buf.size))
st = buf.reshape(4096, 16)[:, :9].astype(np.int64)
st = st[st[:, 0] > 0]
d = np.diff(st, axis=1)
names = ["issue loads+interp", "prim_logs", "Qh/V store + surface flux + barrier", "flux dir0", "flux dir1", "colloc rhs", "viscous", "Pq+store"]
print("samples", len(st), "total cycles/wave median", np.median(st[:, 8] - st[:, 0]))
for n, col in zip(names, d.T):
    print(f"{n:40s} median {np.median(col):9.0f}  mean {col.mean():9.0f}")
