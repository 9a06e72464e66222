import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from esdg_cns_amd import _lib
L = _lib.lib()
L.esdg_debug_log.restype = C.c_int
L.esdg_debug_log.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p]
rng = np.random.default_rng(0)
for name, x in (("near1", 1 + 1e-3 * rng.standard_normal(1 << 20)), ("wide", np.exp(rng.uniform(-30, 30, 1 << 20))),
                ("unit", rng.uniform(0.05, 20, 1 << 20)), ("tiny-f", 1 + rng.uniform(-1e-9, 1e-9, 1 << 20))):
    xd = torch.from_numpy(x).cuda(); yd = torch.empty_like(xd)
    _lib.check(L.esdg_debug_log(xd.data_ptr(), yd.data_ptr(), xd.numel(), None)); torch.cuda.synchronize()
    y = yd.cpu().numpy()
    ref = np.log(x.astype(np.longdouble))
    err = np.abs((y.astype(np.longdouble) - ref))
    ulp = np.spacing(np.abs(ref.astype(np.float64)))
    print(name, "max err in ulps of result:", float((err / ulp).max()), " max abs err:", float(err.max()))
