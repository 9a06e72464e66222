"""Where do repeated evaluations differ?  (diagnostic for run-to-run differences of the tensor kernels)"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from common import product_cns_problem
from esdg_cns_amd import engine as E
N, Kx = int(sys.argv[1]), int(sys.argv[2])
rd, md, ops, Q = product_cns_problem(N, Kx, Kx)
eng = E.RhsEngine(rd, md, ops, E.CNS_MODAL)
Qd = eng.upload(Q)
r1 = eng.rhs(Qd).clone()
torch.cuda.synchronize()
Np = (N + 1) ** 2
for it in range(6):
    r2 = eng.rhs(Qd).clone()
    torch.cuda.synchronize()
    d = (r1 != r2) | ~torch.isfinite(r2)
    idx = d.nonzero().cpu().numpy()
    print(f"eval {it}: {len(idx)} differing entries, nonfinite {int((~torch.isfinite(r2)).sum())}")
    if len(idx):
        shape = tuple(r2.shape)
        flat = np.ravel_multi_index(idx.T, shape) % (shape[-1] * shape[-2] if len(shape) == 3 else shape[-1])
        el = np.unique(flat // Np) if len(shape) == 2 else np.unique(idx[:, 1] if shape[1] != Np else idx[:, 2])
        print("  shape", shape, "first idx", idx[:5].tolist())
        print("  elements:", el[:40].tolist(), "... count", len(el), " mod5:", np.bincount(el % 5, minlength=5).tolist())
