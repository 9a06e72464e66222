import sys, time
sys.path.insert(0, "examples"); sys.path.insert(0, ".")
import numpy as np
import dg2D_CNS_quad as drv, dg2D_euler_quad as eul
t0 = time.time()
Q, integ = drv.run("periodic", N=4, K1D=32, T=2.0, verbose=False)
print(f"CNS periodic vortex N=4 32x32 T=2: {integ.i} attempts, {integ.n_rhs} RHS, finite={all(np.isfinite(q).all() for q in Q)}, rho in [{Q[0].min():.4f},{Q[0].max():.4f}], wall {time.time()-t0:.1f}s")
t0 = time.time()
Q, integ = drv.run("cavity", N=3, K1D=24, T=1.0, BCTYPE=1, verbose=False)
print(f"CNS cavity BCTYPE=1 N=3 24x24 T=1: {integ.i} attempts, finite={all(np.isfinite(q).all() for q in Q)}, max|u| {np.abs(Q[1]/Q[0]).max():.4f}, wall {time.time()-t0:.1f}s")
t0 = time.time()
e, rt = eul.run(N=4, K1D=24, T=5.0, verbose=False)
print(f"Euler vortex N=4 K1D=24 T=5: L2err {e:.3e} rhstest {rt:.2e}, wall {time.time()-t0:.1f}s")
