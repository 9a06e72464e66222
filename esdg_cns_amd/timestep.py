"""Device-resident time integrators: the steps either side of the RHS hot path.

  lsrk45_run  <- the LSRK45 loop of examples/dg2D_euler_quad.jl:196-212 (coefficients src/CommonUtils.jl:29-49)
  Dopri45     <- the adaptive Dormand-Prince loop of examples/CompressibleNS/dg2D_CNS_cavity_optimized.jl:974-1053
                 (tableau :919-934, Hairer error norm :1014-1021, P/PI step-size controller :1027-1037)

The state never leaves the GPU: stage combinations and the error norm run in libesdg_hip
(esdg_lsrk_update / esdg_axpy_stages / esdg_dopri_error_fields).
"""
import ctypes as C
import math

import torch

from . import setup_dg as sd
from ._lib import check_on as _chk


def lsrk45_run_graph(eng, Qd, dt, nsteps):
    """Nsteps LSRK45 steps replayed from ONE captured HIP graph of a full step (5 fused RHS + RK stages = 10-15 kernel
    launches); bitwise equal to lsrk45_run.  Measured on BASELINE cfg1 (256 elements): 66 us/step either way -- the
    loop is bound by the execution latency of ten dependent ~6 us kernels, not by host launch cost, so the graph is
    an option for hosts with slow launch paths rather than a speed-up here.  Unsharded meshes, fixed dt.
    The graph holds the device addresses of Qd and of the low-storage residual: both are kept alive ON the returned
    graph object (g.esdg_buffers), otherwise the residual would go back to torch's caching allocator on return and any
    later allocation could alias it between replays."""
    if eng.halo is not None or not eng.L.esdg_uses_tensor_kernels(eng.ctx):
        raise ValueError("graph capture needs an unsharded mesh on the tensor kernels")
    rk4a, rk4b, _ = sd.rk45_coeffs()
    resd = torch.zeros_like(Qd)
    side = torch.cuda.Stream(device=Qd.device)
    side.wait_stream(torch.cuda.current_stream(Qd.device))
    with torch.cuda.stream(side):                      # warm-up outside capture (module load, workspace touch)
        Qw, rw = Qd.clone(), resd.clone()
        for k in range(5):
            eng.rhs_lsrk_fused(Qw, rw, rk4a[k], rk4b[k], dt)
    torch.cuda.current_stream(Qd.device).wait_stream(side)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for k in range(5):
            eng.rhs_lsrk_fused(Qd, resd, rk4a[k], rk4b[k], dt)
    # the capture itself does not execute: Qd/resd are untouched so far
    g.esdg_buffers = (Qd, resd)                         # lifetime of the captured addresses = lifetime of the graph
    for _ in range(nsteps):
        g.replay()
    return g


def lsrk45_run(eng, Qd, dt, nsteps, rhstest_every=0):
    """Nsteps LSRK45 steps; returns the last rhstest if rhstest_every > 0 (computed on stage 5 like the driver)."""
    rk4a, rk4b, _ = sd.rk45_coeffs()
    resd = torch.zeros_like(Qd)
    rhsd = None
    fused = bool(eng.L.esdg_uses_tensor_kernels(eng.ctx))   # RHS + stage update in one pass, no rhs array
    rt = 0.0
    for i in range(1, nsteps + 1):
        for k in range(5):
            want_rt = rhstest_every and k == 4 and (i % rhstest_every == 0 or i == nsteps)
            if fused and not want_rt:
                eng.rhs_lsrk_fused(Qd, resd, rk4a[k], rk4b[k], dt)
                continue
            if rhsd is None:
                rhsd = torch.empty_like(Qd)
            eng.rhs_into(Qd, rhsd)
            if want_rt:
                rt = eng.rhstest(Qd, rhsd)
            eng.lsrk_update(Qd, resd, rhsd, rk4a[k], rk4b[k], dt)
    return rt


class Dopri45:
    """Adaptive DOPRI45 with FSAL exactly as the CNS drivers run it."""

    def __init__(self, eng, Qd, dt0, err_tol=1e-5, pieces=False, swap=False):
        """swap=True: an accepted step exchanges the roles of the state and the candidate buffer instead of copying the candidate
        over the state (two state-sized sweeps less per accepted step); the current state is then `self.Q`, and the tensor passed
        in is scratch from the first accepted step on.  Default: the tensor passed in is updated in place.
        pieces=True: every attempt from the library's building blocks (esdg_axpy_stages, the RHS, esdg_dopri_error_fields) instead of
        esdg_dopri45_attempt -- what a sharded engine on the torch transport always does, and the partner of the fused attempt
        in the tests."""
        self.eng, self.Q, self.dt, self.dt0, self.tol = eng, Qd, float(dt0), float(dt0), float(err_tol)
        self.pieces = bool(pieces) or not (eng.halo is None or eng.transport == "rccl")
        self.swap = bool(swap)
        self.rka, self.rkE, self.rkc = sd.dopri45_coeffs()
        self.k = [torch.zeros_like(Qd) for _ in range(7)]
        self.Qtmp = torch.empty_like(Qd)
        self.t, self.i, self.prev_err = 0.0, 0, 0.0
        self.n_rhs = 0
        eng.rhs_into(Qd, self.k[0])                     # initialise the FSAL slot (:997-998)
        self.n_rhs += 1

    def _ptrs(self, tensors):
        return (C.c_void_p * len(tensors))(*[t.data_ptr() for t in tensors])

    def step(self):
        """One attempted step; returns (accepted, errEst)."""
        eng, L = self.eng, self.eng.L
        n = self.Q.numel()
        s = eng._stream()
        if not self.pieces:
            # the library's whole attempt: on unsharded CNS contexts the stage combinations and the error norm ride in the last
            # phase of each stage (esdg_dopri45_attempt, StageFuse); on a sharded context with the library's communicator the
            # norm is reduced there
            e = C.c_double(0.0)
            _chk(L, L.esdg_dopri45_attempt(eng.ctx, C.c_void_p(self.Q.data_ptr()), C.c_void_p(self.Qtmp.data_ptr()), self._ptrs(self.k),
                                         self.dt, self.tol, C.byref(e), s))
            self.n_rhs += 6
            return self._finish(e.value)
        for INTRK in range(1, 7):                       # stages 2..7 (:1002-1012)
            coef = (C.c_double * INTRK)(*[float(self.rka[INTRK, j]) for j in range(INTRK)])
            _chk(L, L.esdg_axpy_stages(C.c_void_p(self.Qtmp.data_ptr()), C.c_void_p(self.Q.data_ptr()),
                                     self._ptrs(self.k[:INTRK]), coef, INTRK, self.dt, n, s))
            eng.rhs_into(self.Qtmp, self.k[INTRK])
            self.n_rhs += 1
        coefE = (C.c_double * 7)(*[float(x) for x in self.rkE])
        acc = C.c_double(0.0)
        nfld = int(self.Q.shape[0])                     # (the fused attempt's order: a node's fields first, then the nodes)
        _chk(L, L.esdg_dopri_error_fields(C.c_void_p(self.Q.data_ptr()), self._ptrs(self.k), coefE, 7, self.tol, n // nfld, nfld,
                                          C.byref(acc), s))
        if self.eng.nranks > 1:
            import torch.distributed as dist
            t = torch.tensor([acc.value], dtype=torch.float64, device=self.Q.device)
            dist.all_reduce(t)
            acc.value = float(t.item())
            n_glob = torch.tensor([float(n)], dtype=torch.float64, device=self.Q.device)
            dist.all_reduce(n_glob)
            n = int(n_glob.item())
        return self._finish(math.sqrt(acc.value / n))   # sqrt(sum/(length(Q[1])*4)) (:1021)

    def _finish(self, err):
        L = self.eng.L
        accepted = err < 1.0
        if accepted:
            if self.swap:
                self.Q, self.Qtmp = self.Qtmp, self.Q
            else:
                self.Q.copy_(self.Qtmp)
            self.t += self.dt
            self.k[0], self.k[6] = self.k[6], self.k[0]  # FSAL (:1025)
        # P / PI controller (:1027-1037) in the library's guarded form: err == 0 gives Inf in the reference (then
        # min(10 dt0, .)), a ZeroDivisionError in plain Python
        self.dt = float(L.esdg_dopri45_next_dt(self.dt, self.dt0, err, self.prev_err, self.i))
        self.prev_err = err
        self.i += 1
        return accepted, err

    def run(self, T, max_steps=10 ** 9):
        while self.t < T and self.i < max_steps:
            self.step()
        return self.t
