"""ORACLE (test infrastructure, NOT product code) -- vectorised numpy restatement of the
reference's RHS evaluations.  It is the *second*, independent restatement used to
cross-check the loop-structured C restatement in oracle/oracle_rhs.c (Julia cannot be run
in this pipeline).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this.

  euler_rhs      <- examples/dg2D_euler_quad.jl:102-194
  BCFuns         <- examples/CompressibleNS/dg2D_CNS_cavity_optimized.jl:135-265
  rhs_inviscid   <- ...cavity_optimized.jl:308-348, 447-528
  rhs_viscous    <- ...cavity_optimized.jl:548-611, 613-645, 749-849
  rhsRK          <- ...cavity_optimized.jl:955-972

All (nodes x K) arrays are float64; index maps are 1-based int64 (Julia convention).
"""
import numpy as np

from . import ref_physics as ph


def gather(x, idx):
    """Julia x[idx] with a 1-based linear (column-major) index array."""
    return x.flatten(order="F")[idx - 1]


# ------------------------------------------------------------------------------------
# Euler, collocated quad   (examples/dg2D_euler_quad.jl)
# ------------------------------------------------------------------------------------
def sparse_hadamard_sum(Qh, Qr, Qs, Qnzids, vgeo):
    """dg2D_euler_quad.jl:102-138, vectorised over elements (Qh fields are Nh x K)."""
    rxJ, sxJ, ryJ, syJ = vgeo                       # each length-K (affine: row 1)
    rho, u, v, beta = Qh
    lrho, lbeta = np.log(rho), np.log(beta)
    nrows = Qr.shape[0]
    out = [np.zeros_like(rho) for _ in range(4)]
    for i in range(nrows):
        acc = [np.zeros(rho.shape[1]) for _ in range(4)]
        for j1 in Qnzids[i]:
            j = j1 - 1
            Fx, Fy = ph.euler_fluxes_2D(rho[i], u[i], v[i], beta[i], rho[j], u[j], v[j], beta[j],
                                        lrho[i], lbeta[i], lrho[j], lbeta[j])
            for f in range(4):
                Fr = rxJ * Fx[f] + ryJ * Fy[f]
                Fs = sxJ * Fx[f] + syJ * Fy[f]
                acc[f] = acc[f] + (Qr[i, j] * Fr + Qs[i, j] * Fs)
        for f in range(4):
            out[f][i] = acc[f]
    return out


def euler_rhs(Q, md, ops, compute_rhstest=False):
    """dg2D_euler_quad.jl:141-194.  md.rxJ.. must already be the Vh-interpolated (Nh x K)
    metric arrays (script lines 86-88)."""
    Ph, Lf, Ef = ops["Ph"], ops["Lf"], ops["Ef"]
    Nq, Nh = Ph.shape
    mapP = md.mapP
    VU = ph.v_ufun(*Q)
    Uf = ph.u_vfun(*[Ef @ v for v in VU])
    rho, rhou, rhov, E = [np.vstack([q, uf]) for q, uf in zip(Q, Uf)]
    beta = ph.betafun(rho, rhou, rhov, E)
    Qh = (rho, rhou / rho, rhov / rho, beta)
    QM = [x[Nq:Nh, :] for x in Qh]
    QP = [gather(x, mapP) for x in QM]
    rhoM, rhouM, rhovM, EM = Uf
    rhoUM_n = (rhouM * md.nxJ + rhovM * md.nyJ) / md.sJ
    lam = np.abs(ph.wavespeed(rhoM, rhoUM_n, EM))
    LFc = .5 * np.maximum(lam, gather(lam, mapP)) * md.sJ
    fSx, fSy = ph.euler_fluxes_UL_UR(QM, QP)
    flux = [fx * md.nxJ + fy * md.nyJ - LFc * (gather(uf, mapP) - uf) for fx, fy, uf in zip(fSx, fSy, Uf)]
    rhsQ = [Lf @ f for f in flux]
    vgeo = (md.rxJ[0], md.sxJ[0], md.ryJ[0], md.syJ[0])
    QF = sparse_hadamard_sum(Qh, ops["Qrh_sparse"], ops["Qsh_sparse"], ops["Qrsids"], vgeo)
    rhsQ = [r + 2 * (Ph @ qf) for r, qf in zip(rhsQ, QF)]
    rhsQ = [-r / md.J for r in rhsQ]
    rhstest = 0.0
    if compute_rhstest:
        for f in range(4):
            rhstest += np.sum(md.wJq * VU[f] * rhsQ[f])
    return rhsQ, rhstest


# ------------------------------------------------------------------------------------
# CNS, modal ESDG   (examples/CompressibleNS/dg2D_CNS_cavity_optimized.jl)
# ------------------------------------------------------------------------------------
class BCFuns:
    """dg2D_CNS_cavity_optimized.jl:135-265 (init_BC_funs).  BCTYPE 1 adiabatic no-slip,
    2 isothermal, 3 slip; the boundary node lists are built from md.mapB exactly as the
    reference does (lid = boundary nodes with |y-1|<1e-12).  With an empty mapB (fully
    periodic mesh) every function is a no-op."""

    def __init__(self, md, BCTYPE, vlid=None):
        self.BCTYPE = BCTYPE
        mapB = np.asarray(md.mapB, dtype=np.int64)
        xb, yb = gather(md.xf, mapB), gather(md.yf, mapB)
        self.lid = mapB[np.abs(yb - 1) < 1e-12]
        self.wall = mapB[np.abs(yb - 1) >= 1e-12]
        self.boundary = np.concatenate([self.lid, self.wall])
        self.vlid = np.ones(self.lid.size)                 # cavity_optimized.jl:147
        if vlid is not None:                               # dg2D_CNS_convergence_test.jl:72-76: a function of xlid
            self.vlid = vlid(gather(md.xf, self.lid))
        self.nx = gather(md.nxJ, self.boundary) / gather(md.sJ, self.boundary)
        self.ny = gather(md.nyJ, self.boundary) / gather(md.sJ, self.boundary)
        self.shape = md.xf.shape

    @staticmethod
    def _set(x, idx, val):
        xl = x.flatten(order="F")
        xl[idx - 1] = val
        x[...] = xl.reshape(x.shape, order="F")

    def inviscid(self, QP, Qf):                                   # :157-176
        b = self.boundary
        if b.size == 0:
            return
        u1, u2 = gather(Qf[1], b), gather(Qf[2], b)
        Un = u1 * self.nx + u2 * self.ny
        self._set(QP[0], b, gather(Qf[0], b))
        self._set(QP[3], b, gather(Qf[3], b))
        self._set(QP[1], b, u1 - 2 * Un * self.nx)
        self._set(QP[2], b, u2 - 2 * Un * self.ny)

    def entropyvars(self, VUP, VUf):                              # :178-216
        w, l, b = self.wall, self.lid, self.boundary
        if b.size == 0:
            return
        g = gather
        if self.BCTYPE == 1:
            self._set(VUP[1], w, -g(VUf[1], w))
            self._set(VUP[2], w, -g(VUf[2], w))
            self._set(VUP[3], w, g(VUf[3], w))
            self._set(VUP[1], l, -g(VUf[1], l) - 2 * self.vlid * g(VUf[3], l))
            self._set(VUP[2], l, -g(VUf[2], l))
            self._set(VUP[3], l, g(VUf[3], l))
        elif self.BCTYPE == 2:
            theta = 1.0 / 0.3 ** 2 / 1.4 / 0.4
            self._set(VUP[1], w, -g(VUf[1], w))
            self._set(VUP[2], w, -g(VUf[2], w))
            self._set(VUP[3], w, -2.0 / theta - g(VUf[3], w))
            self._set(VUP[1], l, 2.0 / theta - g(VUf[1], l))
            self._set(VUP[2], l, -g(VUf[2], l))
            self._set(VUP[3], l, -2.0 / theta - g(VUf[3], l))
        elif self.BCTYPE == 3:
            v1, v2 = g(VUf[1], b), g(VUf[2], b)
            VUn = v1 * self.nx + v2 * self.ny
            self._set(VUP[3], b, g(VUf[3], b))
            self._set(VUP[1], b, v1 - 2 * VUn * self.nx)
            self._set(VUP[2], b, v2 - 2 * VUn * self.ny)

    def stress(self, sxP, syP, sxf, syf, VUf):                    # :218-262
        w, l, b = self.wall, self.lid, self.boundary
        if b.size == 0:
            return
        g = gather
        if self.BCTYPE == 1:
            for idx in (w, l):
                for c in (1, 2):
                    self._set(sxP[c], idx, g(sxf[c], idx))
                    self._set(syP[c], idx, g(syf[c], idx))
            self._set(sxP[3], w, -g(sxf[3], w))
            self._set(syP[3], w, -g(syf[3], w))
            self._set(sxP[3], l, -g(sxf[3], l) + 2 * self.vlid * g(sxf[1], l))
            self._set(syP[3], l, -g(syf[3], l) + 2 * self.vlid * g(syf[1], l))
        elif self.BCTYPE == 2:
            for c in (1, 2, 3):
                self._set(sxP[c], b, g(sxf[c], b))
                self._set(syP[c], b, g(syf[c], b))
        elif self.BCTYPE == 3:
            sx1, sx2 = g(sxf[1], b), g(sxf[2], b)
            sy1, sy2 = g(syf[1], b), g(syf[2], b)
            n1, n2 = self.nx, self.ny
            snx = sx1 * n1 + sx2 * n2
            sny = sy1 * n1 + sy2 * n2
            self._set(sxP[1], b, -sx1 + 2 * n1 * snx)
            self._set(syP[1], b, -sy1 + 2 * n1 * sny)
            self._set(sxP[2], b, -sx2 + 2 * n2 * snx)
            self._set(syP[2], b, -sy2 + 2 * n2 * sny)
            self._set(sxP[3], b, -g(sxf[3], b))
            self._set(syP[3], b, -g(syf[3], b))


class InflowBCFuns:
    """examples/CompressibleNS/dg2D_CNS_modalESDG.jl:161-217 (init_BC_funs of the shock-tube driver): Dirichlet inflow
    state on the x-min side, copy of the interior trace on the x-max side, lam = lamP = 0 and sigma+ = sigma- on
    both; the penalty block of that driver is commented out (:494-518).  md.mapB = the x-side boundary nodes."""
    BCTYPE = 4

    def __init__(self, md, inflow):
        mapB = np.asarray(md.mapB, dtype=np.int64)
        xb = gather(md.xf, mapB)
        self.left = mapB[np.abs(xb - md.xf.min()) < 1e-12]
        self.right = mapB[np.abs(xb - md.xf.max()) < 1e-12]
        self.vwall = np.concatenate([self.left, self.right])
        self.rhoL, self.uL, self.vL, self.pL = inflow

    def inviscid(self, QP, Qf):                                   # :168-178
        s = BCFuns._set
        s(QP[0], self.left, self.rhoL)
        s(QP[1], self.left, self.uL)
        s(QP[2], self.left, self.vL)
        s(QP[3], self.left, self.rhoL / (2 * self.pL))
        for c in range(4):
            s(QP[c], self.right, gather(Qf[c], self.right))

    def lam(self, lamP, lam):                                     # :180-185
        for a in (lam, lamP):
            BCFuns._set(a, self.vwall, 0.0)

    def entropyvars(self, VUP, VUf):                              # :187-203
        EL = self.pL / (ph.GAMMA - 1) + .5 * self.rhoL * (self.uL ** 2 + self.vL ** 2)
        VL = ph.v_ufun(np.float64(self.rhoL), np.float64(self.rhoL * self.uL), np.float64(self.rhoL * self.vL), np.float64(EL))
        for c in range(4):
            BCFuns._set(VUP[c], self.left, VL[c])
            BCFuns._set(VUP[c], self.right, gather(VUf[c], self.right))

    def stress(self, sxP, syP, sxf, syf, VUf):                    # :205-216
        for c in range(4):
            BCFuns._set(sxP[c], self.vwall, gather(sxf[c], self.vwall))
            BCFuns._set(syP[c], self.vwall, gather(syf[c], self.vwall))


def _v_hardcoded(Q):
    """dg2D_CNS_cavity_optimized.jl:461-467 (gamma literals 0.4 / 1.4 / 2.4, quirk Q5)."""
    n = Q[1] ** 2 + Q[2] ** 2
    rhoe = Q[3] - .5 * n / Q[0]
    sU = np.log(0.4 * rhoe / (Q[0] ** 1.4))
    return [(-Q[3] + rhoe * (2.4 - sU)) / rhoe, Q[1] / rhoe, Q[2] / rhoe, -Q[0] / rhoe]


def rhs_inviscid(Q, md, ops, bc, inviscid_dissp=True, compute_rhstest=False):
    """dg2D_CNS_cavity_optimized.jl:447-528 (+ update_flux! :308-324, flux_differencing! :326-348)."""
    Qrh, Qsh, VhP, Ph, Lf, Vq = (ops[k] for k in ("Qrhskew", "Qshskew", "VhP", "Ph", "LIFT", "Vq"))
    Nh, Nq = VhP.shape
    K = md.K
    mapP = md.mapP
    Qq = [Vq @ q for q in Q]
    VU = _v_hardcoded(Qq)
    Uh = [VhP @ v for v in VU]
    tmp = Uh[1] ** 2 + Uh[2] ** 2
    tmp2 = (0.4 / ((-Uh[3]) ** 1.4)) ** (1 / 0.4) * np.exp(-(1.4 - Uh[0] + tmp / (2 * Uh[3])) / 0.4)
    Uh = [tmp2 * (-Uh[3]), tmp2 * Uh[1], tmp2 * Uh[2], tmp2 * (1 - tmp / (2 * Uh[3]))]
    beta = Uh[0] / (2 * 0.4 * (Uh[3] - .5 * (Uh[1] ** 2 + Uh[2] ** 2) / Uh[0]))
    Qh = [Uh[0], Uh[1] / Uh[0], Uh[2] / Uh[0], beta]
    QM = [np.array(x[Nq:Nh, :], order="F") for x in Qh]
    QP = [np.array(gather(x, mapP), order="F") for x in QM]
    bc.inviscid(QP, QM)
    Uf = [np.array(x[Nq:Nh, :], order="F") for x in Uh]
    rhoM, rhouM, rhovM, EM = Uf
    rhoUM_n = (rhouM * md.nxJ + rhovM * md.nyJ) / md.sJ
    lam = np.abs(np.sqrt(np.abs(rhoUM_n / rhoM)) + np.sqrt(1.4 * 0.4 * (EM - .5 * rhoUM_n ** 2 / rhoM) / rhoM))
    lamP = np.array(gather(lam, mapP), order="F")
    if hasattr(bc, "lam"):
        lam = np.array(lam, order="F")
        bc.lam(lamP, lam)                                  # impose_BCs_lam! of the shock-tube driver
    LFc = .25 * np.maximum(lam, lamP) * md.sJ
    UP = [gather(x, mapP) for x in Uf]
    fx, fy = ph.euler_fluxes_UL_UR(QP, QM)                 # argument order (QP,QM), quirk Q8
    flux = []
    for d in range(4):
        f = fx[d] * md.nxJ + fy[d] * md.nyJ
        if inviscid_dissp:
            f = f - LFc * (UP[d] - Uf[d])
        flux.append(f)
    rhsQ = [Lf @ f for f in flux]
    # flux_differencing! (symmetric, dense operators, affine metrics from row 1)
    rx, ry, sx, sy = md.rxJ[0], md.ryJ[0], md.sxJ[0], md.syJ[0]
    QF = [np.zeros((Nh, K)) for _ in range(4)]
    lrho, lbeta = np.log(Qh[0]), np.log(Qh[3])
    for j in range(Nh):
        for i in range(j, Nh):
            if i < Nq or j < Nq:
                Fx, Fy = ph.euler_fluxes_2D(Qh[0][i], Qh[1][i], Qh[2][i], Qh[3][i],
                                            Qh[0][j], Qh[1][j], Qh[2][j], Qh[3][j],
                                            lrho[i], lbeta[i], lrho[j], lbeta[j])
                Qr, Qs = Qrh[i, j], Qsh[i, j]
                for d in range(4):
                    val = 2 * ((rx * Qr + sx * Qs) * Fx[d] + (ry * Qr + sy * Qs) * Fy[d])
                    QF[d][i] += val
                    QF[d][j] -= val
    rhsQ = [Ph @ qf + r for qf, r in zip(QF, rhsQ)]
    rhsQ = [-r / md.J for r in rhsQ]
    rhstest = 0.0
    if compute_rhstest:
        for f in range(4):
            rhstest += np.sum(md.wJq * VU[f][:Nq] * (Vq @ rhsQ[f]))
    return rhsQ, rhstest


def viscous_matrices(v, lam, mu, Pr):
    """dg2D_CNS_cavity_optimized.jl:613-645 (note `let lam = -lam`, quirk Q4).  Returns dense
    4x4 (x nodes) Kxx, Kxy, Kyy with the reference's zero pattern."""
    lam = -lam
    v1, v2, v3, v4 = v
    inv = 1 / (v4 ** 3)
    l2m = lam + 2.0 * mu
    z = np.zeros_like(v4)
    Kxx = [[z] * 4 for _ in range(4)]
    Kxy = [[z] * 4 for _ in range(4)]
    Kyy = [[z] * 4 for _ in range(4)]
    g = ph.GAMMA
    Kxx[1][1] = inv * -l2m * v4 ** 2
    Kxx[1][3] = inv * l2m * v2 * v4
    Kxx[2][2] = inv * -mu * v4 ** 2
    Kxx[2][3] = inv * mu * v3 * v4
    Kxx[3][1] = inv * l2m * v2 * v4
    Kxx[3][2] = inv * mu * v3 * v4
    Kxx[3][3] = inv * -(l2m * v2 ** 2 + mu * v3 ** 2 - g * mu * v4 / Pr)
    Kxy[1][2] = inv * -lam * v4 ** 2
    Kxy[1][3] = inv * lam * v3 * v4
    Kxy[2][1] = inv * -mu * v4 ** 2
    Kxy[2][3] = inv * mu * v2 * v4
    Kxy[3][1] = inv * mu * v3 * v4
    Kxy[3][2] = inv * lam * v2 * v4
    Kxy[3][3] = inv * (lam + mu) * (-v2 * v3)
    Kyy[1][1] = inv * -mu * v4 ** 2
    Kyy[1][3] = inv * mu * v2 * v4
    Kyy[2][2] = inv * -l2m * v4 ** 2
    Kyy[2][3] = inv * l2m * v3 * v4
    Kyy[3][1] = inv * mu * v2 * v4
    Kyy[3][2] = inv * l2m * v3 * v4
    Kyy[3][3] = inv * -(l2m * v3 ** 2 + mu * v2 ** 2 - g * mu * v4 / Pr)
    return Kxx, Kxy, Kyy


def rhs_viscous(Q, md, rd, bc, Re, lam, mu, Pr, viscous_dissp=True):
    """dg2D_CNS_cavity_optimized.jl:749-849 (+ dg_grad! :548-569, dg_div! :590-611)."""
    Pq, Vq, Vf, LIFT, Dr, Ds = rd.Pq, rd.Vq, rd.Vf, rd.LIFT, rd.Dr, rd.Ds
    Np = Pq.shape[0]
    mapP, mapB, J = md.mapP, np.asarray(md.mapB, dtype=np.int64), md.J
    rxj, sxj, ryj, syj = (x[:Np, :] for x in (md.rxJ, md.sxJ, md.ryJ, md.syJ))
    Qq = [Vq @ q for q in Q]
    VU = [Pq @ v for v in _v_hardcoded(Qq)]
    VUf = [np.array(Vf @ v, order="F") for v in VU]
    VUP = [np.array(gather(v, mapP), order="F") for v in VUf]
    bc.entropyvars(VUP, VUf)
    # dg_grad!
    VUx, VUy = [], []
    for d in range(4):
        ur, us = Dr @ VU[d], Ds @ VU[d]
        gx = rxj * ur + sxj * us + LIFT @ (.5 * (VUP[d] - VUf[d]) * md.nxJ)
        gy = ryj * ur + syj * us + LIFT @ (.5 * (VUP[d] - VUf[d]) * md.nyJ)
        VUx.append(gx / J)
        VUy.append(gy / J)
    VUx = [Vq @ v for v in VUx]
    VUy = [Vq @ v for v in VUy]
    VUq = [Vq @ v for v in VU]
    Kxx, Kxy, Kyy = viscous_matrices(VUq, lam, mu, Pr)
    sigma_x = [np.zeros_like(VUq[0]) for _ in range(4)]
    sigma_y = [np.zeros_like(VUq[0]) for _ in range(4)]
    for col in range(1, 4):
        for row in range(1, 4):
            sigma_x[row] = sigma_x[row] + (Kxx[row][col] * VUx[col] + Kxy[row][col] * VUy[col])
            sigma_y[row] = sigma_y[row] + (Kxy[col][row] * VUx[col] + Kyy[row][col] * VUy[col])
    rhstest = 0.0
    for f in range(4):
        rhstest += np.sum(md.wJq * VUx[f] * sigma_x[f])
        rhstest += np.sum(md.wJq * VUy[f] * sigma_y[f])
    sigma_x = [Pq @ s for s in sigma_x]
    sigma_y = [Pq @ s for s in sigma_y]
    sxf = [np.array(Vf @ s, order="F") for s in sigma_x]
    syf = [np.array(Vf @ s, order="F") for s in sigma_y]
    sxP = [np.array(gather(s, mapP), order="F") for s in sxf]
    syP = [np.array(gather(s, mapP), order="F") for s in syf]
    bc.stress(sxP, syP, sxf, syf, VUf)
    pen = None
    if viscous_dissp:
        tau = -1 / Re / VUf[3]
        dV = [a - b for a, b in zip(VUP, VUf)]
        avgV = [.5 * (a + b) for a, b in zip(VUP, VUf)]
        pen = [np.zeros_like(VUf[0]), tau * dV[1], tau * dV[2], tau * dV[3]]
        if mapB.size:
            taub = gather(tau, mapB)
            g = gather
            BCFuns._set(pen[1], mapB, taub * g(dV[1], mapB))
            BCFuns._set(pen[2], mapB, taub * g(dV[2], mapB))
            if bc.BCTYPE == 1:
                val = -taub * (g(avgV[1], mapB) * g(dV[1], mapB) + g(avgV[2], mapB) * g(dV[2], mapB)) / g(VUf[3], mapB)
            else:
                val = -taub * (g(avgV[1], mapB) * g(dV[1], mapB) + g(avgV[2], mapB) * g(dV[2], mapB)
                               + g(dV[3], mapB) * g(dV[3], mapB) / 2) / g(VUf[3], mapB)
            BCFuns._set(pen[3], mapB, val)
        pen = [LIFT @ p for p in pen]
    # dg_div!
    rhs = []
    for d in range(4):
        vol = rxj * (Dr @ sigma_x[d]) + sxj * (Ds @ sigma_x[d]) + ryj * (Dr @ sigma_y[d]) + syj * (Ds @ sigma_y[d])
        surf = LIFT @ (.5 * ((sxP[d] - sxf[d]) * md.nxJ + (syP[d] - syf[d]) * md.nyJ))
        rhs.append((vol + surf) / J)
    if viscous_dissp:
        rhs = [r + p for r, p in zip(rhs, pen)]
    return rhs, rhstest


def rhsRK(Q, rd, md, ops, bc, Re, lam, mu, Pr, inviscid_dissp=True, viscous_dissp=True):
    """dg2D_CNS_cavity_optimized.jl:955-972.  Returns rhsQ, rhstest, rhstest_visc."""
    rhsQ, _ = rhs_inviscid(Q, md, ops, bc, inviscid_dissp, False)
    visc, visc_test = rhs_viscous(Q, md, rd, bc, Re, lam, mu, Pr, viscous_dissp)
    rhsQ = [a + b for a, b in zip(rhsQ, visc)]
    Vq, Pq = rd.Vq, rd.Pq
    VU = ph.v_ufun(*[Vq @ q for q in Q])
    VUq = [Vq @ Pq @ v for v in VU]
    rhstest = 0.0
    rhstest_visc = 0.0
    for f in range(4):
        rhstest += np.sum(md.wJq * VUq[f] * (Vq @ rhsQ[f]))
        rhstest_visc += np.sum(md.wJq * VUq[f] * (Vq @ visc[f]))
    rhstest_visc += visc_test
    return rhsQ, rhstest, rhstest_visc


# ------------------------------------------------------------------------------------
# Euler, collocated hex   (examples/dg3D_euler_hex.jl)
# ------------------------------------------------------------------------------------
def sparse_hadamard_sum_hex(Qh, Qr, Qs, Qt, Qnzids, vgeo):
    """dg3D_euler_hex.jl:122-164, vectorised over elements.  vgeo: nine (Nh x K) arrays; the metric of a
    pair is the average of the two nodes' values (:145-146)."""
    rho, u, v, w, beta = Qh
    lrho, lbeta = np.log(rho), np.log(beta)
    nrows = Qr.shape[0]
    out = [np.zeros_like(rho) for _ in range(5)]
    for i in range(nrows):
        acc = [np.zeros(rho.shape[1]) for _ in range(5)]
        for j1 in Qnzids[i]:
            j = j1 - 1
            rxJa, sxJa, txJa, ryJa, syJa, tyJa, rzJa, szJa, tzJa = [.5 * (g[i] + g[j]) for g in vgeo]
            Fx, Fy, Fz = ph.euler_fluxes_3D(rho[i], u[i], v[i], w[i], beta[i], rho[j], u[j], v[j], w[j], beta[j],
                                            lrho[i], lbeta[i], lrho[j], lbeta[j])
            for f in range(5):
                Fr = rxJa * Fx[f] + ryJa * Fy[f] + rzJa * Fz[f]
                Fs = sxJa * Fx[f] + syJa * Fy[f] + szJa * Fz[f]
                Ft = txJa * Fx[f] + tyJa * Fy[f] + tzJa * Fz[f]
                acc[f] = acc[f] + (Qr[i, j] * Fr + Qs[i, j] * Fs + Qt[i, j] * Ft)
        for f in range(5):
            out[f][i] = acc[f]
    return out


def hex_rhs(Q, md, ops, compute_rhstest=False, lf_scale=0.0):
    """dg3D_euler_hex.jl:167-222.  lf_scale replaces the literal `0*.25` of :193;
    md must have been through ref_setup.hex_driver_setup (metrics at the hybrid nodes, J at quadrature nodes)."""
    Ph, Lf, Ef = ops["Ph"], ops["Lf"], ops["Ef"]
    Nq, Nh = Ph.shape
    mapP = md.mapP
    VU = ph.v_ufun_3D(*Q)
    Uf = ph.u_vfun_3D(*[Ef @ v for v in VU])
    rho, rhou, rhov, rhow, E = [np.vstack([q, uf]) for q, uf in zip(Q, Uf)]
    beta = ph.betafun_3D(rho, rhou, rhov, rhow, E)
    Qh = (rho, rhou / rho, rhov / rho, rhow / rho, beta)
    QM = [x[Nq:, :] for x in Qh]
    QP = [gather(x, mapP) for x in QM]
    rhoM, rhouM, rhovM, rhowM, EM = Uf
    rhoU_n = (rhouM * md.nxJ + rhovM * md.nyJ + rhowM * md.nzJ) / md.sJ
    lam = np.abs(ph.wavespeed(rhoM, rhoU_n, EM))
    LFc = lf_scale * np.maximum(lam, gather(lam, mapP)) * md.sJ
    fSx, fSy, fSz = ph.euler_fluxes_UL_UR_3D(QM, QP)
    flux = [fx * md.nxJ + fy * md.nyJ + fz * md.nzJ - LFc * (gather(uf, mapP) - uf)
            for fx, fy, fz, uf in zip(fSx, fSy, fSz, Uf)]
    rhsQ = [Lf @ f for f in flux]
    vgeo = (md.rxJ, md.sxJ, md.txJ, md.ryJ, md.syJ, md.tyJ, md.rzJ, md.szJ, md.tzJ)
    QF = sparse_hadamard_sum_hex(Qh, ops["Qrh_sparse"], ops["Qsh_sparse"], ops["Qth_sparse"], ops["Qnzids"], vgeo)
    rhsQ = [r + Ph @ qf for r, qf in zip(rhsQ, QF)]               # Ph already holds the factor 2 (:96)
    rhsQ = [-r / md.J for r in rhsQ]
    rhstest = 0.0
    if compute_rhstest:
        for f in range(5):
            rhstest += np.sum(md.wJq * VU[f] * rhsQ[f])
    return rhsQ, rhstest
