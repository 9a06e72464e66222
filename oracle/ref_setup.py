"""ORACLE (test infrastructure, NOT product code) -- CPU restatement of the reference's
one-time DG set-up, statement by statement, in numpy.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this.

Every function cites the reference file:line it follows (paths relative to the
yiminllin/ESDG-CNS snapshot).  Conventions are kept Julia-like on purpose so integer
maps can be compared bit-for-bit with what the Julia driver would hold:

  * matrices that are (nodes x K) in Julia are numpy arrays of the same shape in
    Fortran order; "linear index" always means column-major linear index,
  * index arrays (EToV, FToF, mapM, mapP, mapB) are int64 and **1-based**.

Parity status of this file: the reference has no tests for src/ (SURVEY.md section 4), so
set-up parity is pinned by mathematical definitions only (quadrature exactness,
mapP involution, free-stream preservation) -- see tests/test_setup.py.
"""
import math

import numpy as np


# ----------------------------------------------------------------------------------
# helpers mirroring Julia built-ins / third-party calls used by the reference
# ----------------------------------------------------------------------------------
def meshgrid(vx, vy=None):
    """VectorizedRoutines.Matlab.meshgrid (src/CommonUtils.jl:13): MATLAB semantics,
    X[i,j] = vx[j], Y[i,j] = vy[i]; one-argument form uses vy = vx."""
    if vy is None:
        vy = vx
    X, Y = np.meshgrid(np.asarray(vx), np.asarray(vy), indexing="xy")
    return np.asfortranarray(X), np.asfortranarray(Y)


def vec(A):
    """Julia A[:] / vec(A): column-major flattening."""
    return np.asarray(A).flatten(order="F")


def droptol(A, tol):
    """SparseArrays.droptol!(sparse(A), tol) followed by re-densification: entries with
    |a| <= tol become exact zeros."""
    A = np.array(A, dtype=float, copy=True)
    A[np.abs(A) <= tol] = 0.0
    return A


def rdiv(A, B):
    """Julia A / B  ==  A * inv(B), evaluated as a solve with B' (what Julia does)."""
    return np.linalg.solve(B.T, A.T).T


# ----------------------------------------------------------------------------------
# src/Basis1D.jl
# ----------------------------------------------------------------------------------
def jacobiP(x, alpha, beta, N):
    """src/Basis1D.jl:105-138 -- orthonormal Jacobi polynomial P_N^{(alpha,beta)}(x)."""
    xp = np.asarray(x, dtype=float).reshape(-1)
    PL = np.zeros((N + 1, xp.size))
    gamma0 = (2.0 ** (alpha + beta + 1) / (alpha + beta + 1) * math.gamma(alpha + 1)
              * math.gamma(beta + 1) / math.gamma(alpha + beta + 1))
    PL[0, :] = 1.0 / math.sqrt(gamma0)
    if N == 0:
        return PL[0, :].copy()
    gamma1 = (alpha + 1) * (beta + 1) / (alpha + beta + 3) * gamma0
    PL[1, :] = ((alpha + beta + 2) * xp / 2 + (alpha - beta) / 2) / math.sqrt(gamma1)
    if N == 1:
        return PL[1, :].copy()
    aold = 2 / (2 + alpha + beta) * math.sqrt((alpha + 1) * (beta + 1) / (alpha + beta + 3))
    for i in range(1, N):
        h1 = 2 * i + alpha + beta
        anew = 2 / (h1 + 2) * math.sqrt((i + 1) * (i + 1 + alpha + beta) * (i + 1 + alpha)
                                        * (i + 1 + beta) / (h1 + 1) / (h1 + 3))
        bnew = -(alpha ** 2 - beta ** 2) / h1 / (h1 + 2)
        PL[i + 1, :] = 1 / anew * (-aold * PL[i - 1, :] + (xp - bnew) * PL[i, :])
        aold = anew
    return PL[N, :].copy()


def grad_jacobiP(r, alpha, beta, N):
    """src/Basis1D.jl:89-95."""
    r = np.asarray(r, dtype=float).reshape(-1)
    if N == 0:
        return np.zeros(r.size)
    return math.sqrt(N * (N + alpha + beta + 1)) * jacobiP(r, alpha + 1, beta + 1, N - 1)


def vandermonde_1D(N, r):
    """src/Basis1D.jl:148-154."""
    r = np.asarray(r, dtype=float).reshape(-1)
    V = np.zeros((r.size, N + 1))
    for j in range(N + 1):
        V[:, j] = jacobiP(r, 0, 0, j)
    return V


def gauss_quad(alpha, beta, N):
    """src/Basis1D.jl:59-77 -- Golub-Welsch via the symmetric eigenproblem
    (LinearAlgebra.eigen -> LAPACK; numpy.linalg.eigh is the same solver family)."""
    if N == 0:
        return np.array([-(alpha - beta) / (alpha + beta + 2)]), np.array([2.0])
    h1 = 2.0 * np.arange(0, N + 1) + alpha + beta
    with np.errstate(divide="ignore", invalid="ignore"):
        d0 = -0.5 * (alpha ** 2 - beta ** 2) / (h1 + 2) / h1
    k = np.arange(1, N + 1, dtype=float)
    d1 = 2.0 / (h1[:N] + 2) * np.sqrt(k * (k + alpha + beta) * (k + alpha) * (k + beta)
                                       / (h1[:N] + 1) / (h1[:N] + 3))
    J = np.diag(d0) + np.diag(d1, 1)
    if alpha + beta < 10 * np.finfo(float).eps:
        J[0, 0] = 0.0
    J = J + J.T
    x, V = np.linalg.eigh(J)
    w = (V[0, :] ** 2 * 2.0 ** (alpha + beta + 1) / (alpha + beta + 1) * math.gamma(alpha + 1)
         * math.gamma(beta + 1) / math.gamma(alpha + beta + 1))
    return x.copy(), w.copy()


def gauss_lobatto_quad(alpha, beta, N):
    """src/Basis1D.jl:24-47."""
    if alpha != 0 and beta != 0:
        raise ValueError("alpha/beta not zero")
    if N == 0:
        return np.array([0.0]), np.array([2.0])
    if N == 1:
        return np.array([-1.0, 1.0]), np.array([1.0, 1.0])
    xint, _ = gauss_quad(alpha + 1, beta + 1, N - 2)
    x = np.concatenate(([-1.0], xint, [1.0]))
    V = vandermonde_1D(N, x)
    w = np.sum(np.linalg.inv(V @ V.T), axis=1)
    return x, w


# ----------------------------------------------------------------------------------
# src/Basis2DQuad.jl
# ----------------------------------------------------------------------------------
def vandermonde_2D(N, r, s):
    """src/Basis2DQuad.jl:25-37 -- column sk = i*(N+1)+j holds P_i(r) P_j(s)."""
    r = np.asarray(r, dtype=float).reshape(-1)
    s = np.asarray(s, dtype=float).reshape(-1)
    V = np.zeros((r.size, (N + 1) * (N + 1)))
    sk = 0
    for i in range(N + 1):
        for j in range(N + 1):
            V[:, sk] = jacobiP(r, 0, 0, i) * jacobiP(s, 0, 0, j)
            sk += 1
    return V


def grad_vandermonde_2D(N, r, s):
    """src/Basis2DQuad.jl:48-63."""
    r = np.asarray(r, dtype=float).reshape(-1)
    s = np.asarray(s, dtype=float).reshape(-1)
    Np = (N + 1) * (N + 1)
    V2Dr = np.zeros((r.size, Np))
    V2Ds = np.zeros((r.size, Np))
    sk = 0
    for i in range(N + 1):
        for j in range(N + 1):
            V2Dr[:, sk] = grad_jacobiP(r, 0, 0, i) * jacobiP(s, 0, 0, j)
            V2Ds[:, sk] = jacobiP(r, 0, 0, i) * grad_jacobiP(s, 0, 0, j)
            sk += 1
    return V2Dr, V2Ds


def nodes_2D(N):
    """src/Basis2DQuad.jl:77-81 -- tensor LGL nodes, r fastest."""
    r1D, _ = gauss_lobatto_quad(0, 0, N)
    s, r = meshgrid(r1D)
    return vec(r), vec(s)


def equi_nodes_2D(N):
    """src/Basis2DQuad.jl:93-98."""
    r1D = np.linspace(-1, 1, N + 1)
    s, r = meshgrid(r1D)
    return vec(r), vec(s)


def quad_nodes_2D(N):
    """src/Basis2DQuad.jl:110-116 -- tensor Gauss rule, r fastest."""
    r1D, w1D = gauss_quad(0, 0, N)
    s, r = meshgrid(r1D)
    ws, wr = meshgrid(w1D)
    return vec(r), vec(s), vec(wr * ws)


# ----------------------------------------------------------------------------------
# src/UniformQuadMesh.jl
# ----------------------------------------------------------------------------------
def uniform_quad_mesh(Nx, Ny):
    """src/UniformQuadMesh.jl:25-50.  Returns VX, VY and 1-based EToV (K x 4)."""
    Nxp, Nyp = Nx + 1, Ny + 1
    K = Nx * Ny
    x1D = np.linspace(-1, 1, Nxp)
    y1D = np.linspace(-1, 1, Nyp)
    x, y = meshgrid(x1D, y1D)
    I, J = meshgrid(np.arange(1, Nxp + 1), np.arange(1, Nyp + 1))
    inds = (I - 1) * Ny + (J + I - 1)
    EToV = np.zeros((K, 4), dtype=np.int64)
    k = 0
    for i in range(Ny):
        for j in range(Nx):
            EToV[k, :] = [inds[i, j], inds[i, j + 1], inds[i + 1, j], inds[i + 1, j + 1]]
            k += 1
    return vec(x), vec(y), EToV


def quad_face_vertices():
    """src/UniformQuadMesh.jl:67-69 (1-based local vertex ids)."""
    return [1, 2], [2, 4], [3, 4], [1, 3]


# ----------------------------------------------------------------------------------
# src/connect_mesh.jl, src/node_map_functions.jl, src/geometric_factors.jl
# ----------------------------------------------------------------------------------
def connect_mesh(EToV, fv):
    """src/connect_mesh.jl:17-36.  FToF is (Nfaces x K), 1-based linear face ids."""
    Nfaces = len(fv)
    K = EToV.shape[0]
    fnodes = []
    for e in range(K):            # comprehension `for ids = fv, e = 1:K`: ids fastest
        for ids in fv:
            fnodes.append(tuple(sorted(int(EToV[e, i - 1]) for i in ids)))
    # sortperm on a Vector of Vectors: lexicographic, stable (MergeSort)
    p = sorted(range(len(fnodes)), key=lambda q: fnodes[q])
    FToF = np.arange(1, Nfaces * K + 1, dtype=np.int64)
    for f in range(len(fnodes) - 1):
        if fnodes[p[f]] == fnodes[p[f + 1]]:
            f1 = FToF[p[f]]
            f2 = FToF[p[f + 1]]
            FToF[p[f]] = f2
            FToF[p[f + 1]] = f1
    return FToF.reshape((Nfaces, K), order="F")


def build_node_maps(Xf, FToF):
    """src/node_map_functions.jl:23-55.  Xf = tuple of (Nfq x K) arrays.
    Returns mapM, mapP of shape (Nfp, Nfaces*K) and mapB (all 1-based)."""
    NfacesK = FToF.size
    NODETOL = 1e-10
    Nfp = Xf[0].size // NfacesK
    mapM = np.arange(1, Xf[0].size + 1, dtype=np.int64).reshape((Nfp, NfacesK), order="F")
    mapP = mapM.copy(order="F")
    Xfr = [vec(X).reshape((Nfp, NfacesK), order="F") for X in Xf]
    FToFl = vec(FToF)
    for f1 in range(1, NfacesK + 1):
        f2 = int(FToFl[f1 - 1])
        D = np.zeros((Nfp, Nfp))
        for Xfi in Xfr:
            X1i = Xfi[:, f1 - 1][:, None]        # repeat(Xfi[ids,f1],1,Nfp)
            X2i = Xfi[:, f2 - 1][None, :]        # transpose(repeat(Xfi[ids,f2],1,Nfp))
            D += np.abs(X1i - X2i)
        refd = D.max()
        # findall on a matrix enumerates column-major: j outer, i inner
        jj, ii = np.nonzero((D < NODETOL * refd).T)
        idM = ii + 1
        idP = jj + 1
        mapP[idM - 1, f1 - 1] = idP + (f2 - 1) * Nfp
    mapB = np.nonzero(vec(mapM) == vec(mapP))[0].astype(np.int64) + 1
    return mapM, mapP, mapB


def build_periodic_boundary_maps(xf, yf, LX, LY, NfacesTotal, mapM, mapP, mapB):
    """src/node_map_functions.jl:66-136 (2D).  Returns mapPB (1-based), to be used as
    mapP[mapB] = mapPB."""
    xfl, yfl = vec(xf), vec(yf)
    mapMl, mapPl = vec(mapM), vec(mapP)
    xb = xfl[mapB - 1]
    yb = yfl[mapB - 1]
    Nfp = xfl.size // NfacesTotal
    Nbfaces = xb.size // Nfp
    xb = xb.reshape((Nfp, Nbfaces), order="F")
    yb = yb.reshape((Nfp, Nbfaces), order="F")
    xc = xb.sum(axis=0) / Nfp
    yc = yb.sum(axis=0) / Nfp
    mapMB = mapMl[mapB - 1].reshape((Nfp, Nbfaces), order="F")
    mapPB = mapPl[mapB - 1].reshape((Nfp, Nbfaces), order="F").copy(order="F")
    xmax, xmin, ymax, ymin = xc.max(), xc.min(), yc.max(), yc.min()
    NODETOL = 1e-12
    yfaces = np.nonzero((np.abs(yc - ymax) < NODETOL * LY) | (np.abs(yc - ymin) < NODETOL * LY))[0]
    xfaces = np.nonzero((np.abs(xc - xmax) < NODETOL * LX) | (np.abs(xc - xmin) < NODETOL * LX))[0]

    def match(faces, ca, cb, La, Lb, nb, tol_len):
        for i in faces:
            for j in faces:
                if i != j:
                    if abs(ca[i] - ca[j]) < NODETOL * La and abs(abs(cb[i] - cb[j]) - Lb) < NODETOL * Lb:
                        Xa, Xb = meshgrid(nb[:, i], nb[:, j])
                        D = np.abs(Xa - Xb)
                        cols, rows = np.nonzero((D < NODETOL * tol_len).T)   # column-major findall
                        ids = rows
                        mapPB[:, i] = mapMB[ids, j]

    match(yfaces, xc, yc, LX, LY, xb, LX)   # :97-110
    match(xfaces, yc, xc, LY, LX, yb, LY)   # :113-127
    return vec(mapPB)


def geometric_factors_2D(x, y, Dr, Ds):
    """src/geometric_factors.jl:16-27."""
    xr, xs = Dr @ x, Ds @ x
    yr, ys = Dr @ y, Ds @ y
    J = -xs * yr + xr * ys
    return ys, -yr, -xs, xr, J      # rxJ, sxJ, ryJ, syJ, J


def rk45_coeffs():
    """src/CommonUtils.jl:29-49 (Carpenter-Kennedy LSRK45)."""
    rk4a = np.array([0.0,
                     -567301805773.0 / 1357537059087.0,
                     -2404267990393.0 / 2016746695238.0,
                     -3550918686646.0 / 2091501179385.0,
                     -1275806237668.0 / 842570457699.0])
    rk4b = np.array([1432997174477.0 / 9575080441755.0,
                     5161836677717.0 / 13612068292357.0,
                     1720146321549.0 / 2090206949498.0,
                     3134564353537.0 / 4481467310338.0,
                     2277821191437.0 / 14882151754819.0])
    rk4c = np.array([0.0,
                     1432997174477.0 / 9575080441755.0,
                     2526269341429.0 / 6820363962896.0,
                     2006345519317.0 / 3224310063776.0,
                     2802321613138.0 / 2924317926251.0,
                     1.0])
    return rk4a, rk4b, rk4c


def dopri45_coeffs():
    """examples/CompressibleNS/dg2D_CNS_cavity_optimized.jl:919-934."""
    rk4a = np.array([
        [0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0],
        [0.2, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0],
        [3.0 / 40.0, 9.0 / 40.0, 0.0, 0.0, 0.0, 0.0, 0.0],
        [44.0 / 45.0, -56.0 / 15.0, 32.0 / 9.0, 0.0, 0.0, 0.0, 0.0],
        [19372.0 / 6561.0, -25360.0 / 2187.0, 64448.0 / 6561.0, -212.0 / 729.0, 0.0, 0.0, 0.0],
        [9017.0 / 3168.0, -355.0 / 33.0, 46732.0 / 5247.0, 49.0 / 176.0, -5103.0 / 18656.0, 0.0, 0.0],
        [35.0 / 384.0, 0.0, 500.0 / 1113.0, 125.0 / 192.0, -2187.0 / 6784.0, 11.0 / 84.0, 0.0]])
    rk4c = np.array([0.0, 0.2, 0.3, 0.8, 8.0 / 9.0, 1.0, 1.0])
    rk4E = np.array([71.0 / 57600.0, 0.0, -71.0 / 16695.0, 71.0 / 1920.0, -17253.0 / 339200.0,
                     22.0 / 525.0, -1.0 / 40.0])
    return rk4a, rk4E, rk4c


# ----------------------------------------------------------------------------------
# src/SetupDG.jl
# ----------------------------------------------------------------------------------
class RefElemData:
    """src/SetupDG.jl:38-75 (fields filled by init_reference_quad)."""


class MeshData:
    """src/SetupDG.jl:77-115."""


def init_reference_quad(N, quad_nodes_1D=None):
    """src/SetupDG.jl:205-268."""
    if quad_nodes_1D is None:
        quad_nodes_1D = gauss_quad(0, 0, N)
    rd = RefElemData()
    rd.N = N
    rd.fv = quad_face_vertices()
    rd.Nfaces = len(rd.fv)

    r, s = nodes_2D(N)
    VDM = vandermonde_2D(N, r, s)
    Vr, Vs = grad_vandermonde_2D(N, r, s)
    Dr = rdiv(Vr, VDM)
    Ds = rdiv(Vs, VDM)
    rd.r, rd.s, rd.VDM = r, s, VDM

    r1, s1 = nodes_2D(1)
    rd.V1 = rdiv(vandermonde_2D(1, r, s), vandermonde_2D(1, r1, s1))

    r1D, w1D = (np.asarray(a, dtype=float) for a in quad_nodes_1D)
    e = np.ones(r1D.size)
    z = np.zeros(r1D.size)
    rd.rf = np.concatenate([r1D, e, -r1D, -e])
    rd.sf = np.concatenate([-e, r1D, e, -r1D])
    rd.wf = np.tile(w1D, rd.Nfaces)
    rd.nrJ = np.concatenate([z, e, z, -e])
    rd.nsJ = np.concatenate([-e, z, e, z])

    rq, sq = (vec(a) for a in meshgrid(r1D))      # rq = X[:], sq = Y[:]: s fastest
    wr, ws = (vec(a) for a in meshgrid(w1D))
    wq = wr * ws
    Vq = rdiv(vandermonde_2D(N, rq, sq), VDM)
    M = Vq.T @ np.diag(wq) @ Vq
    Pq = np.linalg.solve(M, Vq.T @ np.diag(wq))
    rd.rq, rd.sq, rd.wq, rd.Vq, rd.M, rd.Pq = rq, sq, wq, Vq, M, Pq

    Vf = rdiv(vandermonde_2D(N, rd.rf, rd.sf), VDM)
    LIFT = np.linalg.solve(M, Vf.T @ np.diag(rd.wf))

    rd.Dr = droptol(Dr, 1e-10)
    rd.Ds = droptol(Ds, 1e-10)
    rd.Vf = droptol(Vf, 1e-10)
    rd.LIFT = droptol(LIFT, 1e-10)
    return rd


def init_mesh_2D(VX, VY, EToV, rd):
    """src/SetupDG.jl:275-318."""
    md = MeshData()
    FToF = connect_mesh(EToV, rd.fv)
    Nfaces, K = FToF.shape
    md.FToF, md.K, md.VX, md.VY, md.EToV = FToF, K, VX, VY, EToV

    x = np.asfortranarray(rd.V1 @ VX[EToV.T - 1])
    y = np.asfortranarray(rd.V1 @ VY[EToV.T - 1])
    md.x, md.y = x, y

    xf = np.asfortranarray(rd.Vf @ x)
    yf = np.asfortranarray(rd.Vf @ y)
    mapM, mapP, mapB = build_node_maps((xf, yf), FToF)
    Nfp = rd.Vf.shape[0] // Nfaces
    md.mapM = mapM.reshape((Nfp * Nfaces, K), order="F")
    md.mapP = mapP.reshape((Nfp * Nfaces, K), order="F")
    md.mapB = mapB
    md.xf, md.yf = xf, yf

    rxJ, sxJ, ryJ, syJ, J = geometric_factors_2D(x, y, rd.Dr, rd.Ds)
    md.rxJ, md.sxJ, md.ryJ, md.syJ, md.J = (np.asfortranarray(a) for a in (rxJ, sxJ, ryJ, syJ, J))

    md.xq = np.asfortranarray(rd.Vq @ x)
    md.yq = np.asfortranarray(rd.Vq @ y)
    md.wJq = np.asfortranarray(np.diag(rd.wq) @ (rd.Vq @ J))

    nxJ = (rd.Vf @ rxJ) * rd.nrJ[:, None] + (rd.Vf @ sxJ) * rd.nsJ[:, None]
    nyJ = (rd.Vf @ ryJ) * rd.nrJ[:, None] + (rd.Vf @ syJ) * rd.nsJ[:, None]
    md.nxJ = np.asfortranarray(nxJ)
    md.nyJ = np.asfortranarray(nyJ)
    md.sJ = np.asfortranarray(np.sqrt(nxJ ** 2 + nyJ ** 2))
    return md


def make_periodic_2D(md, rd, VX, VY):
    """examples/dg2D_euler_quad.jl:38-44: patch mapP with the periodic partner nodes."""
    LX = VX.max() - VX.min()
    LY = VY.max() - VY.min()
    mapPB = build_periodic_boundary_maps(md.xf, md.yf, LX, LY, rd.Nfaces * md.K, md.mapM, md.mapP, md.mapB)
    mapPl = vec(md.mapP)
    mapPl[md.mapB - 1] = mapPB
    md.mapP = mapPl.reshape(md.mapP.shape, order="F")
    return md


# ----------------------------------------------------------------------------------
# driver-level SBP assembly
# ----------------------------------------------------------------------------------
def hybridized_sbp_ops(rd):
    """examples/dg2D_euler_quad.jl:47-63 == dg2D_CNS_cavity_optimized.jl:62-83."""
    M, Dr, Ds, Pq, Vf = rd.M, rd.Dr, rd.Ds, rd.Pq, rd.Vf
    Qr = Pq.T @ M @ Dr @ Pq
    Qs = Pq.T @ M @ Ds @ Pq
    Ef = Vf @ Pq
    Br = np.diag(rd.wf * rd.nrJ)
    Bs = np.diag(rd.wf * rd.nsJ)
    Qrh = 0.5 * np.block([[Qr - Qr.T, Ef.T @ Br], [-Br @ Ef, Br]])
    Qsh = 0.5 * np.block([[Qs - Qs.T, Ef.T @ Bs], [-Bs @ Ef, Bs]])
    Qrhskew = 0.5 * (Qrh - Qrh.T)
    Qshskew = 0.5 * (Qsh - Qsh.T)
    return Qrhskew, Qshskew, Ef


def euler_quad_ops(rd):
    """examples/dg2D_euler_quad.jl:47-91: operators of the collocated Euler-quad driver.
    Qrsids[i] = per-row union of nonzero column ids (1-based, Qr's first then Qs's new ones)."""
    Qrhskew, Qshskew, Ef = hybridized_sbp_ops(rd)
    Qrh_sparse = droptol(Qrhskew, 1e-12)
    Qsh_sparse = droptol(Qshskew, 1e-12)
    Qrsids = []
    for i in range(Qrhskew.shape[0]):
        a = list(np.nonzero(Qrh_sparse[i, :])[0] + 1)
        b = list(np.nonzero(Qsh_sparse[i, :])[0] + 1)
        ids = []
        for c in a + b:                      # unique() keeps first occurrences in order
            if c not in ids:
                ids.append(int(c))
        Qrsids.append(ids)
    wq = rd.wq
    Vh = droptol(np.vstack([np.eye(wq.size), Ef]), 1e-12)
    Ph = droptol(np.diag(1.0 / wq) @ Vh.T, 1e-12)
    Lf = droptol(np.diag(1.0 / wq) @ (Ef.T @ np.diag(rd.wf)), 1e-12)
    return dict(Qrhskew=Qrhskew, Qshskew=Qshskew, Qrh_sparse=Qrh_sparse, Qsh_sparse=Qsh_sparse,
                Qrsids=Qrsids, Ph=Ph, Lf=Lf, Ef=Ef, Vh=Vh)


def cns_ops(rd):
    """examples/CompressibleNS/dg2D_CNS_cavity_optimized.jl:62-90 (modal ESDG operators)."""
    Qrhskew, Qshskew, Ef = hybridized_sbp_ops(rd)
    Vh = np.vstack([rd.Vq, rd.Vf])
    Ph = np.linalg.solve(rd.M, Vh.T)
    VhP = Vh @ rd.Pq
    return dict(Qrhskew=Qrhskew, Qshskew=Qshskew, VhP=VhP, Ph=Ph, LIFT=rd.LIFT, Vq=rd.Vq, Vh=Vh, Ef=Ef)


# ==================================================================================
# 3D hexahedra (examples/dg3D_euler_hex.jl and the src/ routines it calls)
#
# The reference flags this driver "TODO: FIX. Currently broken" (dg3D_euler_hex.jl:1).  What is
# broken is src/UniformHexMesh.jl:83-93: `map(x->x[1], findall(...))` on 3-D arrays takes the first
# component of each CartesianIndex{3} instead of the linear vertex id, so `fv` holds ids in {1,2}
# and connect_mesh finds nonsense.  hex_face_vertices() below returns the INTENDED sets (the linear
# ids of the vertices with r=-1, r=+1, s=-1, s=+1, t=-1, t=+1 in meshgrid order), as SURVEY.md
# section 8 (cfg5) prescribes; everything else is restated as written.  Consequence kept on
# purpose: with the reference's vertex order (meshgrid: s fastest) against EToV's (x fastest) the
# element map is a reflection, J < 0 on every element.  -(.)/J still gives the right sign for the
# volume and central surface terms; only an LF penalty would become anti-dissipative, and the
# reference multiplies it by 0 (dg3D_euler_hex.jl:193).
# ==================================================================================
def meshgrid3(vx, vy, vz):
    """VectorizedRoutines.Matlab.meshgrid, 3-argument form: X[i,j,k]=vx[j], Y[i,j,k]=vy[i], Z[i,j,k]=vz[k]."""
    vx, vy, vz = (np.asarray(a, dtype=float) for a in (vx, vy, vz))
    X = np.empty((vy.size, vx.size, vz.size))
    Y = np.empty_like(X)
    Z = np.empty_like(X)
    X[:] = vx[None, :, None]
    Y[:] = vy[:, None, None]
    Z[:] = vz[None, None, :]
    return X, Y, Z


def vandermonde_3D(N, r, s, t):
    """src/Basis3DHex.jl:24-39."""
    r, s, t = (np.asarray(a, dtype=float).reshape(-1) for a in (r, s, t))
    V = np.zeros((r.size, (N + 1) ** 3))
    sk = 0
    for i in range(N + 1):
        for j in range(N + 1):
            for k in range(N + 1):
                V[:, sk] = jacobiP(r, 0, 0, i) * jacobiP(s, 0, 0, j) * jacobiP(t, 0, 0, k)
                sk += 1
    return V


def grad_vandermonde_3D(N, r, s, t):
    """src/Basis3DHex.jl:47-67."""
    r, s, t = (np.asarray(a, dtype=float).reshape(-1) for a in (r, s, t))
    Np = (N + 1) ** 3
    Vr, Vs, Vt = np.zeros((r.size, Np)), np.zeros((r.size, Np)), np.zeros((r.size, Np))
    sk = 0
    for i in range(N + 1):
        for j in range(N + 1):
            for k in range(N + 1):
                Vr[:, sk] = grad_jacobiP(r, 0, 0, i) * jacobiP(s, 0, 0, j) * jacobiP(t, 0, 0, k)
                Vs[:, sk] = jacobiP(r, 0, 0, i) * grad_jacobiP(s, 0, 0, j) * jacobiP(t, 0, 0, k)
                Vt[:, sk] = jacobiP(r, 0, 0, i) * jacobiP(s, 0, 0, j) * grad_jacobiP(t, 0, 0, k)
                sk += 1
    return Vr, Vs, Vt


def nodes_3D(N):
    """src/Basis3DHex.jl:79-82."""
    r1D, _ = gauss_lobatto_quad(0, 0, N)
    return tuple(vec(a) for a in meshgrid3(r1D, r1D, r1D))


def uniform_hex_mesh(Nx, Ny, Nz):
    """src/UniformHexMesh.jl:25-76.  Returns VX, VY, VZ and 1-based EToV (K x 8); vertices x fastest."""
    Nxp, Nyp, Nzp = Nx + 1, Ny + 1, Nz + 1
    K = Nx * Ny * Nz
    x1D, y1D, z1D = np.linspace(-1, 1, Nxp), np.linspace(-1, 1, Nyp), np.linspace(-1, 1, Nzp)
    x = np.zeros(Nxp * Nyp * Nzp)
    y = np.zeros_like(x)
    z = np.zeros_like(x)
    sk = 0
    for k in range(Nzp):
        for j in range(Nyp):
            for i in range(Nxp):
                x[sk], y[sk], z[sk] = x1D[i], y1D[j], z1D[k]
                sk += 1
    EToV = np.zeros((K, 8), dtype=np.int64)
    for e in range(1, K + 1):
        em = e - 1
        k = em // (Nx * Ny)
        j = (em - k * Nx * Ny) // Nx
        i = em % Nx
        for c, (di, dj, dk) in enumerate([(0, 0, 0), (1, 0, 0), (0, 1, 0), (1, 1, 0), (0, 0, 1), (1, 0, 1), (0, 1, 1), (1, 1, 1)]):
            EToV[e - 1, c] = (i + di) + Nxp * (j + dj) + Nxp * Nyp * (k + dk)
    return x, y, z, EToV + 1


def hex_face_vertices():
    """INTENDED result of src/UniformHexMesh.jl:83-93 (see the section header): 1-based linear ids, in the
    2x2x2 meshgrid ordering (s fastest, then r, then t), of the vertices on r=-1, r=+1, s=-1, s=+1, t=-1, t=+1."""
    x1D = np.array([-1.0, 1.0])
    r, s, t = (vec(a) for a in meshgrid3(x1D, x1D, x1D))
    return tuple([int(i) + 1 for i in np.nonzero(np.abs(c - v) < 1e-10)[0]]
                 for c, v in ((r, -1), (r, 1), (s, -1), (s, 1), (t, -1), (t, 1)))


def geometric_factors_3D(x, y, z, Dr, Ds, Dt):
    """src/geometric_factors.jl:34-67 (curl-conservative form, identity filters)."""
    xr, xs, xt = Dr @ x, Ds @ x, Dt @ x
    yr, ys, yt = Dr @ y, Ds @ y, Dt @ y
    zr, zs, zt = Dr @ z, Ds @ z, Dt @ z
    Fr, Fs, Ft = (Dr @ y) * z, (Ds @ y) * z, (Dt @ y) * z
    rxJ = Dt @ Fs - Ds @ Ft
    sxJ = Dr @ Ft - Dt @ Fr
    txJ = Ds @ Fr - Dr @ Fs
    Fr, Fs, Ft = (Dr @ x) * z, (Ds @ x) * z, (Dt @ x) * z
    ryJ = -(Dt @ Fs - Ds @ Ft)
    syJ = -(Dr @ Ft - Dt @ Fr)
    tyJ = -(Ds @ Fr - Dr @ Fs)
    Fr, Fs, Ft = (Dr @ y) * x, (Ds @ y) * x, (Dt @ y) * x
    rzJ = -(Dt @ Fs - Ds @ Ft)
    szJ = -(Dr @ Ft - Dt @ Fr)
    tzJ = -(Ds @ Fr - Dr @ Fs)
    J = xr * (ys * zt - zs * yt) - yr * (xs * zt - zs * xt) + zr * (xs * yt - ys * xt)
    return rxJ, sxJ, txJ, ryJ, syJ, tyJ, rzJ, szJ, tzJ, J


def build_periodic_boundary_maps_3D(xf, yf, zf, LX, LY, LZ, NfacesTotal, mapM, mapP, mapB):
    """src/node_map_functions.jl:139-213."""
    xfl, yfl, zfl = vec(xf), vec(yf), vec(zf)
    mapMl, mapPl = vec(mapM), vec(mapP)
    Nfp = xfl.size // NfacesTotal
    Nbfaces = mapB.size // Nfp
    xb, yb, zb = (a[mapB - 1].reshape((Nfp, Nbfaces), order="F") for a in (xfl, yfl, zfl))
    xc, yc, zc = xb.sum(axis=0) / Nfp, yb.sum(axis=0) / Nfp, zb.sum(axis=0) / Nfp
    mapMB = mapMl[mapB - 1].reshape((Nfp, Nbfaces), order="F")
    mapPB = mapPl[mapB - 1].reshape((Nfp, Nbfaces), order="F").copy(order="F")
    NODETOL = 1e-12
    on = lambda c, L: np.nonzero((np.abs(c - c.max()) < NODETOL * L) | (np.abs(c - c.min()) < NODETOL * L))[0]
    xfaces, yfaces, zfaces = on(xc, LX), on(yc, LY), on(zc, LZ)

    def match(faces, cn, Ln, ca, La, cb, Lb, na, nb, tol_len):
        # faces matched across direction n; tangential coordinates a, b
        for i in faces:
            for j in faces:
                if i != j:
                    if abs(ca[i] - ca[j]) < NODETOL * La and abs(cb[i] - cb[j]) < NODETOL * Lb and \
                            abs(abs(cn[i] - cn[j]) - Ln) < NODETOL * Ln:
                        Aa, Ab = meshgrid(na[:, i], na[:, j])
                        Ba, Bb = meshgrid(nb[:, i], nb[:, j])
                        D = np.abs(Aa - Ab) + np.abs(Ba - Bb)
                        cols, rows = np.nonzero((D < NODETOL * tol_len).T)   # column-major findall, x[1] = row
                        mapPB[:, i] = mapMB[rows, j]

    match(xfaces, xc, LX, yc, LY, zc, LZ, yb, zb, LY)   # :165-177
    match(yfaces, yc, LY, xc, LX, zc, LZ, xb, zb, LX)   # :180-192
    match(zfaces, zc, LZ, xc, LX, yc, LY, xb, yb, LX)   # :195-207
    return vec(mapPB)


def init_reference_hex(N, quad_nodes_1D=None):
    """src/SetupDG.jl:323-387."""
    if quad_nodes_1D is None:
        quad_nodes_1D = gauss_quad(0, 0, N)
    rd = RefElemData()
    rd.N = N
    rd.fv = hex_face_vertices()
    rd.Nfaces = len(rd.fv)
    r, s, t = nodes_3D(N)
    VDM = vandermonde_3D(N, r, s, t)
    Vr, Vs, Vt = grad_vandermonde_3D(N, r, s, t)
    Dr, Ds, Dt = rdiv(Vr, VDM), rdiv(Vs, VDM), rdiv(Vt, VDM)
    rd.r, rd.s, rd.t, rd.VDM = r, s, t, VDM
    r1, s1, t1 = nodes_3D(1)
    rd.V1 = rdiv(vandermonde_3D(1, r, s, t), vandermonde_3D(1, r1, s1, t1))

    r1D, w1D = (np.asarray(a, dtype=float) for a in quad_nodes_1D)
    rquad, squad = (vec(a) for a in meshgrid(r1D, r1D))
    wr, ws = (vec(a) for a in meshgrid(w1D, w1D))
    wquad = wr * ws
    e = np.ones(rquad.size)
    zz = np.zeros(rquad.size)
    rd.rf = np.concatenate([-e, e, rquad, rquad, rquad, rquad])
    rd.sf = np.concatenate([rquad, rquad, -e, e, squad, squad])
    rd.tf = np.concatenate([squad, squad, squad, squad, -e, e])
    rd.wf = np.tile(wquad, rd.Nfaces)
    rd.nrJ = np.concatenate([-e, e, zz, zz, zz, zz])
    rd.nsJ = np.concatenate([zz, zz, -e, e, zz, zz])
    rd.ntJ = np.concatenate([zz, zz, zz, zz, -e, e])

    rq, sq, tq = (vec(a) for a in meshgrid3(r1D, r1D, r1D))
    wr, ws, wt = (vec(a) for a in meshgrid3(w1D, w1D, w1D))
    wq = wr * ws * wt
    Vq = rdiv(vandermonde_3D(N, rq, sq, tq), VDM)
    M = Vq.T @ np.diag(wq) @ Vq
    Pq = np.linalg.solve(M, Vq.T @ np.diag(wq))
    rd.rq, rd.sq, rd.tq, rd.wq, rd.Vq, rd.M, rd.Pq = rq, sq, tq, wq, Vq, M, Pq

    Vf = rdiv(vandermonde_3D(N, rd.rf, rd.sf, rd.tf), VDM)
    LIFT = np.linalg.solve(M, Vf.T @ np.diag(rd.wf))
    rd.Dr, rd.Ds, rd.Dt = droptol(Dr, 1e-12), droptol(Ds, 1e-12), droptol(Dt, 1e-12)
    rd.Vf = droptol(Vf, 1e-12)
    rd.LIFT = droptol(LIFT, 1e-12)
    return rd


def init_mesh_3D(VX, VY, VZ, EToV, rd):
    """src/SetupDG.jl:389-434."""
    md = MeshData()
    FToF = connect_mesh(EToV, rd.fv)
    Nfaces, K = FToF.shape
    md.FToF, md.K, md.VX, md.VY, md.VZ, md.EToV = FToF, K, VX, VY, VZ, EToV
    x, y, z = (np.asfortranarray(rd.V1 @ V[EToV.T - 1]) for V in (VX, VY, VZ))
    md.x, md.y, md.z = x, y, z
    xf, yf, zf = (np.asfortranarray(rd.Vf @ a) for a in (x, y, z))
    mapM, mapP, mapB = build_node_maps((xf, yf, zf), FToF)
    Nfp = rd.Vf.shape[0] // Nfaces
    md.mapM = mapM.reshape((Nfp * Nfaces, K), order="F")
    md.mapP = mapP.reshape((Nfp * Nfaces, K), order="F")
    md.mapB = mapB
    md.xf, md.yf, md.zf = xf, yf, zf
    geo = geometric_factors_3D(x, y, z, rd.Dr, rd.Ds, rd.Dt)
    (md.rxJ, md.sxJ, md.txJ, md.ryJ, md.syJ, md.tyJ, md.rzJ, md.szJ, md.tzJ, md.J) = (np.asfortranarray(a) for a in geo)
    md.xq, md.yq, md.zq = (np.asfortranarray(rd.Vq @ a) for a in (x, y, z))
    md.wJq = np.asfortranarray(np.diag(rd.wq) @ (rd.Vq @ md.J))
    _hex_normals(md, rd, geo)
    return md


def _hex_normals(md, rd, geo):
    """src/SetupDG.jl:424-431 == dg3D_euler_hex.jl:81-86."""
    rxJ, sxJ, txJ, ryJ, syJ, tyJ, rzJ, szJ, tzJ = geo[:9]
    nr, ns, nt = rd.nrJ[:, None], rd.nsJ[:, None], rd.ntJ[:, None]
    Vf = rd.Vf
    md.nxJ = np.asfortranarray(nr * (Vf @ rxJ) + ns * (Vf @ sxJ) + nt * (Vf @ txJ))
    md.nyJ = np.asfortranarray(nr * (Vf @ ryJ) + ns * (Vf @ syJ) + nt * (Vf @ tyJ))
    md.nzJ = np.asfortranarray(nr * (Vf @ rzJ) + ns * (Vf @ szJ) + nt * (Vf @ tzJ))
    md.sJ = np.asfortranarray(np.sqrt(md.nxJ ** 2 + md.nyJ ** 2 + md.nzJ ** 2))


def make_periodic_3D(md, rd, LX=2.0, LY=2.0, LZ=2.0):
    """examples/dg3D_euler_hex.jl:59-65 (LX=LY=LZ=2 literals for the [-1,1]^3 box)."""
    mapPB = build_periodic_boundary_maps_3D(md.xf, md.yf, md.zf, LX, LY, LZ, rd.Nfaces * md.K, md.mapM, md.mapP, md.mapB)
    mapPl = vec(md.mapP)
    mapPl[md.mapB - 1] = mapPB
    md.mapP = mapPl.reshape(md.mapP.shape, order="F")
    return md


def hex_driver_setup(md, rd, a=0.0, A3=None):
    """examples/dg3D_euler_hex.jl:34-98: hybridized SBP operators in the quadrature basis, the (optionally curved)
    geometry re-computation, metrics interpolated to the hybrid nodes, J and wJq at the quadrature nodes.
    Mutates md like the script does and returns the `ops` dictionary."""
    M, Dr, Ds, Dt, Pq, Vq, Vf, wf = rd.M, rd.Dr, rd.Ds, rd.Dt, rd.Pq, rd.Vq, rd.Vf, rd.wf
    Qr, Qs, Qt = Pq.T @ M @ Dr @ Pq, Pq.T @ M @ Ds @ Pq, Pq.T @ M @ Dt @ Pq
    Ef = Vf @ Pq
    sk = []
    for Q, n in ((Qr, rd.nrJ), (Qs, rd.nsJ), (Qt, rd.ntJ)):
        B = np.diag(wf * n)
        Qh = .5 * np.block([[Q - Q.T, Ef.T @ B], [-B @ Ef, B]])
        sk.append(.5 * (Qh - Qh.T))
    Qrhskew, Qshskew, Qthskew = sk
    Qrs, Qss, Qts = (droptol(A, 1e-12) for A in sk)
    Qnzids = []
    for i in range(Qrhskew.shape[0]):
        ids = []
        for A in (Qrs, Qss, Qts):
            for c in np.nonzero(A[i, :])[0] + 1:
                if c not in ids:
                    ids.append(int(c))
        Qnzids.append(ids)
    # curved mapping (:67-73) and geometry re-computation (:75-90)
    x, y, z = md.x, md.y, md.z
    dx = (x - 1) * (x + 1) * (y - 1) * (y + 1) * (z - 1) * (z + 1)
    x, y, z = x + a * dx, y + a * dx, z + a * dx
    if A3 is not None:       # affine map of the nodes (the script re-derives all geometry from x,y,z here, :75-90), for tests
        x, y, z = (A3[i, 0] * x + A3[i, 1] * y + A3[i, 2] * z for i in range(3))
        md.x, md.y, md.z = x, y, z
    md.xq, md.yq, md.zq = (np.asfortranarray(Vq @ c) for c in (x, y, z))
    vgeo = geometric_factors_3D(x, y, z, Dr, Ds, Dt)
    _hex_normals(md, rd, vgeo)
    Vhg = np.vstack([Vq, Vf])
    (md.rxJ, md.sxJ, md.txJ, md.ryJ, md.syJ, md.tyJ, md.rzJ, md.szJ, md.tzJ) = (np.asfortranarray(Vhg @ g) for g in vgeo[:9])
    wq = rd.wq
    Vh = droptol(np.vstack([np.eye(wq.size), Ef]), 1e-12)
    Ph = droptol(2 * np.diag(1.0 / wq) @ Vh.T, 1e-12)              # note the factor 2 (:96, quirk Q9)
    Lf = droptol(np.diag(1.0 / wq) @ (Ef.T @ np.diag(wf)), 1e-12)
    md.J = np.asfortranarray(Vq @ vgeo[9])
    md.wJq = np.asfortranarray(np.diag(wq) @ md.J)
    return dict(Qrhskew=Qrhskew, Qshskew=Qshskew, Qthskew=Qthskew, Qrh_sparse=Qrs, Qsh_sparse=Qss, Qth_sparse=Qts,
                Qnzids=Qnzids, Ph=Ph, Lf=Lf, Ef=Ef, Vh=Vh)
