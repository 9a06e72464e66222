"""Host-side mirror of the reference's set-up surface (`SetupDG`, `CommonUtils`, `Basis1D`,
`Basis2DQuad`, `UniformQuadMesh`): same function names, argument meaning and data
conventions, re-implemented for scale (vectorised O(K) connectivity and node maps, optional
element-range construction for sharded runs).

Reference interface mirrored (file:line in yiminllin/ESDG-CNS):
  init_reference_quad   src/SetupDG.jl:205-268      RefElemData  src/SetupDG.jl:38-75
  init_mesh             src/SetupDG.jl:271-318      MeshData     src/SetupDG.jl:77-115
  uniform_quad_mesh     src/UniformQuadMesh.jl:25-50
  gauss_quad / gauss_lobatto_quad   src/Basis1D.jl:59-77, 24-47
  connect_mesh          src/connect_mesh.jl:17-36
  build_node_maps       src/node_map_functions.jl:23-55
  build_periodic_boundary_maps  src/node_map_functions.jl:66-136
  geometric_factors     src/geometric_factors.jl:16-27
  rk45_coeffs           src/CommonUtils.jl:29-49
  hybridized SBP assembly  examples/dg2D_euler_quad.jl:47-91, CompressibleNS/dg2D_CNS_cavity_optimized.jl:62-90

Conventions kept from the Julia code so a driver can be transliterated line by line:
(nodes x K) matrices are Fortran-ordered float64; EToV/FToF/mapM/mapP/mapB are int64, 1-based,
"linear index" = column-major linear index.  Operators are built from 1D Lagrange/Legendre
data with tensor products (not by inverting a 2D Vandermonde as the reference does); the
results agree with the reference construction to round-off and are checked against the
oracle in tests/.
"""
import numpy as np
from numpy.polynomial import legendre as _leg


# ---------------------------------------------------------------------------------------
# 1D rules and Lagrange operators
# ---------------------------------------------------------------------------------------
def gauss_quad(alpha, beta, N):
    """Gauss-Legendre nodes/weights with N+1 points (only alpha=beta=0 is used on quads/hexes).
    Mirrors src/Basis1D.jl:59-77 (Golub-Welsch there; Newton-polished leggauss here)."""
    if alpha != 0 or beta != 0:
        raise NotImplementedError("only Legendre (alpha=beta=0) rules are needed on quad/hex elements")
    x, w = _leg.leggauss(N + 1)
    x = 0.5 * (x - x[::-1])            # enforce exact symmetry
    w = 0.5 * (w + w[::-1])
    return x, w


def gauss_lobatto_quad(alpha, beta, N):
    """Legendre-Gauss-Lobatto nodes/weights with N+1 points (src/Basis1D.jl:24-47)."""
    if alpha != 0 or beta != 0:
        raise ValueError("alpha/beta not zero")
    if N == 0:
        return np.array([0.0]), np.array([2.0])
    if N == 1:
        return np.array([-1.0, 1.0]), np.array([1.0, 1.0])
    cN = np.zeros(N + 1)
    cN[N] = 1.0
    xi = np.sort(_leg.legroots(_leg.legder(cN)))
    for _ in range(3):                 # Newton polish on P_N'
        d1 = _leg.legval(xi, _leg.legder(cN))
        d2 = _leg.legval(xi, _leg.legder(cN, 2))
        xi = xi - d1 / d2
    x = np.concatenate(([-1.0], xi, [1.0]))
    x = 0.5 * (x - x[::-1])
    w = 2.0 / (N * (N + 1) * _leg.legval(x, cN) ** 2)
    return x, w


def _legendre_vander(x, N):
    """Orthonormal Legendre Vandermonde V[i,n] = P_n(x_i) sqrt((2n+1)/2)."""
    return _leg.legvander(np.asarray(x, dtype=float), N) * np.sqrt((2 * np.arange(N + 1) + 1) / 2.0)


def _legendre_dvander(x, N):
    x = np.asarray(x, dtype=float)
    V = np.zeros((x.size, N + 1))
    for n in range(N + 1):
        c = np.zeros(n + 1)
        c[n] = 1.0
        V[:, n] = _leg.legval(x, _leg.legder(c)) * np.sqrt((2 * n + 1) / 2.0) if n > 0 else 0.0
    return V


def lagrange_interp_1D(nodes, x):
    """Matrix L[i,k] = l_k(x_i) of the Lagrange basis on `nodes`."""
    N = len(nodes) - 1
    return np.linalg.solve(_legendre_vander(nodes, N).T, _legendre_vander(x, N).T).T


def lagrange_diff_1D(nodes):
    """Nodal differentiation matrix D[i,k] = l_k'(x_i)."""
    N = len(nodes) - 1
    return np.linalg.solve(_legendre_vander(nodes, N).T, _legendre_dvander(nodes, N).T).T


def _droptol(A, tol):
    A = np.array(A, dtype=float, copy=True)
    A[np.abs(A) <= tol] = 0.0
    return A


# ---------------------------------------------------------------------------------------
# containers
# ---------------------------------------------------------------------------------------
class RefElemData:
    """Fields as in src/SetupDG.jl:38-75 (those the hot path and its callers use)."""


class MeshData:
    """Fields as in src/SetupDG.jl:77-115, plus elem_offset / Kglobal for sharded construction."""


def quad_face_vertices():
    """src/UniformQuadMesh.jl:67-69."""
    return [1, 2], [2, 4], [3, 4], [1, 3]


def meshgrid(vx, vy=None):
    """MATLAB-style meshgrid as exported by CommonUtils (src/CommonUtils.jl:13-14)."""
    if vy is None:
        vy = vx
    X, Y = np.meshgrid(np.asarray(vx), np.asarray(vy), indexing="xy")
    return np.asfortranarray(X), np.asfortranarray(Y)


# ---------------------------------------------------------------------------------------
# reference element
# ---------------------------------------------------------------------------------------
def init_reference_quad(N, quad_nodes_1D=None):
    """RefElemData of the degree-N quad: LGL nodal basis (r fastest), tensor volume quadrature
    built from `quad_nodes_1D` (s fastest), faces ordered s=-1, r=+1, s=+1 (reversed), r=-1
    (reversed) -- the orderings of src/SetupDG.jl:231-246."""
    if quad_nodes_1D is None:
        quad_nodes_1D = gauss_quad(0, 0, N)
    r1D, w1D = (np.asarray(a, dtype=float) for a in quad_nodes_1D)
    n1 = N + 1
    nq1 = r1D.size
    rd = RefElemData()
    rd.N = N
    rd.fv = quad_face_vertices()
    rd.Nfaces = 4

    x1, _ = gauss_lobatto_quad(0, 0, N)
    rd.r = np.tile(x1, n1)                      # k = i + j*n1 : r = x1[i], s = x1[j]
    rd.s = np.repeat(x1, n1)
    D1 = lagrange_diff_1D(x1)
    I1 = np.eye(n1)
    Dr = np.kron(I1, D1)
    Ds = np.kron(D1, I1)

    rv = np.array([-1.0, 1.0, -1.0, 1.0])
    sv = np.array([-1.0, -1.0, 1.0, 1.0])
    rd.V1 = 0.25 * (1 + np.outer(rd.r, rv)) * (1 + np.outer(rd.s, sv))

    e = np.ones(nq1)
    z = np.zeros(nq1)
    rd.rf = np.concatenate([r1D, e, -r1D, -e])
    rd.sf = np.concatenate([-e, r1D, e, -r1D])
    rd.wf = np.tile(w1D, 4)
    rd.nrJ = np.concatenate([z, e, z, -e])
    rd.nsJ = np.concatenate([-e, z, e, z])

    # volume quadrature: q = a + b*nq1 with sq = r1D[a], rq = r1D[b]  (s fastest)
    rd.rq = np.repeat(r1D, nq1)
    rd.sq = np.tile(r1D, nq1)
    rd.wq = np.repeat(w1D, nq1) * np.tile(w1D, nq1)
    Iq = lagrange_interp_1D(x1, r1D)            # (nq1 x n1)
    Vq = np.einsum("bi,aj->baji", Iq, Iq).reshape(nq1 * nq1, n1 * n1)
    M = Vq.T @ (rd.wq[:, None] * Vq)
    Pq = np.linalg.solve(M, Vq.T * rd.wq[None, :])
    Lr = lagrange_interp_1D(x1, rd.rf)
    Ls = lagrange_interp_1D(x1, rd.sf)
    Vf = np.einsum("fi,fj->fji", Lr, Ls).reshape(rd.rf.size, n1 * n1)
    LIFT = np.linalg.solve(M, Vf.T * rd.wf[None, :])

    rd.Vq, rd.M, rd.Pq = Vq, M, Pq
    rd.Dr = _droptol(Dr, 1e-10)
    rd.Ds = _droptol(Ds, 1e-10)
    rd.Vf = _droptol(Vf, 1e-10)
    rd.LIFT = _droptol(LIFT, 1e-10)
    rd.r1D, rd.w1D = r1D, w1D
    return rd


# ---------------------------------------------------------------------------------------
# meshes
# ---------------------------------------------------------------------------------------
def uniform_quad_mesh(Nx, Ny):
    """Uniform Nx x Ny quad mesh of [-1,1]^2: VX, VY, EToV (K x 4, 1-based); elements numbered
    x-fastest, vertices column-major on the (Ny+1) x (Nx+1) grid (src/UniformQuadMesh.jl:25-50)."""
    Nxp, Nyp = Nx + 1, Ny + 1
    x1D = np.linspace(-1, 1, Nxp)
    y1D = np.linspace(-1, 1, Nyp)
    VX = np.repeat(x1D, Nyp)
    VY = np.tile(y1D, Nxp)
    j = np.tile(np.arange(Nx, dtype=np.int64), Ny)          # x index of element
    i = np.repeat(np.arange(Ny, dtype=np.int64), Nx)        # y index of element
    v0 = j * Nyp + i + 1
    EToV = np.stack([v0, v0 + Nyp, v0 + 1, v0 + Nyp + 1], axis=1)
    return VX, VY, EToV


def connect_mesh(EToV, fv):
    """Face-to-face connectivity FToF (Nfaces x K, 1-based linear face ids f + (e-1)*Nfaces);
    unmatched (boundary) faces map to themselves.  O(K log K) via a sort of the sorted face
    vertex pairs (the reference sorts Julia vectors of vectors, src/connect_mesh.jl:17-36)."""
    K = EToV.shape[0]
    Nfaces = len(fv)
    fvi = np.array(fv, dtype=np.int64) - 1
    nodes = np.sort(EToV[:, fvi], axis=2).reshape(K * Nfaces, -1)        # row = e*Nfaces + f
    order = np.lexsort(tuple(nodes[:, c] for c in range(nodes.shape[1] - 1, -1, -1)))
    sn = nodes[order]
    same = np.all(sn[1:] == sn[:-1], axis=1)
    FToF = np.arange(1, K * Nfaces + 1, dtype=np.int64)
    a = order[:-1][same]
    b = order[1:][same]
    FToF[a] = b + 1
    FToF[b] = a + 1
    return FToF.reshape((Nfaces, K), order="F")


def geometric_factors(x, y, Dr, Ds):
    """src/geometric_factors.jl:16-27 -> rxJ, sxJ, ryJ, syJ, J."""
    xr, xs = Dr @ x, Ds @ x
    yr, ys = Dr @ y, Ds @ y
    J = -xs * yr + xr * ys
    return ys, -yr, -xs, xr, J


def _match_face_nodes(X1, X2, tol):
    """For each face (row) find perm with X1[f,i] == X2[f,perm[f,i]] (all coordinates, to tol).
    X1, X2: (nfaces, Nfp, dim).  Conforming faces match in the same or in reversed order; anything
    else falls back to a distance-matrix search."""
    nf, Nfp, _ = X1.shape
    perm = np.tile(np.arange(Nfp, dtype=np.int64), (nf, 1))
    refd = np.abs(X1[:, :, None, :] - X2[:, None, :, :]).sum(axis=3).max(axis=(1, 2)) if nf * Nfp * Nfp < 5e7 else None
    if refd is None:     # chunk-free estimate of the reference length: max pairwise distance per face
        ext = np.abs(X1.max(axis=1) - X2.min(axis=1)).sum(axis=1)
        refd = np.maximum(ext, np.abs(X2.max(axis=1) - X1.min(axis=1)).sum(axis=1))
    d_same = np.abs(X1 - X2).sum(axis=2).max(axis=1)
    d_rev = np.abs(X1 - X2[:, ::-1, :]).sum(axis=2).max(axis=1)
    ok_same = d_same < tol * refd
    ok_rev = (~ok_same) & (d_rev < tol * refd)
    perm[ok_rev] = np.arange(Nfp - 1, -1, -1, dtype=np.int64)
    rest = np.nonzero(~(ok_same | ok_rev))[0]
    matched = np.ones((nf, Nfp), dtype=bool)
    for f in rest:
        D = np.abs(X1[f][:, None, :] - X2[f][None, :, :]).sum(axis=2)
        hit = D < tol * D.max()
        matched[f] = hit.any(axis=1)
        perm[f] = np.where(matched[f], hit.argmax(axis=1), np.arange(Nfp))
    return perm, matched


def init_mesh(VXYZ, EToV, rd, elem_range=None):
    """MeshData for elements [e0, e1) of the mesh (default: all).  Connectivity is always global,
    so md.mapP holds GLOBAL 1-based linear indices into (Nfq x Kglobal); all (nodes x K) arrays
    are local.  Mirrors init_mesh((VX,VY),EToV,rd), src/SetupDG.jl:271-318."""
    VX, VY = (np.asarray(v, dtype=float) for v in VXYZ)
    EToV = np.asarray(EToV, dtype=np.int64)
    Kg = EToV.shape[0]
    e0, e1 = (0, Kg) if elem_range is None else elem_range
    K = e1 - e0
    md = MeshData()
    Nfaces = rd.Nfaces
    Nfq = rd.Vf.shape[0]
    Nfp = Nfq // Nfaces
    FToF = connect_mesh(EToV, rd.fv)
    md.FToF_global = FToF
    md.FToF = FToF[:, e0:e1]
    md.K, md.Kglobal, md.elem_offset = K, Kg, e0
    md.VX, md.VY, md.EToV = VX, VY, EToV

    ev = EToV[e0:e1].T - 1
    x = np.asfortranarray(rd.V1 @ VX[ev])
    y = np.asfortranarray(rd.V1 @ VY[ev])
    md.x, md.y = x, y
    xf = np.asfortranarray(rd.Vf @ x)
    yf = np.asfortranarray(rd.Vf @ y)
    md.xf, md.yf = xf, yf

    # node maps (build_node_maps, src/node_map_functions.jl:23-55): neighbour face coordinates are
    # evaluated straight from the vertices so that off-range neighbours need no local storage
    f1 = np.arange(e0 * Nfaces, e1 * Nfaces, dtype=np.int64)            # 0-based global face ids
    f2 = FToF.flatten(order="F")[f1] - 1
    VfV1 = (rd.Vf @ rd.V1).reshape(Nfaces, Nfp, 4)                        # face f: (Nfp x 4) vertex weights
    en, fn = f2 // Nfaces, f2 % Nfaces
    vn = EToV[en] - 1                                                     # (nfaces, 4)
    X2 = np.stack([np.einsum("fpv,fv->fp", VfV1[fn], VX[vn]), np.einsum("fpv,fv->fp", VfV1[fn], VY[vn])], axis=2)
    X1 = np.stack([xf.T.reshape(K * Nfaces, Nfp), yf.T.reshape(K * Nfaces, Nfp)], axis=2)
    perm, matched = _match_face_nodes(X1, X2, 1e-10)
    mapM = (np.arange(e0 * Nfq, e1 * Nfq, dtype=np.int64) + 1).reshape(K * Nfaces, Nfp)
    mapP = np.where(matched, perm + (f2 * Nfp)[:, None] + 1, mapM)
    md.mapM = np.asfortranarray(mapM.reshape(K, Nfq).T)
    md.mapP = np.asfortranarray(mapP.reshape(K, Nfq).T)
    md.mapB = md.mapM.flatten(order="F")[(md.mapM == md.mapP).flatten(order="F")]

    rxJ, sxJ, ryJ, syJ, J = geometric_factors(x, y, rd.Dr, rd.Ds)
    md.rxJ, md.sxJ, md.ryJ, md.syJ, md.J = (np.asfortranarray(a) for a in (rxJ, sxJ, ryJ, syJ, J))
    md.xq = np.asfortranarray(rd.Vq @ x)
    md.yq = np.asfortranarray(rd.Vq @ y)
    md.wJq = np.asfortranarray(rd.wq[:, None] * (rd.Vq @ J))
    nxJ = (rd.Vf @ rxJ) * rd.nrJ[:, None] + (rd.Vf @ sxJ) * rd.nsJ[:, None]
    nyJ = (rd.Vf @ ryJ) * rd.nrJ[:, None] + (rd.Vf @ syJ) * rd.nsJ[:, None]
    md.nxJ, md.nyJ = np.asfortranarray(nxJ), np.asfortranarray(nyJ)
    md.sJ = np.asfortranarray(np.sqrt(nxJ ** 2 + nyJ ** 2))
    return md


def build_periodic_boundary_maps(md, rd, LX, LY):
    """Periodic partner nodes for the (local) boundary nodes md.mapB: returns mapPB with
    mapP[mapB] = mapPB, as build_periodic_boundary_maps(xf,yf,LX,LY,Nfaces*K,mapM,mapP,mapB)
    does (src/node_map_functions.jl:130-136 -> 66-128).  Boundary faces are matched globally by
    their centroids (O(Nb log Nb)) instead of the reference's O(Nb^2) double loop."""
    VX, VY, EToV = md.VX, md.VY, md.EToV
    Nfaces, Nfq = rd.Nfaces, rd.Vf.shape[0]
    Nfp = Nfq // Nfaces
    NODETOL = 1e-12
    FToFg = md.FToF_global.flatten(order="F")
    bfaces = np.nonzero(FToFg == np.arange(1, FToFg.size + 1))[0]         # global boundary faces (0-based)
    VfV1 = (rd.Vf @ rd.V1).reshape(Nfaces, Nfp, 4)
    eb, fb = bfaces // Nfaces, bfaces % Nfaces
    vb = EToV[eb] - 1
    xb = np.einsum("fpv,fv->fp", VfV1[fb], VX[vb])
    yb = np.einsum("fpv,fv->fp", VfV1[fb], VY[vb])
    xc, yc = xb.mean(axis=1), yb.mean(axis=1)
    partner = np.full(bfaces.size, -1, dtype=np.int64)

    def pair(sel_lo, sel_hi, tang, tolT):
        lo, hi = np.nonzero(sel_lo)[0], np.nonzero(sel_hi)[0]
        lo = lo[np.argsort(tang[lo], kind="stable")]
        hi = hi[np.argsort(tang[hi], kind="stable")]
        if lo.size != hi.size or np.any(np.abs(tang[lo] - tang[hi]) >= tolT):
            raise ValueError("periodic boundary faces do not pair up")
        partner[lo] = hi
        partner[hi] = lo

    on_ymin = np.abs(yc - yc.min()) < NODETOL * LY
    on_ymax = np.abs(yc - yc.max()) < NODETOL * LY
    on_xmin = np.abs(xc - xc.min()) < NODETOL * LX
    on_xmax = np.abs(xc - xc.max()) < NODETOL * LX
    pair(on_ymin, on_ymax, xc, NODETOL * LX)
    pair(on_xmin, on_xmax, yc, NODETOL * LY)
    # node matching along the tangential coordinate
    is_y = on_ymin | on_ymax
    T1 = np.where(is_y[:, None], xb, yb)[:, :, None]
    T2 = np.where(is_y[:, None], xb[partner], yb[partner])[:, :, None]
    perm, matched = _match_face_nodes(T1, T2, 1e-9)
    if not matched.all():
        raise ValueError("periodic node matching failed")
    gmapP_b = perm + (bfaces[partner] * Nfp)[:, None] + 1                 # (Nb, Nfp) global partner node ids
    # scatter to the local mapB ordering
    face_of = {int(f): i for i, f in enumerate(bfaces)}
    mapB = md.mapB
    gface = (mapB - 1) // Nfp
    gnode = (mapB - 1) % Nfp
    rows = np.array([face_of[int(f)] for f in gface], dtype=np.int64) if mapB.size else np.zeros(0, dtype=np.int64)
    return gmapP_b[rows, gnode] if mapB.size else np.zeros(0, dtype=np.int64)


def make_periodic(md, rd):
    """examples/dg2D_euler_quad.jl:38-44: LX,LY from the vertex extents, mapP[mapB] = mapPB."""
    LX = md.VX.max() - md.VX.min()
    LY = md.VY.max() - md.VY.min()
    mapPB = build_periodic_boundary_maps(md, rd, LX, LY)
    mp = md.mapP.flatten(order="F")
    mp[md.mapB - 1 - md.elem_offset * md.mapP.shape[0]] = mapPB
    md.mapP = np.asfortranarray(mp.reshape(md.mapP.shape, order="F"))
    return md


# ---------------------------------------------------------------------------------------
# time-integration coefficients
# ---------------------------------------------------------------------------------------
def rk45_coeffs():
    """Carpenter-Kennedy LSRK45 (src/CommonUtils.jl:29-49)."""
    rk4a = np.array([0.0, -567301805773.0 / 1357537059087.0, -2404267990393.0 / 2016746695238.0,
                     -3550918686646.0 / 2091501179385.0, -1275806237668.0 / 842570457699.0])
    rk4b = np.array([1432997174477.0 / 9575080441755.0, 5161836677717.0 / 13612068292357.0,
                     1720146321549.0 / 2090206949498.0, 3134564353537.0 / 4481467310338.0,
                     2277821191437.0 / 14882151754819.0])
    rk4c = np.array([0.0, 1432997174477.0 / 9575080441755.0, 2526269341429.0 / 6820363962896.0,
                     2006345519317.0 / 3224310063776.0, 2802321613138.0 / 2924317926251.0, 1.0])
    return rk4a, rk4b, rk4c


def dopri45_coeffs():
    """Dormand-Prince 5(4) tableau and error weights (dg2D_CNS_cavity_optimized.jl:919-934)."""
    rk4a = np.zeros((7, 7))
    rk4a[1, :1] = [0.2]
    rk4a[2, :2] = [3.0 / 40.0, 9.0 / 40.0]
    rk4a[3, :3] = [44.0 / 45.0, -56.0 / 15.0, 32.0 / 9.0]
    rk4a[4, :4] = [19372.0 / 6561.0, -25360.0 / 2187.0, 64448.0 / 6561.0, -212.0 / 729.0]
    rk4a[5, :5] = [9017.0 / 3168.0, -355.0 / 33.0, 46732.0 / 5247.0, 49.0 / 176.0, -5103.0 / 18656.0]
    rk4a[6, :6] = [35.0 / 384.0, 0.0, 500.0 / 1113.0, 125.0 / 192.0, -2187.0 / 6784.0, 11.0 / 84.0]
    rk4c = np.array([0.0, 0.2, 0.3, 0.8, 8.0 / 9.0, 1.0, 1.0])
    rk4E = np.array([71.0 / 57600.0, 0.0, -71.0 / 16695.0, 71.0 / 1920.0, -17253.0 / 339200.0, 22.0 / 525.0,
                     -1.0 / 40.0])
    return rk4a, rk4E, rk4c


# ---------------------------------------------------------------------------------------
# driver-level operator assembly (what the scripts do before their time loops)
# ---------------------------------------------------------------------------------------
def hybridized_sbp_ops(rd):
    """Qrhskew, Qshskew, Ef (dg2D_euler_quad.jl:47-61; cavity_optimized.jl:62-83)."""
    Qr = rd.Pq.T @ rd.M @ rd.Dr @ rd.Pq
    Qs = rd.Pq.T @ rd.M @ rd.Ds @ rd.Pq
    Ef = rd.Vf @ rd.Pq
    Br = np.diag(rd.wf * rd.nrJ)
    Bs = np.diag(rd.wf * rd.nsJ)
    Qrh = .5 * np.block([[Qr - Qr.T, Ef.T @ Br], [-Br @ Ef, Br]])
    Qsh = .5 * np.block([[Qs - Qs.T, Ef.T @ Bs], [-Bs @ Ef, Bs]])
    return .5 * (Qrh - Qrh.T), .5 * (Qsh - Qsh.T), Ef


def euler_quad_ops(rd):
    """`ops` of examples/dg2D_euler_quad.jl:47-91 as a dict (+ Ef, Vh used by the script)."""
    Qrhskew, Qshskew, Ef = hybridized_sbp_ops(rd)
    wq = rd.wq
    Vh = _droptol(np.vstack([np.eye(wq.size), Ef]), 1e-12)
    Ph = _droptol(Vh.T / wq[:, None], 1e-12)
    Lf = _droptol((Ef.T * rd.wf[None, :]) / wq[:, None], 1e-12)
    return dict(Qrhskew=Qrhskew, Qshskew=Qshskew, Qrh_sparse=_droptol(Qrhskew, 1e-12),
                Qsh_sparse=_droptol(Qshskew, 1e-12), Ph=Ph, Lf=Lf, Ef=Ef, Vh=Vh)


def cns_ops(rd):
    """`ops = (Qrhskew,Qshskew,VhP,Ph,LIFT,Vq)` of dg2D_CNS_cavity_optimized.jl:62-90 as a dict."""
    Qrhskew, Qshskew, Ef = hybridized_sbp_ops(rd)
    Vh = np.vstack([rd.Vq, rd.Vf])
    Ph = np.linalg.solve(rd.M, Vh.T)
    return dict(Qrhskew=Qrhskew, Qshskew=Qshskew, VhP=Vh @ rd.Pq, Ph=Ph, LIFT=rd.LIFT, Vq=rd.Vq, Vh=Vh, Ef=Ef)


def interp_geofacs_to_hybrid(md, Vh):
    """rxJ,sxJ,ryJ,syJ <- Vh*(.)  (dg2D_euler_quad.jl:86-88 / cavity_optimized.jl:85-87)."""
    for n in ("rxJ", "sxJ", "ryJ", "syJ"):
        setattr(md, n, np.asfortranarray(Vh @ getattr(md, n)))
    return md
