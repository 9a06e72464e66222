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


# =======================================================================================
# 3D hexahedra: host-side mirror of init_reference_hex / init_mesh (3D) / uniform_hex_mesh and of the
# operator assembly of examples/dg3D_euler_hex.jl:34-98.
#
#   init_reference_hex    src/SetupDG.jl:323-387        uniform_hex_mesh   src/UniformHexMesh.jl:25-76
#   init_mesh (3D)        src/SetupDG.jl:389-434        hex_face_vertices  src/UniformHexMesh.jl:83-93 (*)
#   geometric_factors 3D  src/geometric_factors.jl:34-67
#   build_periodic_boundary_maps (3D)  src/node_map_functions.jl:139-213
#
# (*) the reference's hex_face_vertices is the bug behind the driver's "Currently broken" banner: it
# returns CartesianIndex components instead of vertex ids.  The intended sets are used here.  The reference's
# vertex conventions are kept (reference vertices s-fastest, mesh vertices x-fastest), which makes every
# element map a reflection (J < 0); see DESIGN.md.
# =======================================================================================
def hex_face_vertices():
    """1-based local vertex ids (2x2x2 meshgrid order: s fastest, then r, then t) on r=-1, r=+1, s=-1, s=+1, t=-1, t=+1."""
    return [1, 2, 5, 6], [3, 4, 7, 8], [1, 3, 5, 7], [2, 4, 6, 8], [1, 2, 3, 4], [5, 6, 7, 8]


def _kron3(T, R, S):
    """operator acting on index n = i + n*(j + n*k) with S on i (s, fastest), R on j (r), T on k (t)."""
    return np.kron(T, np.kron(R, S))


def init_reference_hex(N, quad_nodes_1D=None):
    """RefElemData of the degree-N hexahedron.  Nodal (LGL) and quadrature nodes are both ordered s fastest, then
    r, then t (vec.(meshgrid(r1D,r1D,r1D)), src/Basis3DHex.jl:79-82, SetupDG.jl:361); faces r=-1, r=+1, s=-1, s=+1,
    t=-1, t=+1 with face nodes (rquad slow, squad fast) as in SetupDG.jl:343-352."""
    if quad_nodes_1D is None:
        quad_nodes_1D = gauss_quad(0, 0, N)
    r1D, w1D = (np.asarray(a, dtype=float) for a in quad_nodes_1D)
    n1, nq1 = N + 1, r1D.size
    rd = RefElemData()
    rd.N, rd.dim = N, 3
    rd.fv = hex_face_vertices()
    rd.Nfaces = 6
    x1, _ = gauss_lobatto_quad(0, 0, N)
    idx = np.arange(n1 ** 3)
    rd.s, rd.r, rd.t = x1[idx % n1], x1[(idx // n1) % n1], x1[idx // (n1 * n1)]
    D1, I1 = lagrange_diff_1D(x1), np.eye(n1)
    Dr, Ds, Dt = _kron3(I1, D1, I1), _kron3(I1, I1, D1), _kron3(D1, I1, I1)
    v = np.arange(8)
    sv, rv, tv = 2.0 * (v % 2) - 1, 2.0 * ((v // 2) % 2) - 1, 2.0 * (v // 4) - 1
    rd.V1 = 0.125 * (1 + np.outer(rd.r, rv)) * (1 + np.outer(rd.s, sv)) * (1 + np.outer(rd.t, tv))

    m = np.arange(nq1 * nq1)
    rquad, squad = r1D[m // nq1], r1D[m % nq1]
    wquad = w1D[m // nq1] * w1D[m % nq1]
    e, zz = np.ones(nq1 * nq1), np.zeros(nq1 * nq1)
    rd.rf = np.concatenate([-e, e, rquad, rquad, rquad, rquad])
    rd.sf = np.concatenate([rquad, rquad, -e, e, squad, squad])
    rd.tf = np.concatenate([squad, squad, squad, squad, -e, e])
    rd.wf = np.tile(wquad, 6)
    rd.nrJ = np.concatenate([-e, e, zz, zz, zz, zz])
    rd.nsJ = np.concatenate([zz, zz, -e, e, zz, zz])
    rd.ntJ = np.concatenate([zz, zz, zz, zz, -e, e])

    q = np.arange(nq1 ** 3)
    rd.sq, rd.rq, rd.tq = r1D[q % nq1], r1D[(q // nq1) % nq1], r1D[q // (nq1 * nq1)]
    rd.wq = w1D[q % nq1] * w1D[(q // nq1) % nq1] * w1D[q // (nq1 * nq1)]
    Iq = lagrange_interp_1D(x1, r1D)
    Vq = _kron3(Iq, Iq, Iq)
    M = Vq.T @ (rd.wq[:, None] * Vq)
    Pq = np.linalg.solve(M, Vq.T * rd.wq[None, :])
    Ls, Lr, Lt = lagrange_interp_1D(x1, rd.sf), lagrange_interp_1D(x1, rd.rf), lagrange_interp_1D(x1, rd.tf)
    Vf = np.einsum("fk,fj,fi->fkji", Lt, Lr, Ls).reshape(rd.rf.size, n1 ** 3)
    LIFT = np.linalg.solve(M, Vf.T * rd.wf[None, :])
    rd.Vq, rd.M, rd.Pq = Vq, M, Pq
    rd.Dr, rd.Ds, rd.Dt = _droptol(Dr, 1e-12), _droptol(Ds, 1e-12), _droptol(Dt, 1e-12)
    rd.Vf, rd.LIFT = _droptol(Vf, 1e-12), _droptol(LIFT, 1e-12)
    rd.r1D, rd.w1D = r1D, w1D
    return rd


def uniform_hex_mesh(Nx, Ny=None, Nz=None):
    """Uniform Nx x Ny x Nz hex mesh of [-1,1]^3: VX, VY, VZ, EToV (K x 8, 1-based); vertices and elements both
    numbered x fastest, then y, then z (src/UniformHexMesh.jl:25-80)."""
    Ny = Nx if Ny is None else Ny
    Nz = Nx if Nz is None else Nz
    Nxp, Nyp, Nzp = Nx + 1, Ny + 1, Nz + 1
    x1D, y1D, z1D = np.linspace(-1, 1, Nxp), np.linspace(-1, 1, Nyp), np.linspace(-1, 1, Nzp)
    VX = np.tile(x1D, Nyp * Nzp)
    VY = np.tile(np.repeat(y1D, Nxp), Nzp)
    VZ = np.repeat(z1D, Nxp * Nyp)
    e = np.arange(Nx * Ny * Nz, dtype=np.int64)
    k = e // (Nx * Ny)
    j = (e - k * Nx * Ny) // Nx
    i = e % Nx
    v0 = i + Nxp * j + Nxp * Nyp * k + 1
    EToV = np.stack([v0, v0 + 1, v0 + Nxp, v0 + Nxp + 1, v0 + Nxp * Nyp, v0 + Nxp * Nyp + 1,
                     v0 + Nxp * Nyp + Nxp, v0 + Nxp * Nyp + Nxp + 1], axis=1)
    return VX, VY, VZ, EToV


def geometric_factors_3d(x, y, z, Dr, Ds, Dt):
    """Curl-conservative metric terms of src/geometric_factors.jl:34-67 -> rxJ,sxJ,txJ,ryJ,syJ,tyJ,rzJ,szJ,tzJ,J."""
    def curl(a, b):
        Fr, Fs, Ft = (Dr @ a) * b, (Ds @ a) * b, (Dt @ a) * b
        return Dt @ Fs - Ds @ Ft, Dr @ Ft - Dt @ Fr, Ds @ Fr - Dr @ Fs
    rxJ, sxJ, txJ = curl(y, z)
    ryJ, syJ, tyJ = (-g for g in curl(x, z))
    rzJ, szJ, tzJ = (-g for g in curl(y, x))
    xr, xs, xt = Dr @ x, Ds @ x, Dt @ x
    yr, ys, yt = Dr @ y, Ds @ y, Dt @ y
    zr, zs, zt = Dr @ z, Ds @ z, Dt @ z
    J = xr * (ys * zt - zs * yt) - yr * (xs * zt - zs * xt) + zr * (xs * yt - ys * xt)
    return rxJ, sxJ, txJ, ryJ, syJ, tyJ, rzJ, szJ, tzJ, J


def _match_face_nodes_nd(X1, X2, tol, chunk=32768):
    """Like _match_face_nodes for faces with 2D node sets: identical ordering is tried first, the rest goes through
    a chunked distance-matrix search (vectorised over faces)."""
    nf, Nfp, _ = X1.shape
    perm = np.tile(np.arange(Nfp, dtype=np.int64), (nf, 1))
    matched = np.ones((nf, Nfp), dtype=bool)
    ext = np.maximum(np.abs(X1.max(axis=1) - X2.min(axis=1)).sum(axis=1), np.abs(X2.max(axis=1) - X1.min(axis=1)).sum(axis=1))
    ok_same = np.abs(X1 - X2).sum(axis=2).max(axis=1) < tol * ext
    rest = np.nonzero(~ok_same)[0]
    for c0 in range(0, rest.size, chunk):
        f = rest[c0:c0 + chunk]
        D = np.abs(X1[f][:, :, None, :] - X2[f][:, None, :, :]).sum(axis=3)
        hit = D < tol * D.max(axis=(1, 2))[:, None, None]
        matched[f] = hit.any(axis=2)
        perm[f] = np.where(matched[f], hit.argmax(axis=2), np.arange(Nfp))
    return perm, matched


def _face_coords(rd, VXYZ, EToV, gfaces):
    """physical coordinates of the face nodes of the global faces `gfaces` (0-based), straight from the vertices."""
    Nfaces = rd.Nfaces
    Nfp = rd.Vf.shape[0] // Nfaces
    nv = EToV.shape[1]
    VfV1 = (rd.Vf @ rd.V1).reshape(Nfaces, Nfp, nv)
    en, fn = gfaces // Nfaces, gfaces % Nfaces
    vn = EToV[en] - 1
    W = VfV1[fn]
    return np.stack([np.einsum("fpv,fv->fp", W, V[vn]) for V in VXYZ], axis=2)


def init_mesh_3d(VXYZ, EToV, rd, elem_range=None):
    """3D MeshData for elements [e0, e1) (default: all); md.mapP holds GLOBAL 1-based linear indices into
    (Nfq x Kglobal).  Mirrors init_mesh((VX,VY,VZ),EToV,rd), src/SetupDG.jl:389-434."""
    VX, VY, VZ = (np.asarray(v, dtype=float) for v in VXYZ)
    EToV = np.asarray(EToV, dtype=np.int64)
    Kg = EToV.shape[0]
    e0, e1 = (0, Kg) if elem_range is None else elem_range
    K = e1 - e0
    md = MeshData()
    Nfaces, Nfq = rd.Nfaces, rd.Vf.shape[0]
    Nfp = Nfq // Nfaces
    FToF = connect_mesh(EToV, rd.fv)
    md.FToF_global, md.FToF = FToF, FToF[:, e0:e1]
    md.K, md.Kglobal, md.elem_offset, md.dim = K, Kg, e0, 3
    md.VX, md.VY, md.VZ, md.EToV = VX, VY, VZ, EToV
    ev = EToV[e0:e1].T - 1
    x, y, z = (np.asfortranarray(rd.V1 @ V[ev]) for V in (VX, VY, VZ))
    md.x, md.y, md.z = x, y, z
    md.xf, md.yf, md.zf = (np.asfortranarray(rd.Vf @ a) for a in (x, y, z))
    f1 = np.arange(e0 * Nfaces, e1 * Nfaces, dtype=np.int64)
    f2 = FToF.flatten(order="F")[f1] - 1
    X1 = _face_coords(rd, (VX, VY, VZ), EToV, f1)
    X2 = _face_coords(rd, (VX, VY, VZ), EToV, f2)
    perm, matched = _match_face_nodes_nd(X1, X2, 1e-10)
    mapM = (np.arange(e0 * Nfq, e1 * Nfq, dtype=np.int64) + 1).reshape(K * Nfaces, Nfp)
    mapP = np.where(matched & (f2 != f1)[:, None], perm + (f2 * Nfp)[:, None] + 1, mapM)
    md.mapM = np.asfortranarray(mapM.reshape(K, Nfq).T)
    md.mapP = np.asfortranarray(mapP.reshape(K, Nfq).T)
    md.mapB = md.mapM.flatten(order="F")[(md.mapM == md.mapP).flatten(order="F")]
    geo = geometric_factors_3d(x, y, z, rd.Dr, rd.Ds, rd.Dt)
    (md.rxJ, md.sxJ, md.txJ, md.ryJ, md.syJ, md.tyJ, md.rzJ, md.szJ, md.tzJ, md.J) = (np.asfortranarray(a) for a in geo)
    md.xq, md.yq, md.zq = (np.asfortranarray(rd.Vq @ a) for a in (x, y, z))
    md.wJq = np.asfortranarray(rd.wq[:, None] * (rd.Vq @ md.J))
    _hex_normals(md, rd)
    return md


def _hex_normals(md, rd):
    """nxJ = nrJ.*(Vf*rxJ) + nsJ.*(Vf*sxJ) + ntJ.*(Vf*txJ) ... (SetupDG.jl:424-431); needs nodal (Np x K) metrics."""
    nr, ns, nt = rd.nrJ[:, None], rd.nsJ[:, None], rd.ntJ[:, None]
    Vf = rd.Vf
    md.nxJ = np.asfortranarray(nr * (Vf @ md.rxJ) + ns * (Vf @ md.sxJ) + nt * (Vf @ md.txJ))
    md.nyJ = np.asfortranarray(nr * (Vf @ md.ryJ) + ns * (Vf @ md.syJ) + nt * (Vf @ md.tyJ))
    md.nzJ = np.asfortranarray(nr * (Vf @ md.rzJ) + ns * (Vf @ md.szJ) + nt * (Vf @ md.tzJ))
    md.sJ = np.asfortranarray(np.sqrt(md.nxJ ** 2 + md.nyJ ** 2 + md.nzJ ** 2))


def build_periodic_boundary_maps_3d(md, rd, LX, LY, LZ):
    """Periodic partner nodes of the local boundary nodes md.mapB (mapP[mapB] = mapPB), semantics of
    src/node_map_functions.jl:139-213; boundary faces are paired globally by sorting their centroids
    (O(Nb log Nb) instead of the reference's O(Nb^2) double loop)."""
    VXYZ, EToV = (md.VX, md.VY, md.VZ), md.EToV
    Nfaces, Nfq = rd.Nfaces, rd.Vf.shape[0]
    Nfp = Nfq // Nfaces
    NODETOL = 1e-12
    L = (LX, LY, LZ)
    FToFg = md.FToF_global.flatten(order="F")
    bfaces = np.nonzero(FToFg == np.arange(1, FToFg.size + 1))[0]
    Xb = _face_coords(rd, VXYZ, EToV, bfaces)                      # (Nb, Nfp, 3)
    C = Xb.mean(axis=1)
    partner = np.full(bfaces.size, -1, dtype=np.int64)
    normal_dir = np.full(bfaces.size, -1, dtype=np.int64)
    for d in range(3):
        c = C[:, d]
        lo = np.nonzero(np.abs(c - c.min()) < NODETOL * L[d])[0]
        hi = np.nonzero(np.abs(c - c.max()) < NODETOL * L[d])[0]
        a, b = [t for t in range(3) if t != d]
        key = lambda s: np.lexsort((np.round(C[s, b] / L[b] * 1e9), np.round(C[s, a] / L[a] * 1e9)))
        lo, hi = lo[key(lo)], hi[key(hi)]
        if lo.size != hi.size or np.any(np.abs(C[lo][:, [a, b]] - C[hi][:, [a, b]]) >= 1e-9 * max(L)):
            raise ValueError("periodic boundary faces do not pair up")
        partner[lo], partner[hi] = hi, lo
        normal_dir[lo] = normal_dir[hi] = d
    if np.any(partner < 0):
        raise ValueError("boundary face off the box")
    tang = np.array([[1, 2], [0, 2], [0, 1]])[normal_dir]          # (Nb, 2)
    T1 = np.take_along_axis(Xb, tang[:, None, :], axis=2)
    T2 = np.take_along_axis(Xb[partner], tang[:, None, :], axis=2)
    perm, matched = _match_face_nodes_nd(T1, T2, 1e-9)
    if not matched.all():
        raise ValueError("periodic node matching failed")
    gmapP_b = perm + (bfaces[partner] * Nfp)[:, None] + 1
    mapB = md.mapB
    if not mapB.size:
        return np.zeros(0, dtype=np.int64)
    rows = np.searchsorted(bfaces, (mapB - 1) // Nfp)
    return gmapP_b[rows, (mapB - 1) % Nfp]


def make_periodic_3d(md, rd, LX=2.0, LY=2.0, LZ=2.0):
    """examples/dg3D_euler_hex.jl:59-65."""
    mapPB = build_periodic_boundary_maps_3d(md, rd, LX, LY, LZ)
    mp = md.mapP.flatten(order="F")
    mp[md.mapB - 1 - md.elem_offset * md.mapP.shape[0]] = mapPB
    md.mapP = np.asfortranarray(mp.reshape(md.mapP.shape, order="F"))
    return md


def hex_ops(rd):
    """Operators of examples/dg3D_euler_hex.jl:34-56, 92-98 as a dict: skew hybridized SBP matrices, Ef, and the
    quadrature-basis Vh, Ph (= 2 W^-1 Vh', note the 2), Lf."""
    M, Pq, Vf, wf, wq = rd.M, rd.Pq, rd.Vf, rd.wf, rd.wq
    Ef = Vf @ Pq
    out = {}
    for name, D, n in (("Qrhskew", rd.Dr, rd.nrJ), ("Qshskew", rd.Ds, rd.nsJ), ("Qthskew", rd.Dt, rd.ntJ)):
        Q = Pq.T @ M @ D @ Pq
        B = np.diag(wf * n)
        Qh = .5 * np.block([[Q - Q.T, Ef.T @ B], [-B @ Ef, B]])
        out[name] = .5 * (Qh - Qh.T)
    Vh = _droptol(np.vstack([np.eye(wq.size), Ef]), 1e-12)
    out.update(Ef=Ef, Vh=Vh, Ph=_droptol(2 * Vh.T / wq[:, None], 1e-12), Lf=_droptol((Ef.T * wf[None, :]) / wq[:, None], 1e-12))
    return out


def hex_driver_geometry(md, rd, hybrid=True, A3=None, a=0.0):
    """dg3D_euler_hex.jl:88-98: metrics interpolated to the hybrid nodes ([Vq;Vf]*), J and wJq at the quadrature
    nodes.  hybrid=False keeps one row per element instead of Nh (affine meshes; saves 9*Nh*K doubles of host memory
    at scale -- esdg_hex_mesh_t.geo_ld says which)."""
    if A3 is not None or a:
        # the script re-derives all geometry from the (possibly mapped) nodes x,y,z at this point (:67-90): `a` is its
        # curved mapping x,y,z += a (x^2-1)(y^2-1)(z^2-1) (:67-73), A3 an affine map (parallelepiped elements)
        dx = (md.x - 1) * (md.x + 1) * (md.y - 1) * (md.y + 1) * (md.z - 1) * (md.z + 1)
        x, y, z = md.x + a * dx, md.y + a * dx, md.z + a * dx
        if A3 is not None:
            x, y, z = (A3[i, 0] * x + A3[i, 1] * y + A3[i, 2] * z for i in range(3))
        md.x, md.y, md.z = (np.asfortranarray(a) for a in (x, y, z))
        md.xq, md.yq, md.zq = (np.asfortranarray(rd.Vq @ a) for a in (x, y, z))
        geo = geometric_factors_3d(x, y, z, rd.Dr, rd.Ds, rd.Dt)
        (md.rxJ, md.sxJ, md.txJ, md.ryJ, md.syJ, md.tyJ, md.rzJ, md.szJ, md.tzJ, md.J) = (np.asfortranarray(a) for a in geo)
        _hex_normals(md, rd)
    Vhg = np.vstack([rd.Vq, rd.Vf])
    for n in ("rxJ", "sxJ", "txJ", "ryJ", "syJ", "tyJ", "rzJ", "szJ", "tzJ"):
        g = getattr(md, n)
        setattr(md, n, np.asfortranarray(Vhg @ g) if hybrid else np.asfortranarray((rd.Vq[:1] @ g)))
    md.J = np.asfortranarray(rd.Vq @ md.J)
    md.wJq = np.asfortranarray(rd.wq[:, None] * md.J)
    return md


def error_quadrature(N, Nplus=2):
    """(Vq2, wq2) of the error blocks of the drivers: LGL nodal values of degree N -> Gauss rule of degree N+Nplus
    (`rq2,sq2,wq2 = quad_nodes_2D(N+2); Vq2 = vandermonde_2D(N,rq2,sq2)/VDM`, dg2D_euler_quad.jl:218-220).  Quadrature
    nodes in quad_nodes_2D order (r fastest, Basis2DQuad.jl:110-116), nodal columns r fastest."""
    x1, _ = gauss_lobatto_quad(0, 0, N)
    g2, w2 = gauss_quad(0, 0, N + Nplus)
    I2 = lagrange_interp_1D(x1, g2)
    n2 = g2.size
    Vq2 = np.einsum("ai,bj->baji", I2, I2).reshape(n2 * n2, (N + 1) ** 2)
    wq2 = np.repeat(w2, n2) * np.tile(w2, n2)
    return Vq2, wq2
