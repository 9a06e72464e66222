"""ORACLE (test infrastructure, NOT product code) -- ctypes loader for oracle/liboracle.so
(the C restatement in oracle_rhs.c) plus problem builders that run the reference's driver
set-up (oracle/ref_setup.py) for the BASELINE configurations.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this.
"""
import ctypes as C
import os
import subprocess

import numpy as np

from . import ref_physics as ph
from . import ref_setup as rs

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None
_LIBQ = None

_dp = C.POINTER(C.c_double)
_ip = C.POINTER(C.c_int32)
_lp = C.POINTER(C.c_int64)


def build(force=False, quad=False):
    """Compile oracle/liboracle.so (Float64) or oracle/liboracle_quad.so (binary128 truth evaluator) with gcc
    (idempotent)."""
    name = "liboracle_quad.so" if quad else "liboracle.so"
    so = os.path.join(_HERE, name)
    src = os.path.join(_HERE, "oracle_rhs.c")
    if force or not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-B", name], stdout=subprocess.DEVNULL)
    return so


class _CnsT(C.Structure):
    _fields_ = ([("K", C.c_int), ("Np", C.c_int), ("Nq", C.c_int), ("Nfq", C.c_int)]
                + [(n, _dp) for n in ("Vq", "Pq", "Vf", "LIFT", "Dr", "Ds", "VhP", "Ph", "Qrh", "Qsh",
                                      "rxJ", "sxJ", "ryJ", "syJ", "J", "wJq", "nxJ", "nyJ", "sJ")]
                + [("mapP", _lp), ("Nb", C.c_int), ("mapB", _lp), ("bkind", _ip), ("BCTYPE", C.c_int),
                   ("Re", C.c_double), ("lambda_", C.c_double), ("mu", C.c_double), ("Pr", C.c_double),
                   ("inviscid_dissp", C.c_int), ("viscous_dissp", C.c_int), ("inflow", C.c_double * 4), ("vlid", _dp)])


class _HexT(C.Structure):
    _fields_ = ([("K", C.c_int), ("Nq", C.c_int), ("Nfq", C.c_int)]
                + [(n, _dp) for n in ("Ef", "Qr", "Qs", "Qt", "Ph", "Lf")]
                + [("rowptr", _ip), ("colidx", _ip), ("vgeo", _dp * 9)]
                + [(n, _dp) for n in ("J", "wJq", "nxJ", "nyJ", "nzJ", "sJ")]
                + [("mapP", _lp), ("lf_scale", C.c_double)])


def _declare(L):
    if True:
        L.oracle_logmean.restype = C.c_double
        L.oracle_logmean.argtypes = [C.c_double] * 4
        L.oracle_euler_fluxes_2d.argtypes = [_dp] * 6
        L.oracle_v_ufun.argtypes = [_dp, _dp]
        L.oracle_u_vfun.argtypes = [_dp, _dp]
        L.oracle_betafun.restype = C.c_double
        L.oracle_betafun.argtypes = [_dp]
        L.oracle_wavespeed.restype = C.c_double
        L.oracle_wavespeed.argtypes = [C.c_double] * 3
        L.oracle_euler_rhs.restype = C.c_double
        L.oracle_euler_rhs.argtypes = ([C.c_int] * 3 + [_dp] * 4 + [_ip, _ip] + [_dp] * 11 + [_lp, C.c_double, C.c_int, _dp])
        L.oracle_cns_rhs_inviscid.argtypes = [C.POINTER(_CnsT), _dp, _dp]
        L.oracle_cns_rhs_viscous.restype = C.c_double
        L.oracle_cns_rhs_viscous.argtypes = [C.POINTER(_CnsT), _dp, _dp]
        L.oracle_cns_rhsRK.argtypes = [C.POINTER(_CnsT), _dp, _dp, C.c_int, _dp]
        L.oracle_euler_fluxes_3d.argtypes = [_dp] * 7
        L.oracle_v_ufun_3d.argtypes = [_dp, _dp]
        L.oracle_u_vfun_3d.argtypes = [_dp, _dp]
        L.oracle_betafun_3d.restype = C.c_double
        L.oracle_betafun_3d.argtypes = [_dp]
        L.oracle_hex_rhs.restype = C.c_double
        L.oracle_hex_rhs.argtypes = [C.POINTER(_HexT), _dp, C.c_int, _dp]
        L.oracle_set_threads.argtypes = [C.c_int]
        L.oracle_get_max_threads.restype = C.c_int
        L.oracle_real_bits.restype = C.c_int
    return L


def lib():
    global _LIB
    if _LIB is None:
        _LIB = _declare(C.CDLL(build()))
        assert _LIB.oracle_real_bits() == 64
    return _LIB


def lib_quad():
    """The truth evaluator: the same C statements compiled with -DORACLE_QUAD (IEEE binary128 arithmetic on the same
    double inputs, result rounded to double once).  Same symbols and signatures as lib()."""
    global _LIBQ
    if _LIBQ is None:
        _LIBQ = _declare(C.CDLL(build(quad=True)))
        assert _LIBQ.oracle_real_bits() == 128
    return _LIBQ


def _d(a):
    return a.ctypes.data_as(_dp)


def ek(x):
    """(n x K) Julia-layout matrix -> C array [K][n]."""
    return np.ascontiguousarray(np.asarray(x, dtype=np.float64).T)


def stack(Q):
    """list of 4 (n x K) matrices -> [4][K][n]."""
    return np.ascontiguousarray(np.stack([np.asarray(q, dtype=np.float64).T for q in Q]))


def unstack(A):
    """[4][K][n] -> list of 4 (n x K) Fortran matrices."""
    return [np.asfortranarray(A[f].T) for f in range(A.shape[0])]


# ---------------------------------------------------------------------------------------
# pointwise wrappers
# ---------------------------------------------------------------------------------------
def logmean(aL, aR, logL=None, logR=None):
    if logL is None:
        logL, logR = np.log(aL), np.log(aR)
    return lib().oracle_logmean(float(aL), float(aR), float(logL), float(logR))


def euler_fluxes_2d(UL, UR):
    UL = np.array(UL, dtype=float)
    UR = np.array(UR, dtype=float)
    lL = np.log(UL[[0, 3]])
    lR = np.log(UR[[0, 3]])
    Fx, Fy = np.zeros(4), np.zeros(4)
    lib().oracle_euler_fluxes_2d(_d(UL), _d(UR), _d(lL), _d(lR), _d(Fx), _d(Fy))
    return Fx, Fy


def v_ufun(U):
    U = np.array(U, dtype=float)
    V = np.zeros(4)
    lib().oracle_v_ufun(_d(U), _d(V))
    return V


def u_vfun(V):
    V = np.array(V, dtype=float)
    U = np.zeros(4)
    lib().oracle_u_vfun(_d(V), _d(U))
    return U


def euler_fluxes_3d(UL, UR):
    UL = np.array(UL, dtype=float)
    UR = np.array(UR, dtype=float)
    lL = np.log(UL[[0, 4]])
    lR = np.log(UR[[0, 4]])
    Fx, Fy, Fz = np.zeros(5), np.zeros(5), np.zeros(5)
    lib().oracle_euler_fluxes_3d(_d(UL), _d(UR), _d(lL), _d(lR), _d(Fx), _d(Fy), _d(Fz))
    return Fx, Fy, Fz


def v_ufun_3d(U):
    U = np.array(U, dtype=float)
    V = np.zeros(5)
    lib().oracle_v_ufun_3d(_d(U), _d(V))
    return V


def u_vfun_3d(V):
    V = np.array(V, dtype=float)
    U = np.zeros(5)
    lib().oracle_u_vfun_3d(_d(V), _d(U))
    return U


# ---------------------------------------------------------------------------------------
# problem builders (the reference drivers' set-up sections)
# ---------------------------------------------------------------------------------------
class Problem:
    pass


def build_euler_problem(N, Kx, Ky):
    """examples/dg2D_euler_quad.jl:21-91: periodic vortex box [0,15]x[-5,5], Gauss collocation."""
    p = Problem()
    VX, VY, EToV = rs.uniform_quad_mesh(Kx, Ky)
    VX = 15 * (1 + VX) / 2
    VY = 5 * VY
    rd = rs.init_reference_quad(N, rs.gauss_quad(0, 0, N))
    md = rs.init_mesh_2D(VX, VY, EToV, rd)
    rs.make_periodic_2D(md, rd, VX, VY)
    ops = rs.euler_quad_ops(rd)
    for n in ("rxJ", "sxJ", "ryJ", "syJ"):                       # :86-88
        setattr(md, n, np.asfortranarray(ops["Vh"] @ getattr(md, n)))
    rho, u, v, pr = ph.vortex(md.xq, md.yq, 0)                     # :81-83
    p.Q = [np.asfortranarray(q) for q in ph.primitive_to_conservative(rho, u, v, pr)]
    p.rd, p.md, p.ops, p.VX, p.VY, p.EToV, p.N = rd, md, ops, VX, VY, EToV, N
    return p


def becker_constants():
    """Constants of the Becker viscous shock tube, examples/CompressibleNS/dg2D_CNS_modalESDG.jl:31-61."""
    g, M_0, mu = 1.4, 3.0, 0.01
    v_inf, m_0, v_0 = 0.2, 1.0, 1.0
    v_1 = (g - 1 + 2 / M_0 ** 2) / (g + 1)
    v_01 = np.sqrt(v_0 * v_1)
    uL, uR = v_0 + v_inf, v_1 + v_inf
    rhoL, rhoR = m_0 / v_0, m_0 / v_1
    eL = 1 / (2 * g) * ((g + 1) / (g - 1) * v_01 ** 2 - v_0 ** 2)
    eR = 1 / (2 * g) * ((g + 1) / (g - 1) * v_01 ** 2 - v_1 ** 2)
    return dict(mu=mu, lam=2 / 3 * mu, Pr=3 / 4, rhoL=rhoL, rhoR=rhoR, uL=uL, uR=uR, vL=0.0, vR=0.0,
                pL=(g - 1) * rhoL * eL, pR=(g - 1) * rhoR * eR)


def shocktube_state(x, y):
    """Smooth surrogate of the Becker profile between the left and right states (tanh ramp around x = 0.25) with a
    small y-periodic perturbation so that every term of the RHS is exercised."""
    st = becker_constants()
    s = .5 * (1 + np.tanh((x - .25) / .2))
    wob = 1 + .02 * np.sin(2 * np.pi * y) * np.exp(-20 * (x - .3) ** 2)
    rho = (st["rhoL"] + (st["rhoR"] - st["rhoL"]) * s) * wob
    u = st["uL"] + (st["uR"] - st["uL"]) * s
    v = .03 * np.cos(2 * np.pi * y) * np.exp(-20 * (x - .3) ** 2)
    p = st["pL"] + (st["pR"] - st["pL"]) * s
    return rho, u, v, p


def grade_vertices(V, grade):
    """Monotone map of [-1,1] onto itself: graded (non-uniform) rectangles, so that every element has its own J,
    metrics and normals."""
    return V + grade * np.sin(np.pi * V) / np.pi


def build_cns_problem(N, Kx, Ky, bc="periodic", BCTYPE=1, Re=1000.0, Pr=.71, shear=0.0, grade=0.0):
    """Modal-ESDG CNS set-up of examples/CompressibleNS/dg2D_CNS_cavity_optimized.jl:21-90 fed
    with the reference *quad* element (SURVEY.md section 8 config mapping).
      bc="periodic": vortex box [0,15]x[-5,5], mapB emptied after the periodic patch;
      bc="cavity"  : [-1,1]^2 lid-driven cavity walls with BCTYPE 1/2/3;
      bc="shocktube": the set-up of examples/CompressibleNS/dg2D_CNS_modalESDG.jl:62-78 on quads: [-0.5,1]x[0,1],
                     periodic patch on all sides, md.mapB kept (inflow at x=-0.5, copy at x=1), BCTYPE 4,
                     mu=0.01, lambda=+2/3 mu, Pr=3/4, no penalty."""
    p = Problem()
    VX, VY, EToV = rs.uniform_quad_mesh(Kx, Ky)
    if grade:
        VX, VY = grade_vertices(VX, grade), grade_vertices(VY, -0.7 * grade)
    if bc == "periodic":
        VX = 15 * (1 + VX) / 2
        VY = 5 * VY
    if bc == "shocktube":
        VX = VX / 4 * 3 + 1 / 4                                    # dg2D_CNS_modalESDG.jl:63-65
        VY = (VY + 1) / 2
    if shear:                                                      # parallelogram (still affine) elements: all four
        VX = VX + shear * VY                                       # metric terms and both normal components non-zero
    rd = rs.init_reference_quad(N)
    md = rs.init_mesh_2D(VX, VY, EToV, rd)
    if bc == "periodic":
        rs.make_periodic_2D(md, rd, VX, VY)
        md.mapB = np.zeros(0, dtype=np.int64)
    if bc == "shocktube":
        mapB = md.mapB.copy()
        rs.make_periodic_2D(md, rd, VX, VY)                        # :72-78, md.mapB still lists all four sides
        xb = rs.vec(md.xf)[mapB - 1]
        keep = (np.abs(xb + .5) < 1e-12) | (np.abs(xb - 1.0) < 1e-12)   # leftwall / rightwall, :165-166
        md.mapB = mapB[keep]
        BCTYPE = 4
    ops = rs.cns_ops(rd)
    for n in ("rxJ", "sxJ", "ryJ", "syJ"):                       # :85-87
        setattr(md, n, np.asfortranarray(ops["Vh"] @ getattr(md, n)))
    p.mu = 1 / Re
    p.lam = -2 / 3 * p.mu                                          # :33-36
    p.inflow = (0.0, 0.0, 0.0, 0.0)
    if bc == "shocktube":
        st = becker_constants()
        p.mu, p.lam, Pr = st["mu"], st["lam"], st["Pr"]
        p.inflow = (st["rhoL"], st["uL"], st["vL"], st["pL"])
    p.Re, p.Pr, p.BCTYPE, p.bc = Re, Pr, BCTYPE, bc
    if bc == "periodic":
        rho, u, v, pr = ph.vortex(md.x, md.y, 0)
    elif bc == "shocktube":
        rho, u, v, pr = shocktube_state(md.x, md.y)
    else:                                                          # smooth non-trivial cavity state
        x, y = md.x, md.y
        # phases: no exact zero of a velocity component on an element interface -- the LF wavespeed's
        # sqrt(|u_n|) (quirk Q1) turns the 1e-17 round-off of such a zero into 3e-9
        rho = 1.0 + .2 * np.exp(-10 * (x ** 2 + y ** 2))
        u = .1 * np.sin(np.pi * x + .3) * np.cos(np.pi * y + .2)
        v = -.1 * np.cos(np.pi * x + .3) * np.sin(np.pi * y + .2)
        pr = (1 / (.3 ** 2 * ph.GAMMA)) * rho ** ph.GAMMA
    p.Q = [np.asfortranarray(q) for q in ph.primitive_to_conservative(rho, u, v, pr)]
    p.rd, p.md, p.ops, p.VX, p.VY, p.EToV, p.N = rd, md, ops, VX, VY, EToV, N
    return p


def hex_smooth_state(x, y, z):
    """Deterministic smooth periodic state on [-1,1]^3 (the script's own initial condition is random,
    dg3D_euler_hex.jl:101-108: `2 .+ .1*rand`, v=1, `p = 1 + .1*rand`)."""
    rho = 2 + .5 * np.sin(np.pi * x) * np.cos(np.pi * y)
    u = .3 * np.sin(np.pi * z + .2)      # phases keep u_n away from exact zeros at nodes: the LF wavespeed's
    v = 1 + .1 * np.cos(np.pi * x)       # sqrt(|u_n|) (quirk Q1) turns 1e-17 noise there into 3e-9
    w = .1 * np.sin(np.pi * (x + y) + .3)
    p = 1 + .2 * np.cos(np.pi * z) * np.sin(np.pi * y)
    return rho, u, v, w, p


def build_hex_problem(N, Kx, Ky=None, Kz=None, A3=None, grade=0.0, a=0.0):
    """examples/dg3D_euler_hex.jl:21-98: periodic box [-1,1]^3, Gauss collocation, a = 0 (affine)."""
    Ky = Kx if Ky is None else Ky
    Kz = Kx if Kz is None else Kz
    p = Problem()
    VX, VY, VZ, EToV = rs.uniform_hex_mesh(Kx, Ky, Kz)
    if grade:
        VX, VY, VZ = grade_vertices(VX, grade), grade_vertices(VY, -0.7 * grade), grade_vertices(VZ, 0.5 * grade)
    rd = rs.init_reference_hex(N, rs.gauss_quad(0, 0, N))
    md = rs.init_mesh_3D(VX, VY, VZ, EToV, rd)
    rs.make_periodic_3D(md, rd)
    ops = rs.hex_driver_setup(md, rd, a=a, A3=A3)    # a != 0: the script's curved mapping (:67-73)
    p.Q = [np.asfortranarray(q) for q in ph.primitive_to_conservative_3D(*hex_smooth_state(md.xq, md.yq, md.zq))]
    p.rd, p.md, p.ops, p.VX, p.VY, p.VZ, p.EToV, p.N = rd, md, ops, VX, VY, VZ, EToV, N
    return p


# ---------------------------------------------------------------------------------------
# C RHS wrappers
# ---------------------------------------------------------------------------------------
class EulerOracle:
    """C restatement of `rhs` (examples/dg2D_euler_quad.jl:141-194) bound to one problem."""

    def __init__(self, p, quad=False):
        md, ops = p.md, p.ops
        self.L = lib_quad() if quad else lib()     # quad: the binary128 truth evaluator (same statements)
        self.K, self.Nq, self.Nfq = md.K, ops["Ph"].shape[0], ops["Lf"].shape[1]
        rowptr = [0]
        col = []
        for ids in ops["Qrsids"]:
            col += [c - 1 for c in ids]
            rowptr.append(len(col))
        self.rowptr = np.array(rowptr, dtype=np.int32)
        self.col = np.array(col, dtype=np.int32)
        c = np.ascontiguousarray
        self.a = dict(Ef=c(ops["Ef"]), Qr=c(ops["Qrh_sparse"]), Qs=c(ops["Qsh_sparse"]), Ph=c(ops["Ph"]), Lf=c(ops["Lf"]),
                      rxJ=ek(md.rxJ), sxJ=ek(md.sxJ), ryJ=ek(md.ryJ), syJ=ek(md.syJ), J=ek(md.J), wJq=ek(md.wJq),
                      nxJ=ek(md.nxJ), nyJ=ek(md.nyJ), sJ=ek(md.sJ))
        self.mapP = np.ascontiguousarray(md.mapP.T.astype(np.int64))

    def rhs_stacked(self, Qs, lf_scale=.5, compute_rhstest=False):
        out = np.zeros_like(Qs)
        a = self.a
        rt = self.L.oracle_euler_rhs(self.K, self.Nq, self.Nfq, _d(Qs), _d(a["Ef"]), _d(a["Qr"]), _d(a["Qs"]),
                                    self.rowptr.ctypes.data_as(_ip), self.col.ctypes.data_as(_ip), _d(a["Ph"]),
                                    _d(a["Lf"]), _d(a["rxJ"]), _d(a["sxJ"]), _d(a["ryJ"]), _d(a["syJ"]), _d(a["J"]),
                                    _d(a["wJq"]), _d(a["nxJ"]), _d(a["nyJ"]), _d(a["sJ"]),
                                    self.mapP.ctypes.data_as(_lp), float(lf_scale), int(compute_rhstest), _d(out))
        return out, rt

    def rhs(self, Q, lf_scale=.5, compute_rhstest=False):
        out, rt = self.rhs_stacked(stack(Q), lf_scale, compute_rhstest)
        return unstack(out), rt


class CnsOracle:
    """C restatement of rhs_inviscid!/rhs_viscous!/rhsRK! (dg2D_CNS_cavity_optimized.jl) bound to one problem."""

    def __init__(self, p, inviscid_dissp=True, viscous_dissp=True, quad=False):
        md, rd, ops = p.md, p.rd, p.ops
        self.L = lib_quad() if quad else lib()     # quad: the binary128 truth evaluator (same statements)
        c = np.ascontiguousarray
        self.K = md.K
        self.Np = rd.Pq.shape[0]
        self.keep = dict(Vq=c(rd.Vq), Pq=c(rd.Pq), Vf=c(rd.Vf), LIFT=c(rd.LIFT), Dr=c(rd.Dr), Ds=c(rd.Ds), VhP=c(ops["VhP"]),
                         Ph=c(ops["Ph"]), Qrh=c(ops["Qrhskew"]), Qsh=c(ops["Qshskew"]), rxJ=ek(md.rxJ), sxJ=ek(md.sxJ),
                         ryJ=ek(md.ryJ), syJ=ek(md.syJ), J=ek(md.J), wJq=ek(md.wJq), nxJ=ek(md.nxJ), nyJ=ek(md.nyJ),
                         sJ=ek(md.sJ))
        self.mapP = np.ascontiguousarray(md.mapP.T.astype(np.int64))
        mapB = np.asarray(md.mapB, dtype=np.int64)
        yb = md.yf.flatten(order="F")[mapB - 1] if mapB.size else np.zeros(0)
        self.mapB = np.ascontiguousarray(mapB)
        self.bkind = np.ascontiguousarray((np.abs(yb - 1) < 1e-12).astype(np.int32))   # lid test, :139
        if int(p.BCTYPE) == 4:                                                           # inflow = the x = xmin side
            xb = md.xf.flatten(order="F")[mapB - 1]
            self.bkind = np.ascontiguousarray((np.abs(xb - md.xf.min()) < 1e-12).astype(np.int32))
        t = _CnsT()
        t.K, t.Np, t.Nq, t.Nfq = md.K, self.Np, rd.Vq.shape[0], rd.Vf.shape[0]
        for k, v in self.keep.items():
            setattr(t, k, _d(v))
        t.mapP = self.mapP.ctypes.data_as(_lp)
        t.Nb = int(mapB.size)
        t.mapB = self.mapB.ctypes.data_as(_lp)
        t.bkind = self.bkind.ctypes.data_as(_ip)
        t.BCTYPE = int(p.BCTYPE)
        t.Re, t.lambda_, t.mu, t.Pr = float(p.Re), float(p.lam), float(p.mu), float(p.Pr)
        t.inviscid_dissp, t.viscous_dissp = int(inviscid_dissp), int(viscous_dissp)
        for i in range(4):
            t.inflow[i] = float(getattr(p, "inflow", (0, 0, 0, 0))[i])
        if getattr(p, "vlid", None) is not None:      # lid velocity as a function of x (convergence_test.jl:72-76)
            xb = md.xf.flatten(order="F")[mapB - 1]
            self.vlid = np.ascontiguousarray(p.vlid(xb), dtype=np.float64)
            t.vlid = _d(self.vlid)
        self.t = t

    def rhs_inviscid(self, Q):
        Qs = stack(Q)
        out = np.zeros_like(Qs)
        self.L.oracle_cns_rhs_inviscid(C.byref(self.t), _d(Qs), _d(out))
        return unstack(out)

    def rhs_viscous(self, Q):
        Qs = stack(Q)
        out = np.zeros_like(Qs)
        rt = self.L.oracle_cns_rhs_viscous(C.byref(self.t), _d(Qs), _d(out))
        return unstack(out), rt

    def rhsRK_stacked(self, Qs, compute_diag=False):
        out = np.zeros_like(Qs)
        diag = np.zeros(2)
        self.L.oracle_cns_rhsRK(C.byref(self.t), _d(Qs), _d(out), int(compute_diag), _d(diag))
        return out, diag

    def rhsRK(self, Q, compute_diag=True):
        out, diag = self.rhsRK_stacked(stack(Q), compute_diag)
        return unstack(out), diag[0], diag[1]


class HexOracle:
    """C restatement of `rhs` (examples/dg3D_euler_hex.jl:167-222) bound to one problem."""

    def __init__(self, p, lf_scale=0.0, quad=False):
        md, ops = p.md, p.ops
        self.L = lib_quad() if quad else lib()     # quad: the binary128 truth evaluator (same statements)
        c = np.ascontiguousarray
        rowptr, col = [0], []
        for ids in ops["Qnzids"]:
            col += [i - 1 for i in ids]
            rowptr.append(len(col))
        self.keep = dict(Ef=c(ops["Ef"]), Qr=c(ops["Qrh_sparse"]), Qs=c(ops["Qsh_sparse"]), Qt=c(ops["Qth_sparse"]),
                         Ph=c(ops["Ph"]), Lf=c(ops["Lf"]), J=ek(md.J), wJq=ek(md.wJq), nxJ=ek(md.nxJ), nyJ=ek(md.nyJ),
                         nzJ=ek(md.nzJ), sJ=ek(md.sJ))
        self.geo = [ek(getattr(md, n)) for n in ("rxJ", "sxJ", "txJ", "ryJ", "syJ", "tyJ", "rzJ", "szJ", "tzJ")]
        self.rowptr = np.array(rowptr, dtype=np.int32)
        self.col = np.array(col, dtype=np.int32)
        self.mapP = np.ascontiguousarray(md.mapP.T.astype(np.int64))
        t = _HexT()
        t.K, t.Nq, t.Nfq = md.K, ops["Ph"].shape[0], ops["Lf"].shape[1]
        for k, v in self.keep.items():
            setattr(t, k, _d(v))
        t.rowptr = self.rowptr.ctypes.data_as(_ip)
        t.colidx = self.col.ctypes.data_as(_ip)
        for m in range(9):
            t.vgeo[m] = _d(self.geo[m])
        t.mapP = self.mapP.ctypes.data_as(_lp)
        t.lf_scale = float(lf_scale)
        self.t = t

    def rhs_stacked(self, Qs, compute_rhstest=False):
        out = np.zeros_like(Qs)
        rt = self.L.oracle_hex_rhs(C.byref(self.t), _d(Qs), int(compute_rhstest), _d(out))
        return out, rt

    def rhs(self, Q, compute_rhstest=False):
        out, rt = self.rhs_stacked(stack(Q), compute_rhstest)
        return unstack(out), rt
