"""Shared helpers for the test-suite: problem builders on the PRODUCT side (esdg_cns_amd.setup_dg)
mirroring the oracle's builders, and error norms."""
import numpy as np

from esdg_cns_amd import physics as ph
from esdg_cns_amd import setup_dg as sd


def rel_l2(a, b):
    """max over fields of ||a-b||_2 / ||b||_2"""
    return max(np.linalg.norm(x - y) / max(np.linalg.norm(y), 1e-300) for x, y in zip(a, b))


def product_euler_problem(N, Kx, Ky, elem_range=None):
    VX, VY, EToV = sd.uniform_quad_mesh(Kx, Ky)
    VX = 15 * (1 + VX) / 2
    VY = 5 * VY
    rd = sd.init_reference_quad(N, sd.gauss_quad(0, 0, N))
    md = sd.init_mesh((VX, VY), EToV, rd, elem_range=elem_range)
    sd.make_periodic(md, rd)
    ops = sd.euler_quad_ops(rd)
    sd.interp_geofacs_to_hybrid(md, ops["Vh"])
    rho, u, v, p = ph.vortex(md.xq, md.yq, 0)
    Q = [np.asfortranarray(q) for q in ph.primitive_to_conservative(rho, u, v, p)]
    return rd, md, ops, Q


def grade_vertices(V, grade):
    return V + grade * np.sin(np.pi * V) / np.pi


def product_cns_problem(N, Kx, Ky, elem_range=None, grade=0.0):
    VX, VY, EToV = sd.uniform_quad_mesh(Kx, Ky)
    if grade:
        VX, VY = grade_vertices(VX, grade), grade_vertices(VY, -0.7 * grade)
    VX = 15 * (1 + VX) / 2
    VY = 5 * VY
    rd = sd.init_reference_quad(N)
    md = sd.init_mesh((VX, VY), EToV, rd, elem_range=elem_range)
    sd.make_periodic(md, rd)
    md.mapB = np.zeros(0, dtype=np.int64)
    ops = sd.cns_ops(rd)
    sd.interp_geofacs_to_hybrid(md, ops["Vh"])
    rho, u, v, p = ph.vortex(md.x, md.y, 0)
    Q = [np.asfortranarray(q) for q in ph.primitive_to_conservative(rho, u, v, p)]
    return rd, md, ops, Q


def cavity_state(x, y):
    """Smooth non-trivial low-Mach state on the [-1,1]^2 cavity (the same formula oracle.build_cns_problem uses)."""
    rho = 1.0 + .2 * np.exp(-10 * (x ** 2 + y ** 2))
    u = .1 * np.sin(np.pi * x + .3) * np.cos(np.pi * y + .2)     # phases: no exact zero of u_n on an element interface
    v = -.1 * np.cos(np.pi * x + .3) * np.sin(np.pi * y + .2)    # (LF wavespeed sqrt(|u_n|), quirk Q1)
    p = (1 / (.3 ** 2 * ph.GAMMA)) * rho ** ph.GAMMA
    return [np.asfortranarray(q) for q in ph.primitive_to_conservative(rho, u, v, p)]


def product_cavity_problem(N, Kx, Ky, elem_range=None, shear=0.0):
    """Lid-driven-cavity mesh of examples/CompressibleNS/dg2D_CNS_cavity_optimized.jl on quads: walls on all four
    sides (md.mapB kept), lid = the y = +1 side.  shear != 0 turns the squares into parallelograms (affine)."""
    VX, VY, EToV = sd.uniform_quad_mesh(Kx, Ky)
    VX = VX + shear * VY
    rd = sd.init_reference_quad(N)
    md = sd.init_mesh((VX, VY), EToV, rd, elem_range=elem_range)
    ops = sd.cns_ops(rd)
    sd.interp_geofacs_to_hybrid(md, ops["Vh"])
    return rd, md, ops, cavity_state(md.x, md.y)


def perturb(Q, seed=20250117, amp=0.01):
    """Robustness variant of SURVEY.md section 8(d): multiply rho and E by 1 + amp*xi, xi in [-1,1)."""
    rng = np.random.default_rng(seed)
    out = [q.copy() for q in Q]
    out[0] *= 1 + amp * (2 * rng.random(Q[0].shape) - 1)
    out[3] *= 1 + amp * (2 * rng.random(Q[0].shape) - 1)
    return out


# ---- parity against the binary128 truth evaluator -------------------------------------------------------------------
TOL = 1e-12          # BASELINE.json north_star: "<=1e-12 relative L2 vs the Julia reference"
_PARITY_LOG = []


def as_oracle_problem(rd, md, ops, Q, **phys):
    """Wrap set-up objects (the product's setup_dg ones or the oracle's ref_setup ones: same field names) as the
    `Problem` the oracle classes take, so that oracle, truth evaluator and engine all get IDENTICAL inputs -- two
    set-up implementations differ by ~1e-13 in J and the normals on a 64x64 mesh, which the RHS amplifies to ~1e-10.
    Adds the driver's per-row nonzero-column lists (dg2D_euler_quad.jl:64, dg3D_euler_hex.jl:60-64) where missing."""
    from oracle import oracle as orc
    p = orc.Problem()
    p.rd, p.md, p.ops, p.Q = rd, md, dict(ops), Q
    p.N = rd.N
    for k, v in phys.items():
        setattr(p, k, v)
    o = p.ops
    if "Qrh_sparse" in o and "Qrsids" not in o and "Qth_sparse" not in o:
        o["Qrsids"] = []
        for i in range(o["Qrh_sparse"].shape[0]):
            a = list(np.nonzero(o["Qrh_sparse"][i])[0] + 1)
            o["Qrsids"].append(a + [j for j in list(np.nonzero(o["Qsh_sparse"][i])[0] + 1) if j not in a])
    if "Qthskew" in o and "Qnzids" not in o:          # hex driver, dg3D_euler_hex.jl:57-64 (droptol 1e-12, union of ids)
        for a, b in (("Qrh_sparse", "Qrhskew"), ("Qsh_sparse", "Qshskew"), ("Qth_sparse", "Qthskew")):
            M = np.array(o[b], dtype=float)
            M[np.abs(M) < 1e-12] = 0.0
            o[a] = M
        o["Qnzids"] = []
        for i in range(o["Qrh_sparse"].shape[0]):
            ids = []
            for M in (o["Qrh_sparse"], o["Qsh_sparse"], o["Qth_sparse"]):
                ids += [j for j in list(np.nonzero(M[i])[0] + 1) if j not in ids]
            o["Qnzids"].append(ids)
    return p


def truth_gate(label, got, ref64, truth, tol=TOL, factor=2.0):
    """The parity gate: e_gpu = |gpu - truth|/|truth| must not exceed max(tol, factor * e_orc), where
    e_orc = |oracle_f64 - truth|/|truth| is the rounding error of a faithful Float64 evaluation of the reference's
    statements and `truth` the same statements in IEEE binary128 (oracle/liboracle_quad.so); max over the conserved
    fields of the relative L2 norm.  Both errors are printed and recorded (gpurun_out/parity_errors.json)."""
    e_gpu, e_orc, d = rel_l2(got, truth), rel_l2(ref64, truth), rel_l2(got, ref64)
    print(f"{label}: e_gpu={e_gpu:.2e} e_orc={e_orc:.2e} gpu-vs-oracle={d:.2e} gate=max({tol:g}, {factor:g}*e_orc)")
    _PARITY_LOG.append(dict(case=label, e_gpu=e_gpu, e_orc=e_orc, gpu_vs_oracle=d, tol=tol, factor=factor))
    try:
        import json
        import os
        out = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
        os.makedirs(out, exist_ok=True)
        json.dump(_PARITY_LOG, open(os.path.join(out, "parity_errors.json"), "w"), indent=1)
    except OSError:
        pass
    assert e_gpu <= max(tol, factor * e_orc), (label, e_gpu, e_orc)
    return e_gpu, e_orc


def noise_floor(rhs_fn, Q, trials=3, seed=7):
    """Round-off noise floor of the ORACLE itself: the largest relative-L2 change of its output when
    every input entry is perturbed by at most one ulp.  The reference's logmean switches to
    -da/(logL-logR) for |f| >= 1e-4 (examples/EntropyStableEuler/logmean.jl:23-27), which loses up to
    four digits to cancellation, so two faithful implementations (e.g. Julia with another BLAS) differ
    by this much.  GPU-vs-oracle tolerances are max(1e-12, 4 x this floor)."""
    rng = np.random.default_rng(seed)
    base = rhs_fn(Q)
    worst = 0.0
    for _ in range(trials):
        Q2 = [q * (1 + 1.1e-16 * rng.choice([-1.0, 0.0, 1.0], size=q.shape)) for q in Q]
        worst = max(worst, rel_l2(rhs_fn(Q2), base))
    return worst


def steep_state(x, y, LX=15.0, LY=10.0):
    """Monotone exponential profiles (discontinuous across the periodic wrap, which is legitimate
    input): every node pair an SBP operator couples differs by |f| >~ 1e-3 in rho and beta, far
    from the reference logmean's ill-conditioned window around its 1e-4 threshold, so the oracle's
    own round-off noise is ~1e-14 and the strict 1e-12 bound is meaningful."""
    xs, ys = x / LX, (y + 5.0) / LY
    rho = np.exp(2.0 * xs + 1.4 * ys)
    p = np.exp(-1.1 * xs + 0.9 * ys)
    u = 0.3 + 0.25 * xs - 0.1 * ys
    v = -0.2 + 0.15 * xs + 0.3 * ys
    return [np.asfortranarray(q) for q in ph.primitive_to_conservative(rho, u, v, p)]


# ---- hexahedra (examples/dg3D_euler_hex.jl) ---------------------------------------------------------------
def hex_smooth_state(x, y, z):
    """Same formula as oracle.hex_smooth_state (the reference script's own IC is random, dg3D_euler_hex.jl:101-108)."""
    rho = 2 + .5 * np.sin(np.pi * x) * np.cos(np.pi * y)
    u = .3 * np.sin(np.pi * z + .2)      # phases keep u_n away from exact zeros at nodes: the LF wavespeed's
    v = 1 + .1 * np.cos(np.pi * x)       # sqrt(|u_n|) (quirk Q1) turns 1e-17 noise there into 3e-9
    w = .1 * np.sin(np.pi * (x + y) + .3)
    p = 1 + .2 * np.cos(np.pi * z) * np.sin(np.pi * y)
    return [np.asfortranarray(q) for q in ph.primitive_to_conservative_3d(rho, u, v, w, p)]


def hex_steep_state(x, y, z):
    """3D analogue of steep_state: monotone exponential profiles on [-1,1]^3 (discontinuous across the periodic
    wrap), every coupled node pair far from logmean's ill-conditioned |f| ~ 1e-4 window."""
    rho = np.exp(0.3 * x + 0.21 * y - 0.15 * z)
    p = np.exp(-0.165 * x + 0.135 * y + 0.18 * z)
    u = 0.3 + 0.12 * x - 0.05 * y + 0.08 * z
    v = -0.2 + 0.07 * x + 0.15 * y - 0.1 * z
    w = 0.1 - 0.09 * x + 0.06 * y + 0.11 * z
    return [np.asfortranarray(q) for q in ph.primitive_to_conservative_3d(rho, u, v, w, p)]


def hex_random_state(shape, seed=20250117, vel=(0.0, 1.0, 0.0)):
    """The script's own kind of initial condition (:101-108): rho = 2 + .1 rand, (u,v,w) = (0,1,0), p = 1 + .1 rand.
    With the LF term switched on (the script has it multiplied by 0) pass a velocity without exact zeros: the
    wavespeed's sqrt(|u_n|) (quirk Q1) turns the 1e-17 round-off of a tangential normal component into 3e-9."""
    rng = np.random.default_rng(seed)
    rho = 2 + .1 * rng.random(shape)
    p = 1 + .1 * rng.random(shape)
    z = np.zeros(shape)
    return [np.asfortranarray(q) for q in ph.primitive_to_conservative_3d(rho, z + vel[0], z + vel[1], z + vel[2], p)]


def product_hex_problem(N, Kx, Ky=None, Kz=None, elem_range=None, hybrid=True, A3=None, grade=0.0, a=0.0):
    VX, VY, VZ, EToV = sd.uniform_hex_mesh(Kx, Ky, Kz)
    if grade:
        VX, VY, VZ = grade_vertices(VX, grade), grade_vertices(VY, -0.7 * grade), grade_vertices(VZ, 0.5 * grade)
    rd = sd.init_reference_hex(N, sd.gauss_quad(0, 0, N))
    md = sd.init_mesh_3d((VX, VY, VZ), EToV, rd, elem_range=elem_range)
    sd.make_periodic_3d(md, rd)
    ops = sd.hex_ops(rd)
    sd.hex_driver_geometry(md, rd, hybrid=hybrid, A3=A3, a=a)
    return rd, md, ops, hex_smooth_state(md.xq, md.yq, md.zq)


def perturb_hex(Q, seed=20250117, amp=0.01):
    rng = np.random.default_rng(seed)
    out = [q.copy() for q in Q]
    out[0] *= 1 + amp * (2 * rng.random(Q[0].shape) - 1)
    out[4] *= 1 + amp * (2 * rng.random(Q[0].shape) - 1)
    return out


# ---- shock-tube closures (examples/CompressibleNS/dg2D_CNS_modalESDG.jl) on quads -----------------------------
def becker_constants():
    """dg2D_CNS_modalESDG.jl:31-61 (same numbers as oracle.becker_constants)."""
    g, M_0, mu = 1.4, 3.0, 0.01
    v_inf, m_0, v_0 = 0.2, 1.0, 1.0
    v_1 = (g - 1 + 2 / M_0 ** 2) / (g + 1)
    v_01 = np.sqrt(v_0 * v_1)
    eL = 1 / (2 * g) * ((g + 1) / (g - 1) * v_01 ** 2 - v_0 ** 2)
    eR = 1 / (2 * g) * ((g + 1) / (g - 1) * v_01 ** 2 - v_1 ** 2)
    rhoL, rhoR = m_0 / v_0, m_0 / v_1
    return dict(mu=mu, lam=2 / 3 * mu, Pr=3 / 4, rhoL=rhoL, rhoR=rhoR, uL=v_0 + v_inf, uR=v_1 + v_inf, vL=0.0,
                pL=(g - 1) * rhoL * eL, pR=(g - 1) * rhoR * eR)


def shocktube_state(x, y):
    st = becker_constants()
    s = .5 * (1 + np.tanh((x - .25) / .2))
    wob = 1 + .02 * np.sin(2 * np.pi * y) * np.exp(-20 * (x - .3) ** 2)
    rho = (st["rhoL"] + (st["rhoR"] - st["rhoL"]) * s) * wob
    u = st["uL"] + (st["uR"] - st["uL"]) * s
    v = .03 * np.cos(2 * np.pi * y) * np.exp(-20 * (x - .3) ** 2)
    p = st["pL"] + (st["pR"] - st["pL"]) * s
    return [np.asfortranarray(q) for q in ph.primitive_to_conservative(rho, u, v, p)]


def product_shocktube_problem(N, Kx, Ky, elem_range=None):
    """Mesh and maps of dg2D_CNS_modalESDG.jl:62-78 with the quad element: [-0.5,1]x[0,1], periodic patch on all four
    sides, md.mapB = the nodes of the two x-sides (leftwall / rightwall of init_BC_funs :165-166)."""
    VX, VY, EToV = sd.uniform_quad_mesh(Kx, Ky)
    VX, VY = VX / 4 * 3 + 1 / 4, (VY + 1) / 2
    rd = sd.init_reference_quad(N)
    md = sd.init_mesh((VX, VY), EToV, rd, elem_range=elem_range)
    mapB = md.mapB.copy()
    sd.make_periodic(md, rd)
    loc = mapB - 1 - md.elem_offset * md.mapP.shape[0]
    xb = md.xf.flatten(order="F")[loc]
    md.mapB = mapB[(np.abs(xb + .5) < 1e-12) | (np.abs(xb - 1.0) < 1e-12)]
    ops = sd.cns_ops(rd)
    sd.interp_geofacs_to_hybrid(md, ops["Vh"])
    return rd, md, ops, shocktube_state(md.x, md.y)
