"""Set-up parity (CPU): the oracle's statement-by-statement numpy restatement of the reference set-up
(oracle/ref_setup.py) is pinned by mathematical definitions and by the committed, hand-checked map
fixtures; the PRODUCT's vectorised set-up (esdg_cns_amd/setup_dg.py) must reproduce the oracle's
operators to round-off and its integer maps bit-exactly (mapM/mapP/mapB/FToF)."""
import os

import numpy as np
import pytest

from esdg_cns_amd import setup_dg as sd
from oracle import ref_setup as rs

GOLD = os.path.join(os.path.dirname(__file__), "golden")


@pytest.mark.parametrize("N", [1, 2, 3, 4, 5, 6, 7])
def test_gauss_and_lobatto_rules(N):
    for mod in (rs, sd):
        x, w = mod.gauss_quad(0, 0, N)
        assert abs(w.sum() - 2) < 1e-14
        for k in range(0, 2 * N + 2):            # exact to degree 2N+1
            exact = 0.0 if k % 2 else 2.0 / (k + 1)
            assert abs(np.sum(w * x ** k) - exact) < 1e-13
        xl, wl = mod.gauss_lobatto_quad(0, 0, N)
        assert xl[0] == -1 and xl[-1] == 1 and abs(wl.sum() - 2) < 1e-13
        for k in range(0, 2 * N):                # exact to degree 2N-1
            exact = 0.0 if k % 2 else 2.0 / (k + 1)
            assert abs(np.sum(wl * xl ** k) - exact) < 1e-13
    assert np.allclose(rs.gauss_quad(0, 0, N)[0], sd.gauss_quad(0, 0, N)[0], atol=1e-15)
    assert np.allclose(rs.gauss_lobatto_quad(0, 0, N)[1], sd.gauss_lobatto_quad(0, 0, N)[1], atol=1e-14)


@pytest.mark.parametrize("N", [1, 2, 3, 4, 5])
def test_reference_element_product_vs_oracle(N):
    a, b = rs.init_reference_quad(N), sd.init_reference_quad(N)
    for k in ("r", "s", "rq", "sq", "wq", "rf", "sf", "wf", "nrJ", "nsJ", "V1", "Dr", "Ds", "Vq", "Pq", "M", "Vf", "LIFT"):
        assert np.abs(getattr(a, k) - getattr(b, k)).max() < 1e-12, k
    oa, ob = rs.euler_quad_ops(rs.init_reference_quad(N, rs.gauss_quad(0, 0, N))), sd.euler_quad_ops(sd.init_reference_quad(N, sd.gauss_quad(0, 0, N)))
    for k in ("Qrhskew", "Qshskew", "Ph", "Lf", "Ef"):
        assert np.abs(oa[k] - ob[k]).max() < 1e-12, k
    ca, cb = rs.cns_ops(a), sd.cns_ops(b)
    for k in ("Qrhskew", "Qshskew", "VhP", "Ph"):
        assert np.abs(ca[k] - cb[k]).max() < 1e-11, k


@pytest.mark.parametrize("N", [2, 3, 4])
def test_operator_identities(N):
    rd = rs.init_reference_quad(N)
    # differentiation is exact on the polynomial space; Vq*Pq = I on tensor quads (Nq = Np)
    assert np.abs(rd.Dr @ rd.r ** N - N * rd.r ** (N - 1)).max() < 1e-11
    assert np.abs(rd.Ds @ (rd.r * rd.s ** 2) - 2 * rd.r * rd.s).max() < 1e-11
    assert np.abs(rd.Vq @ rd.Pq - np.eye(rd.Vq.shape[0])).max() < 1e-12
    # SBP property of the hybridized operators and the sparsity counts of SURVEY.md section 8
    ops = rs.euler_quad_ops(rs.init_reference_quad(N, rs.gauss_quad(0, 0, N)))
    assert sum(len(r) for r in ops["Qrsids"]) == 2 * (N + 1) ** 2 * (N + 4)
    Nq = (N + 1) ** 2
    assert np.abs(ops["Qrhskew"][Nq:, Nq:]).max() < 1e-14
    assert np.abs(ops["Qrhskew"] + ops["Qrhskew"].T).max() < 1e-14
    assert np.abs(ops["Qrhskew"].sum(axis=1) - np.concatenate([np.zeros(Nq), -.5 * rd.wf * rd.nrJ])).max() < 1e-12


@pytest.mark.parametrize("Kx,Ky,N", [(2, 2, 1), (3, 2, 2)])
def test_maps_match_golden_fixture(Kx, Ky, N):
    g = np.load(os.path.join(GOLD, f"maps_{Kx}x{Ky}_N{N}.npz"))
    for mod, init in ((rs, lambda *a: rs.init_mesh_2D(*a)), (sd, lambda VX, VY, E, rd: sd.init_mesh((VX, VY), E, rd))):
        VX, VY, EToV = mod.uniform_quad_mesh(Kx, Ky)
        assert np.array_equal(EToV, g["EToV"])
        rd = mod.init_reference_quad(N)
        md = init(VX, VY, EToV, rd)
        assert np.array_equal(md.FToF, g["FToF"])
        assert np.array_equal(md.mapM, g["mapM"]) and np.array_equal(md.mapP, g["mapP_walls"])
        assert np.array_equal(md.mapB, g["mapB"])
        if mod is rs:
            rs.make_periodic_2D(md, rd, VX, VY)
        else:
            sd.make_periodic(md, rd)
        assert np.array_equal(md.mapP, g["mapP_periodic"])


def test_hand_checked_2x2_entries():
    """Element 1 of the 2x2, N=1 mesh, checked by hand against the face orderings of src/SetupDG.jl:235-239."""
    g = np.load(os.path.join(GOLD, "maps_2x2_N1.npz"))
    assert list(g["mapP_walls"][:, 0]) == [1, 2, 16, 15, 18, 17, 7, 8]
    assert list(g["mapP_periodic"][:, 0]) == [22, 21, 16, 15, 18, 17, 12, 11]
    assert list(g["FToF"][:, 0]) == [1, 8, 9, 4]


@pytest.mark.parametrize("N,Kx,Ky", [(3, 4, 3), (2, 5, 7), (4, 8, 8), (1, 16, 16)])
def test_mesh_product_vs_oracle_and_invariants(N, Kx, Ky):
    VX, VY, EToV = rs.uniform_quad_mesh(Kx, Ky)
    VX, VY = 15 * (1 + VX) / 2, 5 * VY
    a, b = rs.init_reference_quad(N), sd.init_reference_quad(N)
    ma, mb = rs.init_mesh_2D(VX, VY, EToV, a), sd.init_mesh((VX, VY), EToV, b)
    assert np.array_equal(ma.FToF, mb.FToF) and np.array_equal(ma.mapM, mb.mapM)
    assert np.array_equal(ma.mapP, mb.mapP) and np.array_equal(ma.mapB, mb.mapB)
    for k in ("x", "y", "xf", "yf", "rxJ", "sxJ", "ryJ", "syJ", "J", "xq", "yq", "wJq", "nxJ", "nyJ", "sJ"):
        assert np.abs(getattr(ma, k) - getattr(mb, k)).max() < 1e-12, k
    rs.make_periodic_2D(ma, a, VX, VY)
    sd.make_periodic(mb, b)
    assert np.array_equal(ma.mapP, mb.mapP)
    # invariants of SURVEY.md section 8(c)(1)
    mapM, mapP = rs.vec(ma.mapM), rs.vec(ma.mapP)
    assert np.array_equal(mapM, np.arange(1, mapM.size + 1))
    assert np.array_equal(mapP[mapP - 1], mapM) and not np.any(mapP == mapM)
    xf, yf = rs.vec(ma.xf), rs.vec(ma.yf)
    dx, dy = np.abs(xf[mapP - 1] - xf), np.abs(yf[mapP - 1] - yf)
    assert np.all((dx < 1e-12) | (np.abs(dx - 15) < 1e-12)) and np.all((dy < 1e-12) | (np.abs(dy - 10) < 1e-12))
    assert abs(ma.wJq.sum() - 150.0) < 1e-10
    # sharded construction reproduces slices of the global maps
    K = Kx * Ky
    for (e0, e1) in ((0, K // 2), (K // 2, K), (Kx, 2 * Kx + 1)):
        mc = sd.init_mesh((VX, VY), EToV, b, elem_range=(e0, e1))
        sd.make_periodic(mc, b)
        assert np.array_equal(mc.mapP, ma.mapP[:, e0:e1])
        assert np.allclose(mc.J, ma.J[:, e0:e1]) and np.allclose(mc.nxJ, ma.nxJ[:, e0:e1])


def test_rk_coefficients():
    for x, y in zip(rs.rk45_coeffs(), sd.rk45_coeffs()):
        assert np.array_equal(x, y)
    for x, y in zip(rs.dopri45_coeffs(), sd.dopri45_coeffs()):
        assert np.array_equal(x, y)
    a, E, c = sd.dopri45_coeffs()
    assert np.allclose(a.sum(axis=1), c) and abs(E.sum()) < 1e-15
