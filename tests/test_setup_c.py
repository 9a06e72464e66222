"""The host-only C set-up of libesdg_hip.so (esdg_setup_*, csrc/esdg_setup.cpp; SURVEY.md section 8b: set-up entry
points of the C ABI) against the Python host mirror esdg_cns_amd.setup_dg (itself pinned against the oracle restatement
in tests/test_setup.py): integer maps bit for bit, operators and geometry to round-off.  No GPU needed."""
import ctypes as C

import numpy as np
import pytest

from esdg_cns_amd import _lib
from esdg_cns_amd import setup_dg as sd


def _setup(N, form, VX, VY, EToV, periodic, er=None):
    L = _lib.lib()
    vx, vy = np.ascontiguousarray(VX, dtype=np.float64), np.ascontiguousarray(VY, dtype=np.float64)
    et = np.asfortranarray(EToV.astype(np.int64))
    h = C.c_void_p()
    e0, e1 = er if er else (0, 0)
    rc = L.esdg_setup_quad(N, form, vx.ctypes.data_as(_lib.c_double_p), vy.ctypes.data_as(_lib.c_double_p), vx.size,
                           et.ctypes.data_as(_lib.c_int64_p), EToV.shape[0], int(periodic), e0, e1, C.byref(h))
    assert rc == 0, L.esdg_setup_last_error().decode()
    return h


def _arr(h, name):
    L = _lib.lib()
    r, c = C.c_int64(), C.c_int64()
    p = L.esdg_setup_array(h, name.encode(), C.byref(r), C.byref(c))
    assert p, name
    return np.ctypeslib.as_array(p, shape=(c.value, r.value)).T.copy()


def _map(h, name):
    L = _lib.lib()
    n = C.c_int64()
    p = L.esdg_setup_map(h, name.encode(), C.byref(n))
    assert p or n.value == 0, name
    return np.ctypeslib.as_array(p, shape=(n.value,)).copy() if n.value else np.zeros(0, dtype=np.int64)


def test_uniform_quad_mesh_matches():
    L = _lib.lib()
    for Kx, Ky in ((3, 2), (4, 7), (16, 16)):
        VX, VY, E = sd.uniform_quad_mesh(Kx, Ky)
        vx, vy = np.zeros(VX.size), np.zeros(VY.size)
        et = np.zeros((Kx * Ky, 4), dtype=np.int64, order="F")
        assert L.esdg_setup_uniform_quad_mesh(Kx, Ky, vx.ctypes.data_as(_lib.c_double_p), vy.ctypes.data_as(_lib.c_double_p),
                                              et.ctypes.data_as(_lib.c_int64_p)) == 0
        assert np.array_equal(et, E) and np.allclose(vx, VX, atol=1e-15) and np.allclose(vy, VY, atol=1e-15)


@pytest.mark.parametrize("N,Kx,Ky,form,periodic", [(1, 2, 2, 0, True), (2, 3, 2, 0, True), (3, 4, 3, 1, True), (4, 3, 3, 1, False),
                                                    (2, 5, 4, 2, True), (5, 2, 3, 0, True), (8, 2, 2, 1, False), (9, 2, 2, 1, True), (9, 2, 2, 0, True)])
def test_setup_quad_matches_python_mirror(N, Kx, Ky, form, periodic):
    VX, VY, EToV = sd.uniform_quad_mesh(Kx, Ky)
    if periodic:
        VX, VY = 15 * (1 + VX) / 2, 5 * VY
    rd = sd.init_reference_quad(N)
    md = sd.init_mesh((VX, VY), EToV, rd)
    if periodic:
        sd.make_periodic(md, rd)
    ops = sd.euler_quad_ops(rd) if form == 0 else sd.cns_ops(rd)
    sd.interp_geofacs_to_hybrid(md, ops["Vh"])
    h = _setup(N, form, VX, VY, EToV, periodic)
    try:
        for n in ("r", "s", "V1", "Dr", "Ds", "rf", "sf", "wf", "nrJ", "nsJ", "rq", "sq", "wq", "Vq", "M", "Pq", "Vf", "LIFT"):
            ref = np.asarray(getattr(rd, n), dtype=float)
            got = _arr(h, n)
            assert np.abs(got.reshape(ref.shape) - ref).max() < 2e-14 * max(1.0, np.abs(ref).max()) * (N + 1) ** 2, n
        for n in ("Qrhskew", "Qshskew", "Ef", "Vh", "Ph") + (("Lf",) if form == 0 else ("VhP",)):
            assert np.abs(_arr(h, n) - ops[n]).max() < 2e-14 * max(1.0, np.abs(ops[n]).max()) * (N + 1) ** 2, n
        for n in ("x", "y", "xf", "yf", "xq", "yq", "rxJ", "sxJ", "ryJ", "syJ", "J", "wJq", "nxJ", "nyJ", "sJ"):
            assert np.abs(_arr(h, n) - getattr(md, n)).max() < 1e-12, n
        Nfq, K = md.mapP.shape
        assert np.array_equal(_map(h, "mapP").reshape((Nfq, K), order="F"), md.mapP)
        assert np.array_equal(_map(h, "mapM").reshape((Nfq, K), order="F"), md.mapM)
        assert np.array_equal(_map(h, "FToF").reshape((4, K), order="F"), md.FToF)
        assert np.array_equal(_map(h, "mapB"), md.mapB)
        # esdg_setup_fill: struct pointers into the object, walls only when boundary nodes are still self-mapped
        o, m = _lib.esdg_ops_t(), _lib.esdg_mesh_t()
        assert _lib.lib().esdg_setup_fill(h, C.byref(o), C.byref(m)) == 0
        assert (o.N, o.Np, o.Nq, o.Nfq, m.K, m.geo_ld) == (N, (N + 1) ** 2, (N + 1) ** 2, 4 * (N + 1), K, (N + 1) ** 2 + 4 * (N + 1))
        assert m.NmapB == (0 if periodic else md.mapB.size)
        if not periodic:
            bk = np.ctypeslib.as_array(m.bkind, shape=(m.NmapB,))
            yb = md.yf.flatten(order="F")[md.mapB - 1]
            assert np.array_equal(bk.astype(bool), np.abs(yb - 1) < 1e-12)
    finally:
        _lib.lib().esdg_setup_destroy(h)


def test_setup_quad_element_range():
    N, Kx, Ky = 2, 4, 6
    VX, VY, EToV = sd.uniform_quad_mesh(Kx, Ky)
    rd = sd.init_reference_quad(N)
    full = sd.init_mesh((VX, VY), EToV, rd)
    sd.make_periodic(full, rd)
    e0, e1 = 8, 16
    h = _setup(N, 1, VX, VY, EToV, True, (e0, e1))
    try:
        Nfq = full.mapP.shape[0]
        assert np.array_equal(_map(h, "mapP").reshape((Nfq, e1 - e0), order="F"), full.mapP[:, e0:e1])
        assert np.abs(_arr(h, "J") - full.J[:, e0:e1]).max() < 1e-13
    finally:
        _lib.lib().esdg_setup_destroy(h)


# ---- hexahedra ---------------------------------------------------------------------------------------------------
def _setup_hex(N, VX, VY, VZ, EToV, periodic, er=None):
    L = _lib.lib()
    v = [np.ascontiguousarray(a, dtype=np.float64) for a in (VX, VY, VZ)]
    et = np.asfortranarray(EToV.astype(np.int64))
    h = C.c_void_p()
    e0, e1 = er if er else (0, 0)
    rc = L.esdg_setup_hex(N, *[a.ctypes.data_as(_lib.c_double_p) for a in v], v[0].size, et.ctypes.data_as(_lib.c_int64_p),
                          EToV.shape[0], int(periodic), e0, e1, C.byref(h))
    assert rc == 0, L.esdg_setup_last_error().decode()
    return h


def test_uniform_hex_mesh_matches():
    L = _lib.lib()
    for K3 in ((2, 2, 2), (3, 2, 4)):
        VX, VY, VZ, E = sd.uniform_hex_mesh(*K3)
        v = [np.zeros(VX.size) for _ in range(3)]
        et = np.zeros((E.shape[0], 8), dtype=np.int64, order="F")
        assert L.esdg_setup_uniform_hex_mesh(*K3, *[a.ctypes.data_as(_lib.c_double_p) for a in v], et.ctypes.data_as(_lib.c_int64_p)) == 0
        assert np.array_equal(et, E) and all(np.allclose(a, b, atol=1e-15) for a, b in zip(v, (VX, VY, VZ)))


@pytest.mark.parametrize("N,K3,periodic", [(1, (2, 2, 2), True), (2, (3, 2, 2), True), (3, (2, 3, 2), True), (2, (2, 2, 3), False)])
def test_setup_hex_matches_python_mirror(N, K3, periodic):
    VX, VY, VZ, EToV = sd.uniform_hex_mesh(*K3)
    rd = sd.init_reference_hex(N)
    md = sd.init_mesh_3d((VX, VY, VZ), EToV, rd)
    if periodic:
        sd.make_periodic_3d(md, rd)
    ops = sd.hex_ops(rd)
    sd.hex_driver_geometry(md, rd)
    h = _setup_hex(N, VX, VY, VZ, EToV, periodic)
    try:
        sc = (N + 1) ** 3
        for n in ("r", "s", "t", "V1", "Dr", "Ds", "Dt", "rf", "sf", "tf", "wf", "nrJ", "nsJ", "ntJ", "rq", "sq", "tq", "wq", "Vq", "M", "Pq", "Vf", "LIFT"):
            ref = np.asarray(getattr(rd, n), dtype=float)
            assert np.abs(_arr(h, n).reshape(ref.shape) - ref).max() < 2e-14 * max(1.0, np.abs(ref).max()) * sc, n
        for n in ("Qrhskew", "Qshskew", "Qthskew", "Ef", "Vh", "Ph", "Lf"):
            assert np.abs(_arr(h, n) - ops[n]).max() < 2e-14 * max(1.0, np.abs(ops[n]).max()) * sc, n
        for n in ("x", "y", "z", "xf", "yf", "zf", "xq", "yq", "zq", "rxJ", "sxJ", "txJ", "ryJ", "syJ", "tyJ", "rzJ", "szJ", "tzJ", "J", "wJq",
                  "nxJ", "nyJ", "nzJ", "sJ"):
            assert np.abs(_arr(h, n) - getattr(md, n)).max() < 1e-12, n
        Nfq, K = md.mapP.shape
        assert np.array_equal(_map(h, "mapP").reshape((Nfq, K), order="F"), md.mapP)
        assert np.array_equal(_map(h, "FToF").reshape((6, K), order="F"), md.FToF)
        assert np.array_equal(_map(h, "mapB"), md.mapB)
        o, m = _lib.esdg_hex_ops_t(), _lib.esdg_hex_mesh_t()
        assert _lib.lib().esdg_setup_fill_hex(h, C.byref(o), C.byref(m)) == 0
        assert (o.N, o.Nq, o.Nfq, m.K, m.geo_ld) == (N, (N + 1) ** 3, 6 * (N + 1) ** 2, K, (N + 1) ** 3 + 6 * (N + 1) ** 2)
    finally:
        _lib.lib().esdg_setup_destroy(h)


def test_setup_hex_element_range():
    N, K3 = 2, (2, 2, 4)
    VX, VY, VZ, EToV = sd.uniform_hex_mesh(*K3)
    rd = sd.init_reference_hex(N)
    full = sd.init_mesh_3d((VX, VY, VZ), EToV, rd)
    sd.make_periodic_3d(full, rd)
    e0, e1 = 4, 12
    h = _setup_hex(N, VX, VY, VZ, EToV, True, (e0, e1))
    try:
        Nfq = full.mapP.shape[0]
        assert np.array_equal(_map(h, "mapP").reshape((Nfq, e1 - e0), order="F"), full.mapP[:, e0:e1])
    finally:
        _lib.lib().esdg_setup_destroy(h)
