"""GPU parity of the hexahedral path (esdg_create_hex; `rhs` of examples/dg3D_euler_hex.jl:167-222) against the
CPU oracle (oracle_hex_rhs), through the C ABI.  Gate as in test_gpu_parity.py (tests/common.py:truth_gate), identical
inputs on both sides: e_gpu = |gpu - truth|/|truth| <= max(1e-12, 2 x e_orc), e_orc = the Float64 oracle's own distance
from the binary128 evaluation of the same statements; the strict 1e-12 on the well-conditioned state."""
import numpy as np
import pytest

from common import (as_oracle_problem, hex_random_state, hex_steep_state, perturb_hex, product_hex_problem, rel_l2, truth_gate)

pytestmark = pytest.mark.gpu
TOL = 1e-12


@pytest.fixture(scope="module")
def eng_mod():
    from esdg_cns_amd import engine
    return engine


def _gpu_rhs(eng, Q):
    return eng.download(eng.rhs(eng.upload(Q)))


# (N >= 4: more than one wavefront of Gauss nodes per element -> the degree-generic kernels kh_project_g / kh_rhs_g)
@pytest.mark.parametrize("N,K3", [(3, (4, 4, 4)), (3, (5, 3, 2)), (2, (3, 4, 5)), (1, (4, 3, 3)), (4, (3, 2, 2)), (5, (2, 2, 3)), (6, (2, 2, 2)), (7, (2, 2, 2)),
                                  (8, (2, 2, 2)), (9, (2, 2, 2)), (10, (2, 2, 1))])   # (N = 8 ... 10: round 5, kh_project_g + kh_rhs_l; one element per workgroup)
@pytest.mark.parametrize("lf", [0.0, 0.25])
def test_hex_matches_oracle(eng_mod, oracle_lib, N, K3, lf):
    from oracle import oracle as orc
    po = orc.build_hex_problem(N, *K3)
    rd, md, ops, Q = product_hex_problem(N, *K3)
    assert np.array_equal(md.mapP, po.md.mapP)
    p = as_oracle_problem(rd, md, ops, Q)
    ho, hq = orc.HexOracle(p, lf), orc.HexOracle(p, lf, quad=True)
    eng = eng_mod.RhsEngine(rd, md, ops, eng_mod.EULER_HEX_COLLOCATED, lf_scale=lf)
    for name, state in (("smooth", Q), ("steep", hex_steep_state(md.xq, md.yq, md.zq)), ("random", hex_random_state(Q[0].shape, vel=(0, 1, 0) if lf == 0 else (.13, 1, -.07)))):
        e_gpu, _ = truth_gate(f"hex N={N} {K3} lf={lf} {name}", _gpu_rhs(eng, state), ho.rhs(state)[0], hq.rhs(state)[0])
        if name == "steep" and N >= 2:
            assert e_gpu <= TOL, e_gpu


def test_hex_rhstest_entropy_conservation(eng_mod, oracle_lib):
    """The script's own check (`@show rhstest`, :224-226, "for testing EC"): with the LF term off the discrete
    entropy production sum(wJq v.rhs) vanishes to round-off; it matches the oracle's value with LF on."""
    from oracle import oracle as orc
    rd, md, ops, Q = product_hex_problem(3, 4, 4, 4)
    Qr = hex_random_state(Q[0].shape, vel=(.13, 1, -.07))
    eng0 = eng_mod.RhsEngine(rd, md, ops, eng_mod.EULER_HEX_COLLOCATED, lf_scale=0.0)
    Qd = eng0.upload(Qr)
    r = eng0.rhs(Qd)
    rt = eng0.rhstest(Qd, r)
    scale = float(np.sum(np.abs(md.wJq)) * max(np.abs(x).max() for x in eng0.download(r)))
    print("hex rhstest (LF off)", rt, "scale", scale)
    assert abs(rt) < 1e-12 * scale
    p = orc.build_hex_problem(3, 4, 4, 4)
    ho = orc.HexOracle(p, 0.25)
    eng1 = eng_mod.RhsEngine(rd, md, ops, eng_mod.EULER_HEX_COLLOCATED, lf_scale=0.25)
    Qd = eng1.upload(Qr)
    r = eng1.rhs(Qd)
    rt1 = eng1.rhstest(Qd, r)
    _, rto = ho.rhs(Qr, True)
    print("hex rhstest (LF .25)", rt1, rto)
    assert abs(rt1 - rto) < 1e-10 * max(1.0, abs(rto))


@pytest.mark.parametrize("N,K3", [(3, (6, 4, 4)), (4, (3, 2, 4))])   # N = 4: the degree-generic kernels kh_*_g
def test_hex_invariants_shards_and_fused_rk(eng_mod, N, K3):
    """Free-stream preservation, conservation, two shards on one GPU == one engine bit for bit, bitwise
    reproducibility, host drop-in entry point, fused LSRK stage == unfused."""
    import torch
    from esdg_cns_amd import setup_dg as sd
    rd, md, ops, Q = product_hex_problem(N, *K3)
    eng = eng_mod.RhsEngine(rd, md, ops, eng_mod.EULER_HEX_COLLOCATED, lf_scale=0.25)
    # free stream
    c = [np.full_like(Q[0], v) for v in (1.3, 0.4, -0.3, 0.2, 2.9)]
    r = _gpu_rhs(eng, c)
    assert max(np.abs(x).max() for x in r) < 1e-11
    # conservation on a perturbed state
    Qp = perturb_hex(Q)
    Qd = eng.upload(Qp)
    rd_ = eng.rhs(Qd)
    r = eng.download(rd_)
    for f in range(5):
        assert abs(np.sum(md.wJq * r[f])) < 1e-11 * np.sum(np.abs(md.wJq * r[f]))
    # reproducibility
    assert torch.equal(rd_, eng.rhs(Qd))
    # host entry point
    rh = eng.rhs_host(Qp)
    assert all(np.array_equal(a, b) for a, b in zip(rh, r))
    # two z-slab shards on one GPU, traces exchanged by device copies
    Kg = md.K
    cut = (K3[0] * K3[1]) * 2
    offs = [0, cut, Kg]
    engs, mds = [], []
    for rk in range(2):
        _, mdl, _, _ = product_hex_problem(N, *K3, elem_range=(offs[rk], offs[rk + 1]))
        mds.append(mdl)
        engs.append(eng_mod.RhsEngine(rd, mdl, ops, eng_mod.EULER_HEX_COLLOCATED, lf_scale=0.25, rank=rk, nranks=2,
                                      rank_offsets=offs))
    Qs = [engs[rk].upload([q[:, offs[rk]:offs[rk + 1]] for q in Qp]) for rk in range(2)]
    outs = [torch.empty_like(q) for q in Qs]
    import ctypes as C
    for ph in range(2):
        for rk in range(2):
            e = engs[rk]
            e.L.esdg_rhs_phase(e.ctx, ph, C.c_void_p(Qs[rk].data_ptr()), C.c_void_p(outs[rk].data_ptr()), e._stream())
        if ph == 0:
            torch.cuda.synchronize()
            for rk in range(2):
                for peer, so, sb, ro, rb in engs[rk].halo.segments[0]:
                    src = engs[peer]
                    seg = [s for s in src.halo.segments[0] if s[0] == rk][0]
                    assert seg[2] == rb
                    engs[rk].ws[ro:ro + rb].copy_(src.ws[seg[1]:seg[1] + seg[2]])
            torch.cuda.synchronize()
    full = torch.cat(outs, dim=1)
    if N <= 3:
        assert torch.equal(full, rd_)
    else:
        # (at N = 4 the product set-up's own arrays differ in the last bit between the full mesh and an element range --
        # numpy matmul blocks differently for 24 and 12 columns: tzJ 6e-16, nxJ 4e-16 -- so the shards are fed different
        # geometry; the kernels themselves are order-deterministic, see the reproducibility check above)
        assert float((full - rd_).abs().max()) <= 1e-11 * float(rd_.abs().max())
    # fused low-storage RK stage
    a, b, dt = -0.4178904745, 0.3792103130, 1e-3
    Q1, res1 = Qd.clone(), torch.full_like(Qd, 0.01)
    Q2, res2 = Q1.clone(), res1.clone()
    eng.rhs_lsrk_fused(Q1, res1, a, b, dt)
    eng.lsrk_update(Q2, res2, eng.rhs(Q2), a, b, dt)
    assert torch.equal(Q1, Q2) and torch.equal(res1, res2)


def test_hex_full_size_properties_128x128x16(eng_mod):
    """One GPU's share of BASELINE config 5 (N=3, 128x128x16 of the 128^3 box = 262 144 hexahedra): the
    size-independent properties that pin the scheme -- free stream, discrete conservation, entropy conservation with
    the LF term at the script's factor 0 (`@show rhstest`), bitwise reproducibility."""
    import torch
    rd, md, ops, Q = product_hex_problem(3, 128, 128, 16, hybrid=False)
    eng = eng_mod.RhsEngine(rd, md, ops, eng_mod.EULER_HEX_COLLOCATED, lf_scale=0.0)
    c = [np.full_like(Q[0], v) for v in (1.3, 0.4, -0.3, 0.2, 2.9)]
    r = eng.rhs(eng.upload(c))
    assert float(r.abs().max()) < 1e-11 / float(np.abs(md.J).min())       # round-off of O(1) fluxes times 1/|J|
    Qp = perturb_hex(Q)
    Qd = eng.upload(Qp)
    rd_ = eng.rhs(Qd)
    rh = eng.download(rd_)
    for x in rh:
        assert abs(float((md.wJq * x).sum())) <= 2e-9 * max(float(np.abs(md.wJq * x).sum()), 1.0)
    rt = eng.rhstest(Qd, rd_)
    scale = float(np.abs(md.wJq).sum()) * max(float(np.abs(x).max()) for x in rh)
    print(f"hex 128x128x16: rhstest {rt:.3e} (scale {scale:.3e})")
    assert abs(rt) < 1e-11 * scale
    assert torch.equal(rd_, eng.rhs(Qd))


def test_hex_sheared_parallelepiped_mesh_matches_oracle(eng_mod, oracle_lib):
    """Affine but not axis-aligned hexahedra: the nodes are mapped by a shear after the (topological) periodic maps are
    built -- the script re-derives its geometry from x,y,z at that point (dg3D_euler_hex.jl:67-90) -- so all nine
    metric terms and all three components of every normal are non-zero.  A sheared periodic lattice still tiles space:
    the free stream must be preserved too."""
    from oracle import oracle as orc
    A3 = np.array([[1.0, 0.3, 0.2], [0.0, 1.0, 0.15], [0.1, 0.0, 1.0]])
    N, K3 = 3, (4, 3, 3)
    po = orc.build_hex_problem(N, *K3, A3=A3)
    assert min(np.abs(getattr(po.md, n)).min() for n in ("sxJ", "txJ", "ryJ", "tyJ", "rzJ", "szJ")) > 1e-4
    rd, md, ops, Q = product_hex_problem(N, *K3, A3=A3)
    for n in ("rxJ", "tyJ", "szJ", "J", "nxJ", "nzJ", "sJ"):
        assert np.abs(getattr(md, n) - getattr(po.md, n)).max() < (1e-12 if N <= 8 else 1e-11), n   # (two set-ups; conditioning grows with N)
    p = as_oracle_problem(rd, md, ops, Q)
    for lf in (0.0, 0.25):
        ho, hq = orc.HexOracle(p, lf), orc.HexOracle(p, lf, quad=True)
        eng = eng_mod.RhsEngine(rd, md, ops, eng_mod.EULER_HEX_COLLOCATED, lf_scale=lf)
        for name, state in (("smooth", p.Q), ("random", hex_random_state(p.Q[0].shape, vel=(.13, 1, -.07)))):
            truth_gate(f"sheared hex lf={lf} {name}", _gpu_rhs(eng, state), ho.rhs(state)[0], hq.rhs(state)[0])
        c = [np.full_like(p.Q[0], v) for v in (1.3, 0.4, -0.3, 0.2, 2.9)]
        assert max(np.abs(x).max() for x in _gpu_rhs(eng, c)) < 1e-10


def test_hex_graded_mesh_every_element_its_own_geometry(eng_mod, oracle_lib):
    from oracle import oracle as orc
    N, K3, g = 3, (5, 4, 3), 0.45
    po = orc.build_hex_problem(N, *K3, grade=g)
    assert np.abs(po.md.J).max() / np.abs(po.md.J).min() > 2
    rd, md, ops, Q = product_hex_problem(N, *K3, grade=g)
    assert np.array_equal(md.mapP, po.md.mapP)
    p = as_oracle_problem(rd, md, ops, Q)
    for lf in (0.0, 0.25):
        ho, hq = orc.HexOracle(p, lf), orc.HexOracle(p, lf, quad=True)
        eng = eng_mod.RhsEngine(rd, md, ops, eng_mod.EULER_HEX_COLLOCATED, lf_scale=lf)
        truth_gate(f"graded hex mesh lf={lf}", _gpu_rhs(eng, Q), ho.rhs(Q)[0], hq.rhs(Q)[0])


@pytest.mark.parametrize("N,K3", [(3, (4, 4, 4)), (2, (3, 3, 4)), (4, (2, 2, 3)), (5, (2, 2, 2)), (8, (2, 2, 1)), (10, (2, 1, 1))])   # (kh_rhs_l<N1, 1>; N = 10: 40 spilled registers, one wave per SIMD)
def test_hex_curved_mesh_matches_oracle(eng_mod, oracle_lib, N, K3):
    """The script's curved mapping x,y,z += a (x^2-1)(y^2-1)(z^2-1) (dg3D_euler_hex.jl:67-73; a = 0 in the script
    itself): per-node metric terms at the hybrid nodes, per-pair metric averages (:145-151), per-node normals and J.
    Entropy conservation (`@show rhstest`) and the free stream hold on the curved mesh too (curl-form metrics)."""
    from oracle import oracle as orc
    a = 0.12
    po = orc.build_hex_problem(N, *K3, a=a)
    assert np.abs(po.md.rxJ - po.md.rxJ[0]).max() > 1e-3                      # really non-affine
    rd, md, ops, Q = product_hex_problem(N, *K3, a=a)
    for n in ("rxJ", "tyJ", "szJ", "J", "nxJ", "nzJ", "sJ"):
        assert np.abs(getattr(md, n) - getattr(po.md, n)).max() < (1e-12 if N <= 8 else 1e-11), n   # (two set-ups; conditioning grows with N)
    p = as_oracle_problem(rd, md, ops, Q)
    for lf in (0.0, 0.25):
        ho, hq = orc.HexOracle(p, lf), orc.HexOracle(p, lf, quad=True)
        eng = eng_mod.RhsEngine(rd, md, ops, eng_mod.EULER_HEX_COLLOCATED, lf_scale=lf)
        for name, state in (("smooth", p.Q), ("random", hex_random_state(p.Q[0].shape, vel=(.13, 1, -.07)))):
            truth_gate(f"curved hex N={N} lf={lf} {name}", _gpu_rhs(eng, state), ho.rhs(state)[0], hq.rhs(state)[0])
        c = [np.full_like(p.Q[0], v) for v in (1.3, 0.4, -0.3, 0.2, 2.9)]
        assert max(np.abs(x).max() for x in _gpu_rhs(eng, c)) < 1e-10          # free stream on the curved mesh
        if lf == 0.0:
            Qd = eng.upload(hex_random_state(p.Q[0].shape, vel=(.13, 1, -.07)))
            r = eng.rhs(Qd)
            rt = eng.rhstest(Qd, r)
            scale = float(np.sum(np.abs(md.wJq)) * float(r.abs().max()))
            print(f"curved hex rhstest (LF off) {rt:.3e} scale {scale:.3e}")
            assert abs(rt) < 1e-12 * scale


def _slab_periodic_state(x, y, z, LZ):
    """Smooth 3D state with period 2 in x and y and LZ in z (one z-slab of the sharded box is then periodic on its own);
    no exact zeros of a normal velocity at nodes (quirk Q1)."""
    from esdg_cns_amd import physics as ph
    cz = 2 * np.pi * (z - z.min()) / LZ
    rho = 2 + .5 * np.sin(np.pi * x + .1) * np.cos(np.pi * y) * (1 + .2 * np.cos(cz + .3))
    u = .3 * np.sin(cz + .2) + .05
    v = 1 + .1 * np.cos(np.pi * x + .4)
    w = .1 * np.sin(np.pi * (x + y) + .3) + .02 * np.cos(cz - .5)
    p = 1 + .2 * np.cos(cz + .6) * np.sin(np.pi * y + .2)
    return [np.asfortranarray(q) for q in ph.primitive_to_conservative_3d(rho, u, v, w, p)]


@pytest.mark.gpu
@pytest.mark.parametrize("Kx,Kzr,lf", [(8, 2, 0.25), (32, 4, 0.0), (8, 16, 0.25)])   # 16 layers: the nested two-stream schedule
def test_cfg5_rank0_slab_of_the_8_rank_box_over_the_library_rccl_transport(eng_mod, Kx, Kzr, lf):
    """BASELINE config 5 in its 8-rank form (z-slabs of the periodic box): rank 0's slab with its ghost slots, send lists
    and the library's RCCL transport in loopback (what goes to the slab below comes in from above: for a state that is
    periodic over the slab exactly what ranks 1 and 7 would send).  Bit for bit the stand-alone periodic slab, also
    through the fused LSRK stage."""
    import copy
    import torch
    from esdg_cns_amd import setup_dg as sd
    N, nr = 3, 8
    Kzt = Kzr * nr

    def build(Kz_total, e0, e1):
        VX, VY, VZ, EToV = sd.uniform_hex_mesh(Kx, Kx, Kz_total)
        VZ = VZ * (Kz_total / Kx)                       # same element size whatever the number of layers
        rd = sd.init_reference_hex(N, sd.gauss_quad(0, 0, N))
        md = sd.init_mesh_3d((VX, VY, VZ), EToV, rd, elem_range=(e0, e1))
        sd.make_periodic_3d(md, rd)
        ops = sd.hex_ops(rd)
        sd.hex_driver_geometry(md, rd, hybrid=False)
        return rd, md, ops

    Ks = Kx * Kx * Kzr
    rd, md, ops = build(Kzt, 0, Ks)
    offsets = np.array([Ks * r for r in range(nr + 1)], dtype=np.int64)
    sh = eng_mod.RhsEngine(rd, md, ops, eng_mod.EULER_HEX_COLLOCATED, lf_scale=lf, rank=0, nranks=nr, rank_offsets=offsets)
    assert sh.attach_rccl(loopback=True) == 1 and sh.transport == "rccl"
    LZ = 2.0 * Kzr / Kx
    Q = _slab_periodic_state(md.xq, md.yq, md.zq, LZ)
    _, md1s, _ = build(Kzr, 0, Ks)                      # the stand-alone periodic slab: the shard's own arrays, its mapP
    md1 = copy.copy(md)
    md1.mapP, md1.elem_offset, md1.Kglobal = md1s.mapP, 0, md.K
    one = eng_mod.RhsEngine(rd, md1, ops, eng_mod.EULER_HEX_COLLOCATED, lf_scale=lf)
    Qd = sh.upload(Q)
    for _ in range(2):
        got, ref = sh.rhs(Qd), one.rhs(Qd)
        torch.cuda.synchronize()
        assert torch.isfinite(got).all()
        assert torch.equal(got, ref), float((got - ref).abs().max() / ref.abs().max())
    q1, q2 = Qd.clone(), Qd.clone()
    r1, r2 = torch.zeros_like(Qd), torch.zeros_like(Qd)
    for k in range(3):
        sh.rhs_lsrk_fused(q1, r1, -0.4 * k, 0.3, 1e-3)
        one.rhs_lsrk_fused(q2, r2, -0.4 * k, 0.3, 1e-3)
    torch.cuda.synchronize()
    assert torch.equal(q1, q2)
    # one DOPRI45 attempt (round 5: the stage combinations and the error norm in kh_rhs_l's node rounds) through the sharded
    # schedule -- the last phase in up to three launches, the norm's terms at their entries' own indices and added in one order,
    # the sum reduced over the communicator -- against the stand-alone slab: same state bits, same estimate
    import ctypes as C
    from esdg_cns_amd.engine import check
    outs = []
    for eng in (sh, one):
        k = [torch.zeros_like(Qd) for _ in range(7)]
        eng.rhs_into(Qd, k[0])
        Qtmp = torch.empty_like(Qd)
        ptrs = (C.c_void_p * 7)(*[t.data_ptr() for t in k])
        err = C.c_double(0.0)
        check(eng.L.esdg_dopri45_attempt(eng.ctx, C.c_void_p(Qd.data_ptr()), C.c_void_p(Qtmp.data_ptr()), ptrs, 2e-3, 1e-5, C.byref(err),
                                         eng._stream()))
        torch.cuda.synchronize()
        outs.append((Qtmp, [t.clone() for t in k], err.value))
    assert torch.equal(outs[0][0], outs[1][0]) and all(torch.equal(a, b) for a, b in zip(outs[0][1], outs[1][1]))
    assert outs[0][2] > 0 and outs[0][2] == outs[1][2], (outs[0][2], outs[1][2])


@pytest.mark.parametrize("K", [16])
def test_hex_per_node_geometry_of_affine_meshes_at_a_size_where_it_matters(eng_mod, oracle_lib, K):
    """Round 3: the reference uses every node's own metric terms and normals (sparse_hadamard_sum :145-151, rhs :193-198); on an
    affine mesh those arrays are constants plus the set-up's round-off, and the per-node use turns it into a multiple of the
    reference's own rounding error that grows with refinement (element record: 1.08 x e_orc at 8^3, 2.5 x at 16^3, 3.8 x at
    24^3).  Default on affine meshes whose driver passed per-node arrays: the record plus 10-bit differences (geometry mode 2
    of kh_rhs) -- inside the ordinary 2 x e_orc gate; ESDG_HEX_GEOMETRY=element (mode 0) and ESDG_HEX_PER_NODE=1 (mode 1, the
    kernels of curved meshes) are recorded beside it."""
    import os
    from oracle import oracle as orc
    n = len(os.sched_getaffinity(0))
    orc.lib().oracle_set_threads(n); orc.lib_quad().oracle_set_threads(n)
    try:
        # (the set-up as the reference performs it, oracle/ref_setup.py: its metric arrays carry more round-off than those of
        # the product's own set-up, with which the element record is at 1.03 x e_orc at this size)
        p = orc.build_hex_problem(3, K)
        rd, md, ops, Q = p.rd, p.md, p.ops, p.Q
        ref, tru = orc.HexOracle(p, 0.0).rhs(Q)[0], orc.HexOracle(p, 0.0, quad=True).rhs(Q)[0]
        eng = eng_mod.RhsEngine(rd, md, ops, eng_mod.EULER_HEX_COLLOCATED, lf_scale=0.0)
        e2, e_orc = truth_gate(f"hex N=3 {K}^3 smooth (per-node geometry, 10-bit differences)", _gpu_rhs(eng, Q), ref, tru)
        os.environ["ESDG_HEX_PER_NODE"] = "1"
        try:
            e1, _ = truth_gate(f"hex N=3 {K}^3 smooth (per-node geometry, full arrays)", _gpu_rhs(eng_mod.RhsEngine(rd, md, ops, eng_mod.EULER_HEX_COLLOCATED, lf_scale=0.0, ab_hooks=True), Q), ref, tru)
        finally:
            del os.environ["ESDG_HEX_PER_NODE"]
        os.environ["ESDG_HEX_GEOMETRY"] = "element"
        try:
            e0 = rel_l2(_gpu_rhs(eng_mod.RhsEngine(rd, md, ops, eng_mod.EULER_HEX_COLLOCATED, lf_scale=0.0, ab_hooks=True), Q), tru)
        finally:
            del os.environ["ESDG_HEX_GEOMETRY"]
        print(f"hex N=3 {K}^3: e_orc {e_orc:.2e}; e_gpu mode 2 {e2:.2e}, mode 1 {e1:.2e}, mode 0 (element record) {e0:.2e}")
        assert e2 <= 1.1 * max(e1, e_orc)      # the 10-bit differences cost at most a few per cent over the full arrays
        assert e0 <= 20 * e_orc                # the element record filters the set-up's round-off: documented deviation
    finally:
        orc.lib().oracle_set_threads(1); orc.lib_quad().oracle_set_threads(1)


@pytest.mark.parametrize("N", [1, 2, 3, 4])
def test_hex_line_per_lane_and_node_per_lane_kernels_agree(eng_mod, N):
    """kh_rhs_l (production on affine meshes) against kh_rhs / the row-wise kh_rhs_g (ESDG_HEX_LINE=0), element record and per-node
    geometry: two mappings of the same formulas, round-off apart.  (Both settings run in child processes on the A/B build of the
    library, the one that reads the switch.)"""
    import os
    import subprocess
    import sys
    code = r'''
import os, sys, numpy as np
sys.path.insert(0, os.path.join(%r, "tests")); sys.path.insert(0, %r)
import torch
from common import hex_random_state
from esdg_cns_amd import engine, setup_dg as sd
N = %d
for per_node in (False, True):
    VX, VY, VZ, EToV = sd.uniform_hex_mesh(5, 3, 4)
    rd = sd.init_reference_hex(N, sd.gauss_quad(0, 0, N))
    md = sd.init_mesh_3d((VX, VY, VZ), EToV, rd)
    sd.make_periodic_3d(md, rd)
    ops = sd.hex_ops(rd)
    sd.hex_driver_geometry(md, rd, hybrid=per_node)
    Q = hex_random_state(md.xq.shape, seed=5, vel=(0.3, 1.0, -0.2))
    eng = engine.RhsEngine(rd, md, ops, engine.EULER_HEX_COLLOCATED, lf_scale=0.25, ab_hooks=True)
    out = eng.download(eng.rhs(eng.upload(Q)))
    np.save(sys.argv[1] + ("_pn" if per_node else "_el") + ".npy", np.stack(out))
'''
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    import tempfile
    with tempfile.TemporaryDirectory() as td:
        res = {}
        for tag, env in (("line", {}), ("node", {"ESDG_HEX_LINE": "0"})):
            p = subprocess.run([sys.executable, "-c", code % (root, root, N), os.path.join(td, tag)], env=dict(os.environ, **env),
                               capture_output=True, text=True, timeout=600)
            assert p.returncode == 0, p.stderr[-2000:]
            res[tag] = {k: np.load(os.path.join(td, tag + "_" + k + ".npy")) for k in ("el", "pn")}
        for k in ("el", "pn"):
            a, b = res["line"][k], res["node"][k]
            assert np.linalg.norm(a - b) <= 1e-12 * np.linalg.norm(b), (N, k)
