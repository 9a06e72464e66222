"""Element-index sharding on CPU: the host-only halo plan exported by the C ABI
(esdg_halo_plan_*), and a world_size-2 `gloo` run of the product's HaloExchanger moving synthetic
face traces exactly as the GPU path does (pack by send list -> isend/irecv of byte segments -> ghost
slots gathered through the local mapP)."""
import ctypes as C
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from esdg_cns_amd import _lib
from esdg_cns_amd import setup_dg as sd


def _plan(mapP, K, Nfq, e0, Kg, nranks, offsets):
    L = _lib.lib()
    mp_ = np.asfortranarray(mapP.astype(np.int64))
    off = np.ascontiguousarray(np.asarray(offsets, dtype=np.int64))
    h = C.c_void_p()
    _lib.check(L.esdg_halo_plan_create(mp_.ctypes.data_as(_lib.c_int64_p), K, Nfq, e0, Kg, nranks,
                                       off.ctypes.data_as(_lib.c_int64_p), C.byref(h)))
    nn = L.esdg_halo_plan_num_neighbors(h)
    nbrs = []
    for n in range(nn):
        peer = C.c_int32()
        so, sc, ro, rc = C.c_int64(), C.c_int64(), C.c_int64(), C.c_int64()
        _lib.check(L.esdg_halo_plan_neighbor(h, n, C.byref(peer), C.byref(so), C.byref(sc), C.byref(ro), C.byref(rc)))
        nbrs.append((peer.value, so.value, sc.value, ro.value, rc.value))
    ng, ns = L.esdg_halo_plan_num_ghosts(h), L.esdg_halo_plan_num_sends(h)
    lmap = np.ctypeslib.as_array(L.esdg_halo_plan_mapP(h), shape=(K * Nfq,)).copy()
    sl = np.ctypeslib.as_array(L.esdg_halo_plan_sendlist(h), shape=(ns,)).copy() if ns else np.zeros(0, dtype=np.int32)
    L.esdg_halo_plan_destroy(h)
    return dict(nbrs=nbrs, nghost=ng, nsend=ns, mapP=lmap, sendlist=sl)


def _mesh(N, Kx, Ky):
    VX, VY, EToV = sd.uniform_quad_mesh(Kx, Ky)
    rd = sd.init_reference_quad(N)
    return (15 * (1 + VX) / 2, 5 * VY), EToV, rd


@pytest.mark.parametrize("nranks", [1, 2, 3, 4])
def test_plan_single_process(nranks):
    N, Kx, Ky = 2, 4, 6
    V, EToV, rd = _mesh(N, Kx, Ky)
    K, Nfq = Kx * Ky, rd.wf.size
    rows = np.linspace(0, Ky, nranks + 1).astype(int)
    offsets = rows * Kx
    full = sd.init_mesh(V, EToV, rd)
    sd.make_periodic(full, rd)
    gval = np.arange(K * Nfq) * 1.0 + 0.5                    # synthetic trace = f(global node id)
    plans = []
    for r in range(nranks):
        md = sd.init_mesh(V, EToV, rd, elem_range=(offsets[r], offsets[r + 1]))
        sd.make_periodic(md, rd)
        plans.append((md, _plan(md.mapP, md.K, Nfq, int(offsets[r]), K, nranks, offsets)))
    for r, (md, pl) in enumerate(plans):
        Kl = md.K
        buf = np.full(Kl * Nfq + pl["nghost"], np.nan)
        buf[:Kl * Nfq] = gval[offsets[r] * Nfq:offsets[r + 1] * Nfq]
        assert nranks > 1 or (pl["nghost"] == 0 and not pl["nbrs"])
        for (peer, so, sc, ro, rc) in pl["nbrs"]:
            # what the peer packs for us
            pmd, ppl = plans[peer]
            seg = [s for s in ppl["nbrs"] if s[0] == r][0]
            assert seg[2] == rc                                   # their send count == our recv count
            sent = gval[offsets[peer] * Nfq + ppl["sendlist"][seg[1]:seg[1] + seg[2]]]
            buf[Kl * Nfq + ro:Kl * Nfq + ro + rc] = sent
        got = buf[pl["mapP"]]
        expect = gval[md.mapP.flatten(order="F") - 1]
        assert np.array_equal(got, expect)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _gloo_worker(rank, world, port, ret):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from esdg_cns_amd.engine import HaloExchanger
        N, Kx, Ky = 3, 5, 4
        V, EToV, rd = _mesh(N, Kx, Ky)
        K, Nfq = Kx * Ky, rd.wf.size
        offsets = np.array([0, 2 * Kx, K])
        md = sd.init_mesh(V, EToV, rd, elem_range=(offsets[rank], offsets[rank + 1]))
        sd.make_periodic(md, rd)
        pl = _plan(md.mapP, md.K, Nfq, int(offsets[rank]), K, world, offsets)
        ok = True
        for ncomp in (5, 3):                                      # A_U-like and A_v/B-like records
            nloc = md.K * Nfq
            trace = np.zeros((nloc + pl["nghost"], ncomp))
            gid = np.arange(offsets[rank] * Nfq, offsets[rank + 1] * Nfq)
            trace[:nloc] = gid[:, None] * 10.0 + np.arange(ncomp)[None, :]
            send = trace[pl["sendlist"]].copy()                   # k_pack
            rec = ncomp * 8
            ws = torch.zeros(trace.nbytes + send.nbytes, dtype=torch.uint8)
            ws[:trace.nbytes] = torch.from_numpy(trace.view(np.uint8).reshape(-1))
            ws[trace.nbytes:] = torch.from_numpy(send.view(np.uint8).reshape(-1))
            segs = [[(peer, trace.nbytes + so * rec, sc * rec, (nloc + ro) * rec, rc * rec) for (peer, so, sc, ro, rc) in pl["nbrs"]]]
            hx = HaloExchanger(segs)
            HaloExchanger.wait(hx.start(ws, 0))
            out = ws[:trace.nbytes].numpy().view(np.float64).reshape(-1, ncomp)
            got = out[pl["mapP"]]
            gP = md.mapP.flatten(order="F") - 1
            expect = gP[:, None] * 10.0 + np.arange(ncomp)[None, :]
            ok = ok and np.array_equal(got, expect)
        ret[rank] = bool(ok)
    finally:
        dist.destroy_process_group()


def test_gloo_world_size_2_exchange():
    world = 2
    port = _free_port()
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_gloo_worker, args=(world, port, ret), nprocs=world, join=True)
    assert ret[0] is True and ret[1] is True
