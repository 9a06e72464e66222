"""Host-side driver of the MI355X RHS engine: wraps the C ABI (include/esdg_hip.h) and mirrors the
reference drivers' operator surface so their time loops transliterate one to one.

  rhs(Q, md, ops, flux_fun, compute_rhstest)        examples/dg2D_euler_quad.jl:141
  rhsRK(Q, rd, md, ops, ...)                        examples/CompressibleNS/dg2D_CNS_cavity_optimized.jl:955
  rhs_hex(Q, md, ops, flux_fun, compute_rhstest)    examples/dg3D_euler_hex.jl:167
  lsrk45 loop                                       examples/dg2D_euler_quad.jl:196-212

torch supplies device memory, streams and torch.distributed only; all arithmetic happens in the
hand-written HIP kernels.  There is no CPU fallback: without the HIP extension or a GPU this raises.
"""
import ctypes as C

import numpy as np
import torch

from . import _lib
from ._lib import check_on as _chk, esdg_hex_mesh_t, esdg_hex_ops_t, esdg_mesh_t, esdg_ops_t, esdg_phys_t

EULER_COLLOCATED, CNS_MODAL, EULER_MODAL, EULER_HEX_COLLOCATED = 0, 1, 2, 3


def _f(a):
    """float64 Fortran-contiguous copy/view (Julia layout)."""
    return np.asfortranarray(np.asarray(a, dtype=np.float64))


def _dp(a):
    return a.ctypes.data_as(_lib.c_double_p)


class HaloExchanger:
    """Face-trace halo exchange over torch.distributed (backend "nccl" == RCCL over xGMI on ROCm;
    "gloo" in the CPU tests).  Works on any 1-D uint8 workspace tensor: segments are byte ranges.
    With a device workspace on the gloo backend (rehearsals of the multi-rank path on a box without enough GPUs
    for RCCL) the segments are staged through host memory; RCCL sends/receives the device ranges directly."""

    def __init__(self, segments, group=None):
        # segments[xch] = list of (peer, send_off, send_bytes, recv_off, recv_bytes)
        self.segments = segments
        self.group = group
        self._backend = None

    def start(self, ws_bytes, xch):
        """Post the sends/receives of exchange `xch` (an index or a list of indices: exchanges produced by the same phase
        go out as ONE batch, i.e. one RCCL group call).  Returns a token for wait()."""
        import torch.distributed as dist
        if self._backend is None:
            self._backend = dist.get_backend(self.group)
        staged = ws_bytes.is_cuda and self._backend == "gloo"
        if staged:
            torch.cuda.current_stream(ws_bytes.device).synchronize()
        ops, copies = [], []
        for x in ([xch] if isinstance(xch, int) else list(xch)):
            for peer, so, sb, ro, rb in self.segments[x]:
                if rb:
                    dst = ws_bytes[ro:ro + rb]
                    buf = torch.empty(rb, dtype=torch.uint8) if staged else dst
                    if staged:
                        copies.append((dst, buf))
                    ops.append(dist.P2POp(dist.irecv, buf, peer, self.group))
                if sb:
                    src = ws_bytes[so:so + sb]
                    ops.append(dist.P2POp(dist.isend, src.cpu() if staged else src, peer, self.group))
        works = dist.batch_isend_irecv(ops) if ops else []
        return [works, copies, False]

    @staticmethod
    def wait(pending):
        """Idempotent: exchanges posted in one batch share their token."""
        if pending[2]:
            return
        works, copies = pending[0], pending[1]
        for w in works:
            w.wait()
        for dst, buf in copies:
            dst.copy_(buf)
        pending[2] = True


def check(rc):
    """For callers that drive the C ABI themselves (tests, tools): raise on a non-zero status, with the message of whichever
    loaded build of the library holds one (the engine's own calls go through _lib.check_on with their library)."""
    if rc != 0:
        msgs = [L.esdg_last_error().decode() for L in _lib._LIBS.values()]
        raise _lib.EsdgError(f"libesdg_hip error {rc}: " + " | ".join(m for m in msgs if m))


class RhsEngine:
    """One esdg_ctx: operators + (local shard of the) mesh resident on one MI355X."""

    def __init__(self, rd, md, ops, formulation, lf_scale=None, inviscid_dissp=True, viscous_dissp=True, BCTYPE=1,
                 Re=1000.0, mu=None, lam=None, Pr=.71, device=None, rank=0, nranks=1, rank_offsets=None, group=None,
                 inflow=None, inflow_nodes=None, vlid=None, ab_hooks=False):
        # ab_hooks: the A/B build of the library (libesdg_hip_ab.so), whose esdg_create reads the ESDG_* environment switches that
        # select partner kernels / geometry modes / schedule variants; the shipped library (default) reads no environment variable
        L = _lib.lib(ab=ab_hooks)
        if not torch.cuda.is_available() or L.esdg_device_count() < 1:
            raise _lib.EsdgError("no MI355X/HIP device visible: the RHS engine has no CPU path")
        self.device = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
        torch.cuda.set_device(self.device)
        self.formulation = formulation
        self.nfld = 5 if formulation == EULER_HEX_COLLOCATED else 4
        keep = self._keep = {}
        if formulation == EULER_HEX_COLLOCATED:
            ctx = self._create_hex(L, rd, md, ops, 0.0 if lf_scale is None else lf_scale, rank, nranks, rank_offsets)
        else:
            ctx = self._create_2d(L, rd, md, ops, formulation, lf_scale, inviscid_dissp, viscous_dissp, BCTYPE, Re, mu, lam, Pr,
                                  rank, nranks, rank_offsets, inflow, inflow_nodes, vlid)
        self.ctx = ctx
        self.L = L
        self.nphases = L.esdg_num_phases(ctx)
        nbytes = int(L.esdg_workspace_bytes(ctx))
        self.ws = torch.zeros(max(nbytes, 256), dtype=torch.uint8, device=self.device)
        _chk(L, L.esdg_bind_workspace(ctx, C.c_void_p(self.ws.data_ptr()), nbytes))

        # halo plan
        self.nranks = nranks
        self.group = group
        self.halo = None
        nn = L.esdg_halo_num_neighbors(ctx)
        self.xinfo = []
        segs = []
        for x in range(L.esdg_num_exchanges(ctx)):
            a, b, nc = C.c_int32(), C.c_int32(), C.c_int32()
            _chk(L, L.esdg_exchange_info(ctx, x, C.byref(a), C.byref(b), C.byref(nc)))
            self.xinfo.append((a.value, b.value, nc.value))
            s = []
            for n in range(nn):
                peer = C.c_int32()
                so, sb, ro, rb = C.c_size_t(), C.c_size_t(), C.c_size_t(), C.c_size_t()
                _chk(L, L.esdg_halo_segment(ctx, x, n, C.byref(peer), C.byref(so), C.byref(sb), C.byref(ro), C.byref(rb)))
                s.append((peer.value, so.value, sb.value, ro.value, rb.value))
            segs.append(s)
        if nn:
            self.halo = HaloExchanger(segs, group)
        lo, hi = C.c_int64(), C.c_int64()
        _chk(L, L.esdg_interior_range(ctx, C.byref(lo), C.byref(hi)))
        self.interior = (lo.value, hi.value)
        # overlap exchanges with the interior elements (tensor / hex kernels); ESDG_NO_OVERLAP=1 restores phase-by-phase
        import os
        self.overlap = bool(L.esdg_uses_tensor_kernels(ctx)) and os.environ.get("ESDG_NO_OVERLAP", "0") != "1"
        # "torch": exchanges through torch.distributed P2P driven from Python (HaloExchanger; gloo in the CPU tests);
        # "rccl": the library's own RCCL transport and schedule (attach_rccl), the path a Julia / C host uses
        self.transport = "torch"

    def attach_rccl(self, group=None, loopback=False):
        """Attach the library's RCCL communicator (esdg_comm_init): afterwards rhs_into / rhs_lsrk_fused hand the whole
        sharded evaluation -- overlapped phases, packs, grouped ncclSend/ncclRecv on the library's comm stream -- to
        esdg_rhs / esdg_rhs_lsrk.  The 128-byte ncclUniqueId is created on rank 0 and broadcast through `group`
        (any torch.distributed backend; only the bootstrap uses it).  loopback=True: one-GPU rehearsal, see
        include/esdg_hip.h:esdg_comm_set_loopback."""
        L, ctx = self.L, self.ctx
        idb = (C.c_ubyte * 128)()
        if loopback:
            _chk(L, L.esdg_comm_set_loopback(ctx, 1))
            _chk(L, L.esdg_comm_unique_id(idb))
            _chk(L, L.esdg_comm_init(ctx, idb, 0, 1))
        else:
            import torch.distributed as dist
            rank = dist.get_rank(group)
            on_dev = dist.get_backend(group) == "nccl"
            t = torch.zeros(128, dtype=torch.uint8, device=self.device if on_dev else "cpu")
            if rank == 0:      # (an error here must not keep rank 0 out of the broadcast its peers are waiting in)
                try:
                    _chk(L, L.esdg_comm_unique_id(idb))
                    t.copy_(torch.frombuffer(bytearray(bytes(idb)), dtype=torch.uint8))
                except Exception:  # noqa: BLE001
                    t.zero_()
            dist.broadcast(t, src=0, group=group)
            raw = bytes(t.cpu().numpy().tobytes())
            if not any(raw):
                raise RuntimeError("rank 0 could not create the RCCL unique id (esdg_comm_unique_id)")
            _chk(L, L.esdg_comm_init(ctx, C.create_string_buffer(raw, 128), rank, dist.get_world_size(group)))
        self.transport = "rccl"
        return int(L.esdg_comm_size(ctx))

    def allreduce(self, vals, op="sum"):
        """Sum / max / min of a few host doubles over the ranks of the library's communicator (esdg_comm_allreduce)."""
        a = (C.c_double * len(vals))(*[float(v) for v in vals])
        _chk(self.L, self.L.esdg_comm_allreduce(self.ctx, a, len(vals), {"sum": 0, "max": 1, "min": 2}[op], self._stream()))
        return list(a)

    def _create_hex(self, L, rd, md, ops, lf_scale, rank, nranks, rank_offsets):
        """esdg_create_hex from the arrays examples/dg3D_euler_hex.jl holds when it calls `rhs` (:167)."""
        keep = self._keep
        Nq, Nfq = rd.wq.size, rd.wf.size
        self.Np, self.Nq, self.Nfq, self.K = Nq, Nq, Nfq, int(md.K)
        o = esdg_hex_ops_t()
        o.N, o.Nq, o.Nfq = rd.N, Nq, Nfq
        for n in ("Qrhskew", "Qshskew", "Qthskew", "Ph", "Lf", "Ef"):
            keep[n] = _f(ops[n])
            setattr(o, n, _dp(keep[n]))
        keep["wq"], keep["wf"] = _f(rd.wq), _f(rd.wf)
        o.wq, o.wf = _dp(keep["wq"]), _dp(keep["wf"])
        m = esdg_hex_mesh_t()
        m.K = int(md.K)
        m.geo_ld = int(md.rxJ.shape[0])
        for n in ("rxJ", "sxJ", "txJ", "ryJ", "syJ", "tyJ", "rzJ", "szJ", "tzJ", "J", "wJq", "nxJ", "nyJ", "nzJ", "sJ"):
            keep["m_" + n] = _f(getattr(md, n))
            setattr(m, n, _dp(keep["m_" + n]))
        keep["mapP"] = np.asfortranarray(np.asarray(md.mapP, dtype=np.int64))
        m.mapP = keep["mapP"].ctypes.data_as(_lib.c_int64_p)
        m.elem_offset = int(getattr(md, "elem_offset", 0))
        m.Kglobal = int(getattr(md, "Kglobal", md.K))
        m.nranks, m.rank = int(nranks), int(rank)
        if nranks > 1:
            keep["ro"] = np.ascontiguousarray(np.asarray(rank_offsets, dtype=np.int64))
            m.rank_offsets = keep["ro"].ctypes.data_as(_lib.c_int64_p)
        p = esdg_phys_t()
        p.formulation = EULER_HEX_COLLOCATED
        p.lf_scale = float(lf_scale)
        ctx = C.c_void_p()
        _chk(L, L.esdg_create_hex(C.byref(o), C.byref(m), C.byref(p), C.byref(ctx)))
        return ctx

    def _create_2d(self, L, rd, md, ops, formulation, lf_scale, inviscid_dissp, viscous_dissp, BCTYPE, Re, mu, lam, Pr,
                   rank, nranks, rank_offsets, inflow=None, inflow_nodes=None, vlid=None):
        keep = self._keep
        modal = formulation != EULER_COLLOCATED
        Nq, Nfq = rd.wq.size, rd.wf.size
        Np = rd.Pq.shape[0] if modal else Nq
        self.Np, self.Nq, self.Nfq, self.K = Np, Nq, Nfq, int(md.K)

        o = esdg_ops_t()
        o.N, o.Np, o.Nq, o.Nfq = rd.N, Np, Nq, Nfq
        names = ["Qrhskew", "Qshskew", "Ph"] + (["VhP", "LIFT", "Vq"] if modal else ["Ef", "Lf"])
        for n in names:
            keep[n] = _f(ops[n])
            setattr(o, n, _dp(keep[n]))
        keep["wq"], keep["wf"] = _f(rd.wq), _f(rd.wf)
        o.wq, o.wf = _dp(keep["wq"]), _dp(keep["wf"])
        if modal:
            for n in ("Pq", "Vf", "Dr", "Ds"):
                keep[n] = _f(getattr(rd, n))
                setattr(o, n, _dp(keep[n]))

        m = esdg_mesh_t()
        m.K = int(md.K)
        m.geo_ld = int(md.rxJ.shape[0])
        for n in ("rxJ", "sxJ", "ryJ", "syJ", "J", "wJq", "nxJ", "nyJ", "sJ"):
            keep["m_" + n] = _f(getattr(md, n))
            setattr(m, n, _dp(keep["m_" + n]))
        keep["mapP"] = np.asfortranarray(np.asarray(md.mapP, dtype=np.int64))
        m.mapP = keep["mapP"].ctypes.data_as(_lib.c_int64_p)
        mapB = np.asarray(getattr(md, "mapB", np.zeros(0)), dtype=np.int64)
        if BCTYPE == 4:
            # closures of the shock-tube driver (dg2D_CNS_modalESDG.jl:161-217): md.mapB = the x-side boundary nodes
            # (they may carry periodic partners in mapP); Dirichlet inflow on `inflow_nodes` (default: the x-min side)
            if inflow is None:
                raise ValueError("BCTYPE 4 needs inflow=(rho, u, v, p)")
            keep["mapB"] = np.ascontiguousarray(mapB)
            loc = mapB - 1 - int(getattr(md, "elem_offset", 0)) * Nfq
            inside = (loc >= 0) & (loc < md.K * Nfq)
            if inflow_nodes is None:
                xb = np.full(mapB.size, np.inf)
                xb[inside] = md.xf.flatten(order="F")[loc[inside]]
                kinds = np.abs(xb - md.VX.min()) < 1e-12
            else:
                kinds = np.isin(mapB, np.asarray(inflow_nodes, dtype=np.int64))
            keep["bkind"] = np.ascontiguousarray(kinds.astype(np.uint8))
            m.mapB = keep["mapB"].ctypes.data_as(_lib.c_int64_p)
            m.NmapB = int(mapB.size)
            m.bkind = keep["bkind"].ctypes.data_as(_lib.c_uint8_p)
            mapB = np.zeros(0, dtype=np.int64)
        elif mapB.size:
            # md.mapB survives the periodic patch of the drivers (mapP[mapB] = mapPB); only nodes that still map to
            # themselves are walls
            locB = mapB - 1 - int(getattr(md, "elem_offset", 0)) * Nfq
            okB = (locB >= 0) & (locB < md.K * Nfq)
            selfmap = np.zeros(mapB.size, dtype=bool)
            selfmap[okB] = keep["mapP"].flatten(order="F")[locB[okB]] == mapB[okB]
            mapB = mapB[selfmap]
        if mapB.size:
            # wall nodes; lid = boundary nodes with |y - 1| < 1e-12 (init_BC_funs, cavity_optimized.jl:139-148)
            keep["mapB"] = np.ascontiguousarray(mapB)
            loc = mapB - 1 - int(getattr(md, "elem_offset", 0)) * Nfq
            inside = (loc >= 0) & (loc < md.K * Nfq)
            yb = np.zeros(mapB.size)
            yb[inside] = md.yf.flatten(order="F")[loc[inside]]
            keep["bkind"] = np.ascontiguousarray(((np.abs(yb - 1) < 1e-12) & inside).astype(np.uint8))
            m.mapB = keep["mapB"].ctypes.data_as(_lib.c_int64_p)
            m.NmapB = int(mapB.size)
            m.bkind = keep["bkind"].ctypes.data_as(_lib.c_uint8_p)
            if vlid is not None:
                # lid velocity: a callable of the lid nodes' x (dg2D_CNS_convergence_test.jl:72-76) or one value per mapB entry
                xb = np.zeros(mapB.size)
                xb[inside] = md.xf.flatten(order="F")[loc[inside]]
                vl = vlid(xb) if callable(vlid) else np.broadcast_to(np.asarray(vlid, dtype=np.float64), mapB.shape)
                keep["vlid"] = np.ascontiguousarray(vl, dtype=np.float64)
                m.vlid = _dp(keep["vlid"])
        elif BCTYPE != 4:
            m.mapB, m.NmapB, m.bkind = None, 0, None
        m.elem_offset = int(getattr(md, "elem_offset", 0))
        m.Kglobal = int(getattr(md, "Kglobal", md.K))
        m.nranks, m.rank = int(nranks), int(rank)
        if nranks > 1:
            keep["ro"] = np.ascontiguousarray(np.asarray(rank_offsets, dtype=np.int64))
            m.rank_offsets = keep["ro"].ctypes.data_as(_lib.c_int64_p)

        p = esdg_phys_t()
        p.formulation = formulation
        p.lf_scale = float(lf_scale if lf_scale is not None else (.5 if formulation == EULER_COLLOCATED else .25))
        p.inviscid_dissp, p.viscous_dissp, p.BCTYPE = int(inviscid_dissp), int(viscous_dissp), int(BCTYPE)
        mu = 1.0 / Re if mu is None else mu
        lam = -2.0 / 3.0 * mu if lam is None else lam
        p.Re, p.mu, p.lambda_, p.Pr = float(Re), float(mu), float(lam), float(Pr)
        if inflow is not None:
            p.inflow_rho, p.inflow_u, p.inflow_v, p.inflow_p = (float(x) for x in inflow)

        ctx = C.c_void_p()
        _chk(L, L.esdg_create(C.byref(o), C.byref(m), C.byref(p), C.byref(ctx)))
        return ctx

    def __del__(self):
        try:
            if getattr(self, "ctx", None):
                self.L.esdg_destroy(self.ctx)
                self.ctx = None
        except Exception:
            pass

    # -- state helpers ------------------------------------------------------------------------
    def upload(self, Q):
        """tuple/list of 4 (Np x K) host matrices -> device tensor [4][K][Np]."""
        h = np.ascontiguousarray(np.stack([np.asarray(q, dtype=np.float64).T for q in Q]))
        return torch.from_numpy(h).to(self.device)

    @staticmethod
    def download(Qd):
        """device tensor [4][K][Np] -> list of 4 Fortran (Np x K) host matrices."""
        h = Qd.detach().cpu().numpy()
        return [np.asfortranarray(h[f].T) for f in range(h.shape[0])]

    def new_state(self):
        return torch.zeros((self.nfld, self.K, self.Np), dtype=torch.float64, device=self.device)

    # -- the hot path -------------------------------------------------------------------------
    def _stream(self):
        return C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

    def _phases(self, full, ranged):
        """Drive the phases of one RHS evaluation with its halo exchanges.  `full(ph)` launches phase ph on all local
        elements (and packs what it produces); `ranged(ph, e0, n)` launches it on elements [e0, e0+n) without packing.
        With an interior range (esdg_interior_range) the exchanges overlap computation: phase 0 runs on the boundary
        ranges first, later phases on the interior first (see include/esdg_hip.h)."""
        L, ctx = self.L, self.ctx
        pending = {}
        lo, hi = self.interior
        overlap = self.overlap and hi > lo and (hi - lo) < self.K
        for ph in range(self.nphases):
            incoming = [x for x, (a, b, _) in enumerate(self.xinfo) if b == ph and x in pending]
            outgoing = [x for x, (a, b, _) in enumerate(self.xinfo) if a == ph]
            if not overlap:
                for x in incoming:
                    HaloExchanger.wait(pending.pop(x))
                full(ph)
            else:
                if ph > 0:
                    ranged(ph, lo, hi - lo)                      # interior: needs no ghost data
                for x in incoming:
                    HaloExchanger.wait(pending.pop(x))
                ranged(ph, 0, lo)                                # boundary ranges
                ranged(ph, hi, self.K - hi)
                for x in outgoing:
                    _chk(L, L.esdg_halo_pack(ctx, x, self._stream()))
            if outgoing:                                         # one batch (one RCCL group) per producing phase
                tok = self.halo.start(self.ws, outgoing)
                for x in outgoing:
                    pending[x] = tok
            if overlap and ph == 0:
                ranged(ph, lo, hi - lo)                          # interior of phase 0 overlaps the first exchange

    def rhs_into(self, Qd, out):
        """One RHS evaluation, state resident on device; asynchronous on torch's current stream."""
        assert Qd.is_contiguous() and out.is_contiguous() and Qd.dtype == torch.float64
        L, ctx, s = self.L, self.ctx, self._stream()
        if self.halo is None or self.transport == "rccl":
            _chk(L, L.esdg_rhs(ctx, C.c_void_p(Qd.data_ptr()), C.c_void_p(out.data_ptr()), s))
            return out
        q, o = C.c_void_p(Qd.data_ptr()), C.c_void_p(out.data_ptr())
        self._phases(lambda ph: _chk(L, L.esdg_rhs_phase(ctx, ph, q, o, self._stream())),
                     lambda ph, e0, n: _chk(L, L.esdg_rhs_phase_range(ctx, ph, e0, n, q, o, self._stream())))
        return out

    def rhs(self, Qd):
        return self.rhs_into(Qd, torch.empty_like(Qd))

    def rhstest(self, Qd, rhsd):
        """sum(wJq .* v(u) .* rhs) over the local elements (euler_quad.jl:186-191)."""
        diag = (C.c_double * 2)()
        _chk(self.L, self.L.esdg_rhstest(self.ctx, C.c_void_p(Qd.data_ptr()), C.c_void_p(rhsd.data_ptr()), diag, self._stream()))
        return diag[0]

    # ---- error functionals of the drivers, evaluated on the device ----
    def setup_errors(self, rd, md, Vq2=None, wq2=None, boundary=False):
        """Upload what the error functionals need: the error quadrature (Vq2, wq2; setup_dg.error_quadrature), the
        coordinates and J at the state's nodes, and rd.Vf/rd.wf for the boundary-velocity error."""
        from ._lib import esdg_err_ops_t
        keep = self._keep
        colloc = self.formulation == EULER_COLLOCATED
        e = esdg_err_ops_t()
        xs, ys, Js = (md.xq, md.yq, rd.Vq @ md.J) if colloc else (md.x, md.y, md.J)
        for n, a in (("x", xs), ("y", ys), ("J", Js)):
            keep["e_" + n] = _f(a)
            setattr(e, n, _dp(keep["e_" + n]))
        if Vq2 is not None:
            V = Vq2 @ rd.Pq if colloc else Vq2          # "project solution back to GLL nodes", dg2D_euler_quad.jl:215
            keep["e_Vq2"], keep["e_wq2"] = _f(V), _f(wq2)
            e.Nq2, e.Vq2, e.wq2 = V.shape[0], _dp(keep["e_Vq2"]), _dp(keep["e_wq2"])
        if boundary:
            keep["e_Vf"], keep["e_wf"] = _f(rd.Vf), _f(rd.wf)
            e.Vf, e.wf = _dp(keep["e_Vf"]), _dp(keep["e_wf"])
        _chk(self.L, self.L.esdg_error_setup(self.ctx, C.byref(e)))

    @staticmethod
    def _par(par):
        return None if par is None else (C.c_double * 6)(*[float(v) for v in par])

    def _reduce(self, sums=None, maxs=None):
        """Sum / max of per-rank partial results over the engine's process group (sharded meshes)."""
        if self.nranks <= 1:
            return sums, maxs
        import torch.distributed as dist
        dev = self.device if dist.get_backend(self.group) == "nccl" else "cpu"
        res = []
        for vals, op in ((sums, dist.ReduceOp.SUM), (maxs, dist.ReduceOp.MAX)):
            if vals is None:
                res.append(None)
                continue
            t = torch.tensor(vals, dtype=torch.float64, device=dev)
            dist.all_reduce(t, op=op, group=self.group)
            res.append(t.tolist())
        return res[0], res[1]

    def l2_error(self, Qd, t, exact=0, par=None):
        """(L2err, per-field sums of squares) against vortex (exact=0) or the Becker shock (exact=1, par) at time t,
        dg2D_euler_quad.jl:214-233; on a sharded mesh the sums are added over the ranks before the square root."""
        out = (C.c_double * 5)()
        _chk(self.L, self.L.esdg_error_l2(self.ctx, C.c_void_p(Qd.data_ptr()), exact, self._par(par), float(t), out, self._stream()))
        sums, _ = self._reduce(sums=list(out)[1:])
        return float(np.sqrt(sum(sums))) if self.nranks > 1 else out[0], sums

    def nodal_error(self, Qd, t, exact=1, par=None):
        """(L1err, Linferr, raw) of dg2D_CNS_modalESDG.jl:745-771 (sums and maxima reduced over the ranks)."""
        out = (C.c_double * 14)()
        _chk(self.L, self.L.esdg_error_nodal(self.ctx, C.c_void_p(Qd.data_ptr()), exact, self._par(par), float(t), out, self._stream()))
        raw = list(out)[2:]
        if self.nranks <= 1:
            return out[0], out[1], raw
        sums, maxs = self._reduce(sums=[raw[4 * c + i] for c in range(3) for i in (0, 1)],
                                  maxs=[raw[4 * c + i] for c in range(3) for i in (2, 3)])
        raw = [v for c in range(3) for v in (sums[2 * c], sums[2 * c + 1], maxs[2 * c], maxs[2 * c + 1])]
        return (sum(raw[4 * c] / raw[4 * c + 1] for c in range(3)), sum(raw[4 * c + 2] / raw[4 * c + 3] for c in range(3)), raw)

    def boundary_velocity_error(self, Qd, Jf):
        """dg2D_CNS_convergence_test.jl:1055-1080 (Jf = 2/K1D there) -> (err as executed by Julia: the u_2 term only,
        err as written: all three terms, the three sums); see include/esdg_hip.h."""
        out = (C.c_double * 5)()
        _chk(self.L, self.L.esdg_error_boundary_velocity(self.ctx, C.c_void_p(Qd.data_ptr()), float(Jf), out, self._stream()))
        sums, _ = self._reduce(sums=list(out)[2:])
        return float(np.sqrt(sums[0])), float(np.sqrt(sum(sums))), sums

    def check_state(self, Qd):
        """(min rho, min p) over the local nodal values; the reference raises DomainError where these are <= 0."""
        out = (C.c_double * 2)()
        _chk(self.L, self.L.esdg_check_state(self.ctx, C.c_void_p(Qd.data_ptr()), out, self._stream()))
        return out[0], out[1]

    def set_parts(self, parts):
        """1 = rhs_inviscid! only, 2 = rhs_viscous! only, 3 = rhsRK! (default)."""
        _chk(self.L, self.L.esdg_set_parts(self.ctx, int(parts)))

    def rhsRK_diagnostics(self, Qd, rhsd=None):
        """(rhstest, rhstest_visc) of rhsRK! (cavity_optimized.jl:958-969) for the state Qd."""
        if rhsd is None:
            rhsd = self.rhs(Qd)
        rt = self.rhstest(Qd, rhsd)
        self.set_parts(2)
        try:
            visc = self.rhs(Qd)
        finally:
            self.set_parts(3)
        vt = C.c_double(0.0)
        _chk(self.L, self.L.esdg_viscous_entropy_test(self.ctx, C.c_void_p(Qd.data_ptr()), C.byref(vt), self._stream()))
        return rt, self.rhstest(Qd, visc) + vt.value

    def rhs_host(self, Q):
        """Literal drop-in on host arrays through esdg_rhs_host (H2D + rhs + D2H)."""
        Qh = [_f(q) for q in Q]
        out = [np.zeros_like(q) for q in Qh]
        qa = (_lib.c_double_p * len(Qh))(*[_dp(q) for q in Qh])
        ra = (_lib.c_double_p * len(Qh))(*[_dp(r) for r in out])
        _chk(self.L, self.L.esdg_rhs_host(self.ctx, qa, ra))
        return out

    # -- time integration (the step either side of the path) -------------------------------------
    def lsrk_update(self, Qd, resd, rhsd, a, b, dt):
        _chk(self.L, self.L.esdg_lsrk_update(C.c_void_p(Qd.data_ptr()), C.c_void_p(resd.data_ptr()), C.c_void_p(rhsd.data_ptr()),
                                      float(a), float(b), float(dt), Qd.numel(), self._stream()))

    def rhs_lsrk_fused(self, Qd, resd, a, b, dt):
        """RHS + low-storage RK stage in one pass: resQ = a*resQ + dt*rhs(Q); Q += b*resQ (no rhs array)."""
        L, ctx, s = self.L, self.ctx, self._stream()
        q, r = C.c_void_p(Qd.data_ptr()), C.c_void_p(resd.data_ptr())
        a, b, dt = float(a), float(b), float(dt)
        if self.halo is None or self.transport == "rccl":
            _chk(L, L.esdg_rhs_lsrk(ctx, q, r, a, b, dt, s))
            return
        self._phases(lambda ph: _chk(L, L.esdg_rhs_phase_lsrk(ctx, ph, q, r, a, b, dt, self._stream())),
                     lambda ph, e0, n: _chk(L, L.esdg_rhs_phase_range_lsrk(ctx, ph, e0, n, q, r, a, b, dt, self._stream())))

    def lsrk45_step_fused(self, Qd, resd, dt, coeffs):
        rk4a, rk4b = coeffs[0], coeffs[1]
        for k in range(5):
            self.rhs_lsrk_fused(Qd, resd, rk4a[k], rk4b[k], dt)

    def lsrk45_step(self, Qd, resd, rhsd, dt, coeffs):
        """for INTRK = 1:5: rhs; resQ = rk4a*resQ + dt*rhs; Q += rk4b*resQ  (euler_quad.jl:200-206)."""
        rk4a, rk4b = coeffs[0], coeffs[1]
        for k in range(5):
            self.rhs_into(Qd, rhsd)
            self.lsrk_update(Qd, resd, rhsd, rk4a[k], rk4b[k], dt)


# -----------------------------------------------------------------------------------------------
# reference-signature wrappers
# -----------------------------------------------------------------------------------------------
def _engine_for(md, rd, ops, formulation, **kw):
    key = ("_esdg_engine", formulation, tuple(sorted(kw.items())))
    cache = md.__dict__.setdefault("_esdg_cache", {})
    if key not in cache:
        cache[key] = RhsEngine(rd, md, ops, formulation, **kw)
    return cache[key]


def rhs(Q, md, ops, flux_fun=None, compute_rhstest=False, rd=None):
    """Drop-in for `rhs(Q,md,ops,flux_fun,compute_rhstest)` of examples/dg2D_euler_quad.jl:141:
    Q = tuple of 4 (Nq x K) matrices at the Gauss nodes; returns (rhsQ, rhstest).  `flux_fun` is
    accepted and ignored (only euler_fluxes exists in the reference).  Host arrays in, host arrays
    out: PCIe-bound, for validation; time loops should keep the state on device via RhsEngine."""
    eng = _engine_for(md, rd if rd is not None else ops["rd"], ops, EULER_COLLOCATED)
    Qd = eng.upload(Q)
    r = eng.rhs(Qd)
    rt = eng.rhstest(Qd, r) if compute_rhstest else 0
    return tuple(eng.download(r)), rt


def rhs_hex(Q, md, ops, flux_fun=None, compute_rhstest=False, rd=None, lf_scale=0.0):
    """Drop-in for `rhs(Q,md,ops,flux_fun,compute_rhstest)` of examples/dg3D_euler_hex.jl:167: Q = tuple of 5
    (Nq x K) matrices at the Gauss nodes; returns (rhsQ, rhstest).  lf_scale is the literal `0*.25` of :193."""
    eng = _engine_for(md, rd if rd is not None else ops["rd"], ops, EULER_HEX_COLLOCATED, lf_scale=lf_scale)
    Qd = eng.upload(Q)
    r = eng.rhs(Qd)
    rt = eng.rhstest(Qd, r) if compute_rhstest else 0
    return tuple(eng.download(r)), rt


def rhsRK(Q, rd, md, ops, Re=1000.0, lam=None, mu=None, Pr=.71, inviscid_dissp=True, viscous_dissp=True, BCTYPE=1):
    """Drop-in for `rhsRK!` of dg2D_CNS_cavity_optimized.jl:955 on quad elements: Q = 4 (Np x K)
    nodal coefficient matrices; returns (rhsQ, rhstest, rhstest_visc)."""
    eng = _engine_for(md, rd, ops, CNS_MODAL, Re=Re, lam=lam, mu=mu, Pr=Pr, inviscid_dissp=inviscid_dissp,
                      viscous_dissp=viscous_dissp, BCTYPE=BCTYPE)
    Qd = eng.upload(Q)
    r = eng.rhs(Qd)
    rt, rtv = eng.rhsRK_diagnostics(Qd, r)
    return tuple(eng.download(r)), rt, rtv
