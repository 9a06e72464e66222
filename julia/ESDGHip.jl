# ESDGHip.jl -- thin `ccall` layer over libesdg_hip.so (include/esdg_hip.h) for the reference's Julia drivers.
#
# STATUS: reviewed but UN-RUN source.  Julia is not available in the pipeline that builds and tests this repository;
# the identical C ABI is exercised by the Python ctypes binding (esdg_cns_amd/_lib.py, engine.py) and by the C drivers
# in examples/c/.  `__init__` checks every struct mirror below against the library (`esdg_abi_sizeof`), so a layout
# mistake fails at load time instead of corrupting memory.  INTEGRATION.md shows the edits to each driver script.
#
# No CUDA.jl / AMDGPU.jl: device memory is handled by the library (esdg_dmalloc / esdg_memcpy_*), the state stays on
# the device between right-hand-side evaluations.
module ESDGHip

export Engine, CnsEngine, HexEngine, upload!, download!, rhs!, rhsRK!, lsrk!, rhs_lsrk!, lsrk45_step!,
       dopri45_attempt!, dopri45_next_dt, setup_errors!, l2_error, nodal_error, boundary_velocity_error, rhs, destroy!,
       Setup, setup_uniform_quad_mesh, setup_uniform_hex_mesh, setup_quad, setup_hex, setup_array, setup_map, setup_destroy!,
       set_device, device_count, comm_unique_id, comm_init!, comm_size, comm_allreduce, comm_destroy!, halo_exchange!, halo_wait!

const LIB = joinpath(@__DIR__, "..", "esdg_cns_amd", "libesdg_hip.so")   # built by `python -m esdg_cns_amd.build`

# ---- mirrors of the C structs (field order and types of include/esdg_hip.h) ------------------------------------
struct OpsT
    N::Int32; Np::Int32; Nq::Int32; Nfq::Int32
    Qrhskew::Ptr{Float64}; Qshskew::Ptr{Float64}; Ph::Ptr{Float64}; wq::Ptr{Float64}; wf::Ptr{Float64}
    Ef::Ptr{Float64}; Lf::Ptr{Float64}
    Vq::Ptr{Float64}; Pq::Ptr{Float64}; VhP::Ptr{Float64}; LIFT::Ptr{Float64}
    Vf::Ptr{Float64}; Dr::Ptr{Float64}; Ds::Ptr{Float64}
end
struct MeshT
    K::Int64; geo_ld::Int32
    rxJ::Ptr{Float64}; sxJ::Ptr{Float64}; ryJ::Ptr{Float64}; syJ::Ptr{Float64}
    J::Ptr{Float64}; wJq::Ptr{Float64}; nxJ::Ptr{Float64}; nyJ::Ptr{Float64}; sJ::Ptr{Float64}
    mapP::Ptr{Int64}; mapB::Ptr{Int64}; NmapB::Int64; bkind::Ptr{UInt8}
    elem_offset::Int64; Kglobal::Int64; nranks::Int32; rank::Int32; rank_offsets::Ptr{Int64}
    vlid::Ptr{Float64}
end
struct PhysT
    formulation::Int32; lf_scale::Float64; inviscid_dissp::Int32; viscous_dissp::Int32; BCTYPE::Int32
    Re::Float64; mu::Float64; lambda::Float64; Pr::Float64
    inflow_rho::Float64; inflow_u::Float64; inflow_v::Float64; inflow_p::Float64
end
struct HexOpsT
    N::Int32; Nq::Int32; Nfq::Int32
    Qrhskew::Ptr{Float64}; Qshskew::Ptr{Float64}; Qthskew::Ptr{Float64}
    Ph::Ptr{Float64}; Lf::Ptr{Float64}; Ef::Ptr{Float64}; wq::Ptr{Float64}; wf::Ptr{Float64}
end
struct HexMeshT
    K::Int64; geo_ld::Int32
    rxJ::Ptr{Float64}; sxJ::Ptr{Float64}; txJ::Ptr{Float64}; ryJ::Ptr{Float64}; syJ::Ptr{Float64}; tyJ::Ptr{Float64}
    rzJ::Ptr{Float64}; szJ::Ptr{Float64}; tzJ::Ptr{Float64}
    J::Ptr{Float64}; wJq::Ptr{Float64}; nxJ::Ptr{Float64}; nyJ::Ptr{Float64}; nzJ::Ptr{Float64}; sJ::Ptr{Float64}
    mapP::Ptr{Int64}; elem_offset::Int64; Kglobal::Int64; nranks::Int32; rank::Int32; rank_offsets::Ptr{Int64}
end
struct ErrOpsT
    Nq2::Int32; Vq2::Ptr{Float64}; wq2::Ptr{Float64}; x::Ptr{Float64}; y::Ptr{Float64}; J::Ptr{Float64}
    Vf::Ptr{Float64}; wf::Ptr{Float64}
end

const EULER_COLLOCATED, CNS_MODAL, EULER_MODAL, EULER_HEX_COLLOCATED = Int32(0), Int32(1), Int32(2), Int32(3)
const EXACT_VORTEX, EXACT_BECKER = Int32(0), Int32(1)

check(rc) = rc == 0 || error(unsafe_string(ccall((:esdg_last_error, LIB), Cstring, ())))

function __init__()
    for (T, name) in ((OpsT, "esdg_ops_t"), (MeshT, "esdg_mesh_t"), (PhysT, "esdg_phys_t"), (HexOpsT, "esdg_hex_ops_t"),
                      (HexMeshT, "esdg_hex_mesh_t"), (ErrOpsT, "esdg_err_ops_t"))
        want = ccall((:esdg_abi_sizeof, LIB), Int64, (Cstring,), name)
        want == sizeof(T) || error("ESDGHip: $name is $want bytes in libesdg_hip.so but $(sizeof(T)) here (ABI drift)")
    end
end

dense(A) = Matrix{Float64}(A)          # rd/ops fields may be SparseMatrixCSC (SetupDG.jl:59-70)

"""
Element-index sharding of one engine (one process per GPU): the local elements are the global elements
`rank_offsets[rank+1]+1 : rank_offsets[rank+2]` (0-based offsets, length nranks+1), `md` holds only those and `md.mapP`
GLOBAL 1-based indices into (Nfq x Kglobal) -- what `setup_quad(...; e_begin, e_end)` produces.  `NoShard` = one rank.
"""
struct Shard
    rank::Int; nranks::Int; rank_offsets::Vector{Int64}; Kglobal::Int
end
const NoShard = Shard(0, 1, Int64[], 0)
"(elem_offset, Kglobal, nranks, rank, rank_offsets) of esdg_mesh_t / esdg_hex_mesh_t"
function shard_fields(s::Shard, K, keep)
    s.nranks == 1 && return (Int64(0), Int64(K), Int32(1), Int32(0), Ptr{Int64}(C_NULL))
    push!(keep, s.rank_offsets)
    (s.rank_offsets[s.rank + 1], Int64(s.Kglobal), Int32(s.nranks), Int32(s.rank), pointer(s.rank_offsets))
end

mutable struct Engine
    ctx::Ptr{Cvoid}; ws::Ptr{Cvoid}; K::Int; Np::Int; nfld::Int
    Qd::Ptr{Float64}; rhsd::Ptr{Float64}; resd::Ptr{Float64}     # device state [nfld][K][Np]
    keep::Vector{Any}                                            # host arrays the create call read from
end

dmalloc(n) = ccall((:esdg_dmalloc, LIB), Ptr{Cvoid}, (Csize_t,), n)

function finish_engine(ctx, K, Np, keep)
    nfld = Int(ccall((:esdg_num_fields, LIB), Cint, (Ptr{Cvoid},), ctx))
    nb = ccall((:esdg_workspace_bytes, LIB), Csize_t, (Ptr{Cvoid},), ctx)
    ws = dmalloc(nb)
    check(ccall((:esdg_bind_workspace, LIB), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Csize_t), ctx, ws, nb))
    n = nfld * K * Np * 8
    e = Engine(ctx, ws, K, Np, nfld, Ptr{Float64}(dmalloc(n)), Ptr{Float64}(dmalloc(n)), Ptr{Float64}(dmalloc(n)), keep)
    z = zeros(K * Np * nfld)                                    # resQ starts from zero (dg2D_euler_quad.jl:84)
    check(ccall((:esdg_memcpy_h2d, LIB), Cint, (Ptr{Cvoid}, Ptr{Float64}, Csize_t), e.resd, z, n))
    e
end

function destroy!(e::Engine)
    for p in (e.Qd, e.rhsd, e.resd, e.ws)
        ccall((:esdg_dfree, LIB), Cint, (Ptr{Cvoid},), p)
    end
    ccall((:esdg_destroy, LIB), Cint, (Ptr{Cvoid},), e.ctx)
    e.ctx = C_NULL
end

"Euler-quad driver (examples/dg2D_euler_quad.jl, after line 91): Q lives at the Gauss nodes; LF factor .5 (:165)."
function Engine(rd, md, ops, Ef; lf_scale = 0.5, shard = NoShard)
    Qrhskew, Qshskew, _, _, _, Ph, Lf = ops                       # dg2D_euler_quad.jl:91
    k = Any[dense(Qrhskew), dense(Qshskew), dense(Ph), Vector{Float64}(rd.wq), Vector{Float64}(rd.wf), dense(Ef), dense(Lf)]
    Nq, Nfq = length(rd.wq), length(rd.wf)
    o = OpsT(Int32(round(Int, sqrt(Nq)) - 1), Nq, Nq, Nfq, pointer.(k[1:7])...,
             C_NULL, C_NULL, C_NULL, C_NULL, C_NULL, C_NULL, C_NULL)
    m = MeshT(md.K, size(md.rxJ, 1), pointer(md.rxJ), pointer(md.sxJ), pointer(md.ryJ), pointer(md.syJ),
              pointer(md.J), pointer(md.wJq), pointer(md.nxJ), pointer(md.nyJ), pointer(md.sJ),
              pointer(md.mapP), C_NULL, 0, C_NULL, shard_fields(shard, md.K, k)..., C_NULL)
    p = PhysT(EULER_COLLOCATED, lf_scale, 1, 0, 0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0)
    ctx = Ref{Ptr{Cvoid}}()
    GC.@preserve k md check(ccall((:esdg_create, LIB), Cint, (Ref{OpsT}, Ref{MeshT}, Ref{PhysT}, Ref{Ptr{Cvoid}}), o, m, p, ctx))
    finish_engine(ctx[], md.K, Nq, k)
end

"""
CNS drivers (dg2D_CNS_cavity_optimized.jl after line 90, dg2D_CNS_convergence_test.jl, dg2D_CNS_modalESDG.jl):
`ops = (Qrhskew,Qshskew,VhP,Ph,LIFT,Vq)`, Q = LGL nodal values.  Walls: `md.mapB` with `lid[i] = 1` where the node
lies on the lid (init_BC_funs :139-148) and optionally the lid velocity per mapB entry (`vlid`, convergence_test :76).
BCTYPE = 4: `md.mapB = [leftwall; rightwall]`, `lid[i] = 1` on the inflow side, `inflow = (rhoL,uL,vL,pL)`,
`viscous_dissp = false` (dg2D_CNS_modalESDG.jl:161-217).
"""
function CnsEngine(rd, md, ops; Re, mu, lambda, Pr, BCTYPE = 1, inviscid_dissp = true, viscous_dissp = true,
                   lid = nothing, vlid = nothing, inflow = (0.0, 0.0, 0.0, 0.0), lf_scale = 0.25, formulation = CNS_MODAL,
                   shard = NoShard)
    Qrhskew, Qshskew, VhP, Ph, LIFT, Vq = ops
    k = Any[dense(Qrhskew), dense(Qshskew), dense(Ph), Vector{Float64}(rd.wq), Vector{Float64}(rd.wf),
            dense(Vq), dense(rd.Pq), dense(VhP), dense(LIFT), dense(rd.Vf), dense(rd.Dr), dense(rd.Ds)]
    Np, Nq, Nfq = size(rd.Pq, 1), length(rd.wq), length(rd.wf)
    o = OpsT(Int32(round(Int, sqrt(Np)) - 1), Np, Nq, Nfq, pointer(k[1]), pointer(k[2]), pointer(k[3]), pointer(k[4]), pointer(k[5]),
             C_NULL, C_NULL, pointer(k[6]), pointer(k[7]), pointer(k[8]), pointer(k[9]), pointer(k[10]), pointer(k[11]), pointer(k[12]))
    mapB = Vector{Int64}(md.mapB)
    bk = lid === nothing ? UInt8[] : Vector{UInt8}(lid)
    vl = vlid === nothing ? Float64[] : Vector{Float64}(vlid)
    push!(k, mapB, bk, vl)
    m = MeshT(md.K, size(md.rxJ, 1), pointer(md.rxJ), pointer(md.sxJ), pointer(md.ryJ), pointer(md.syJ),
              pointer(md.J), pointer(md.wJq), pointer(md.nxJ), pointer(md.nyJ), pointer(md.sJ),
              pointer(md.mapP), isempty(mapB) ? C_NULL : pointer(mapB), length(mapB), isempty(bk) ? C_NULL : pointer(bk),
              shard_fields(shard, md.K, k)..., isempty(vl) ? C_NULL : pointer(vl))
    p = PhysT(formulation, lf_scale, Int32(inviscid_dissp), Int32(viscous_dissp), BCTYPE, Re, mu, lambda, Pr, inflow...)
    ctx = Ref{Ptr{Cvoid}}()
    GC.@preserve k md check(ccall((:esdg_create, LIB), Cint, (Ref{OpsT}, Ref{MeshT}, Ref{PhysT}, Ref{Ptr{Cvoid}}), o, m, p, ctx))
    finish_engine(ctx[], md.K, Np, k)
end

"dg3D_euler_hex.jl after line 98; lf_scale = the literal 0*.25 of line 193"
function HexEngine(rd, md, Qrhskew, Qshskew, Qthskew, Ph, Lf, Ef; lf_scale = 0.0, shard = NoShard)
    k = Any[dense(Qrhskew), dense(Qshskew), dense(Qthskew), dense(Ph), dense(Lf), dense(Ef),
            Vector{Float64}(rd.wq), Vector{Float64}(rd.wf)]
    Nq, Nfq = length(rd.wq), length(rd.wf)
    o = HexOpsT(Int32(round(Int, cbrt(Nq)) - 1), Nq, Nfq, pointer.(k)...)
    m = HexMeshT(md.K, size(md.rxJ, 1), pointer.((md.rxJ, md.sxJ, md.txJ, md.ryJ, md.syJ, md.tyJ, md.rzJ, md.szJ, md.tzJ,
                 md.J, md.wJq, md.nxJ, md.nyJ, md.nzJ, md.sJ))..., pointer(md.mapP), shard_fields(shard, md.K, k)...)
    p = PhysT(EULER_HEX_COLLOCATED, lf_scale, 1, 0, 0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0)
    ctx = Ref{Ptr{Cvoid}}()
    GC.@preserve k md check(ccall((:esdg_create_hex, LIB), Cint, (Ref{HexOpsT}, Ref{HexMeshT}, Ref{PhysT}, Ref{Ptr{Cvoid}}), o, m, p, ctx))
    finish_engine(ctx[], md.K, Nq, k)
end

# ---- state movement ------------------------------------------------------------------------------------------
"upload the driver's Q (tuple/vector of nfld Matrix{Float64}) -- once, before the time loop"
function upload!(e::Engine, Q)
    nb = e.K * e.Np * 8
    for f in 1:e.nfld
        check(ccall((:esdg_memcpy_h2d, LIB), Cint, (Ptr{Cvoid}, Ptr{Float64}, Csize_t), e.Qd + (f - 1) * nb, Q[f], nb))
    end
end
function download!(Q, e::Engine; from = e.Qd)
    nb = e.K * e.Np * 8
    for f in 1:e.nfld
        check(ccall((:esdg_memcpy_d2h, LIB), Cint, (Ptr{Float64}, Ptr{Cvoid}, Csize_t), Q[f], from + (f - 1) * nb, nb))
    end
    Q
end

# ---- the hot path --------------------------------------------------------------------------------------------
"device-resident `rhsQ,rhstest = rhs(Q,md,ops,euler_fluxes,compute_rhstest)` (dg2D_euler_quad.jl:141, hex :167): rhs -> e.rhsd"
function rhs!(e::Engine; compute_rhstest = false)
    check(ccall((:esdg_rhs, LIB), Cint, (Ptr{Cvoid}, Ptr{Float64}, Ptr{Float64}, Ptr{Cvoid}), e.ctx, e.Qd, e.rhsd, C_NULL))
    compute_rhstest || return 0.0
    diag = zeros(2)
    check(ccall((:esdg_rhstest, LIB), Cint, (Ptr{Cvoid}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Cvoid}),
                e.ctx, e.Qd, e.rhsd, diag, C_NULL))
    diag[1]
end

"""
`rhsQ,rhstest,rhstest_visc = rhsRK!(...)` (dg2D_CNS_cavity_optimized.jl:955-972): rhs -> e.rhsd, returns the two scalars.
On a sharded engine (comm_init! done) both are this rank's shares: `comm_allreduce(e, [rhstest, rhstest_visc])` adds them.
"""
function rhsRK!(e::Engine; compute_rhstest = false)
    rhstest = rhs!(e; compute_rhstest = compute_rhstest)
    compute_rhstest || return 0.0, 0.0
    # rhstest_visc = sum(wJq .* v .* rhs_viscous) + visc_test (:962-969): the viscous part alone goes to a scratch buffer
    scratch = Ptr{Float64}(dmalloc(e.nfld * e.K * e.Np * 8))
    check(ccall((:esdg_set_parts, LIB), Cint, (Ptr{Cvoid}, Cint), e.ctx, 2))
    check(ccall((:esdg_rhs, LIB), Cint, (Ptr{Cvoid}, Ptr{Float64}, Ptr{Float64}, Ptr{Cvoid}), e.ctx, e.Qd, scratch, C_NULL))
    check(ccall((:esdg_set_parts, LIB), Cint, (Ptr{Cvoid}, Cint), e.ctx, 3))
    d = zeros(2); vt = zeros(1)
    check(ccall((:esdg_rhstest, LIB), Cint, (Ptr{Cvoid}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Cvoid}),
                e.ctx, e.Qd, scratch, d, C_NULL))
    check(ccall((:esdg_viscous_entropy_test, LIB), Cint, (Ptr{Cvoid}, Ptr{Float64}, Ptr{Float64}, Ptr{Cvoid}), e.ctx, e.Qd, vt, C_NULL))
    ccall((:esdg_dfree, LIB), Cint, (Ptr{Cvoid},), scratch)
    rhstest, d[1] + vt[1]
end

"resQ = a*resQ + dt*rhsQ ; Q += b*resQ  (dg2D_euler_quad.jl:204-205) on the device"
lsrk!(e::Engine, a, b, dt) = check(ccall((:esdg_lsrk_update, LIB), Cint,
    (Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Cdouble, Cdouble, Cdouble, Int64, Ptr{Cvoid}),
    e.Qd, e.resd, e.rhsd, a, b, dt, e.nfld * e.K * e.Np, C_NULL))

"RHS fused with the RK stage that consumes it (no rhs array is written)"
rhs_lsrk!(e::Engine, a, b, dt) = check(ccall((:esdg_rhs_lsrk, LIB), Cint,
    (Ptr{Cvoid}, Ptr{Float64}, Ptr{Float64}, Cdouble, Cdouble, Cdouble, Ptr{Cvoid}), e.ctx, e.Qd, e.resd, a, b, dt, C_NULL))

"the five stages of dg2D_euler_quad.jl:200-206 in one call"
lsrk45_step!(e::Engine, dt) = check(ccall((:esdg_lsrk45_step, LIB), Cint,
    (Ptr{Cvoid}, Ptr{Float64}, Ptr{Float64}, Cdouble, Ptr{Cvoid}), e.ctx, e.Qd, e.resd, dt, C_NULL))

"""
Stages 2..7 and the Hairer estimate of one DOPRI45 attempt (cavity_optimized.jl:1002-1021).  `k` = 7 device state
buffers (Ptr{Float64}), k[1] = rhs(Q) on entry (FSAL).  Returns errEst; the candidate state is in `Qtmp`.
"""
function dopri45_attempt!(e::Engine, Qtmp::Ptr{Float64}, k::Vector{Ptr{Float64}}, dt; errTol = 1e-5)
    err = zeros(1)
    check(ccall((:esdg_dopri45_attempt, LIB), Cint,
                (Ptr{Cvoid}, Ptr{Float64}, Ptr{Float64}, Ptr{Ptr{Float64}}, Cdouble, Cdouble, Ptr{Float64}, Ptr{Cvoid}),
                e.ctx, e.Qd, Qtmp, k, dt, errTol, err, C_NULL))
    err[1]
end
dopri45_next_dt(dt, dt0, errEst, prevErrEst, i) =
    ccall((:esdg_dopri45_next_dt, LIB), Cdouble, (Cdouble, Cdouble, Cdouble, Cdouble, Int64), dt, dt0, errEst, prevErrEst, i)

# ---- error blocks of the scripts (INTEGRATION.md section 2c) ----------------------------------------------------
"Vq2: (Nq2 x Np) state nodes -> error quadrature (fold Pq in when Q lives at the Gauss nodes); x, y, J at the state's nodes"
function setup_errors!(e::Engine, x, y, J; Vq2 = nothing, wq2 = nothing, Vf = nothing, wf = nothing)
    k = Any[dense(x), dense(y), dense(J)]
    pv(a) = a === nothing ? Ptr{Float64}(C_NULL) : (push!(k, Array{Float64}(a)); pointer(k[end]))
    o = ErrOpsT(Vq2 === nothing ? 0 : size(Vq2, 1), pv(Vq2), pv(wq2), pointer(k[1]), pointer(k[2]), pointer(k[3]), pv(Vf), pv(wf))
    GC.@preserve k check(ccall((:esdg_error_setup, LIB), Cint, (Ptr{Cvoid}, Ref{ErrOpsT}), e.ctx, o))
end
function l2_error(e::Engine, t; exact = EXACT_VORTEX, par = nothing)
    out = zeros(5)
    check(ccall((:esdg_error_l2, LIB), Cint, (Ptr{Cvoid}, Ptr{Float64}, Int32, Ptr{Float64}, Cdouble, Ptr{Float64}, Ptr{Cvoid}),
                e.ctx, e.Qd, exact, par === nothing ? C_NULL : Vector{Float64}(par), t, out, C_NULL))
    out[1]
end
"(L1err, Linferr) of dg2D_CNS_modalESDG.jl:745-771; par = [v_0, v_1, v_01, m_0, kappa/m_0/cv, v_inf]"
function nodal_error(e::Engine, t, par; exact = EXACT_BECKER)
    out = zeros(14)
    check(ccall((:esdg_error_nodal, LIB), Cint, (Ptr{Cvoid}, Ptr{Float64}, Int32, Ptr{Float64}, Cdouble, Ptr{Float64}, Ptr{Cvoid}),
                e.ctx, e.Qd, exact, Vector{Float64}(par), t, out, C_NULL))
    out[1], out[2]
end
"err of dg2D_CNS_convergence_test.jl:1055-1080: (as the script executes it, with all three terms)"
function boundary_velocity_error(e::Engine, Jf)
    out = zeros(5)
    check(ccall((:esdg_error_boundary_velocity, LIB), Cint, (Ptr{Cvoid}, Ptr{Float64}, Cdouble, Ptr{Float64}, Ptr{Cvoid}),
                e.ctx, e.Qd, Jf, out, C_NULL))
    out[1], out[2]
end

# ---- set-up inside the library (INTEGRATION.md section 3) -------------------------------------------------------
# The reference's build_periodic_boundary_maps is O(Nbfaces^2) (src/node_map_functions.jl:66-128) and connect_mesh sorts
# all faces: neither gets to 2048^2 elements.  esdg_setup_* (csrc/esdg_setup.cpp) builds the same RefElemData / MeshData /
# driver operators in O(K log K), for an element range when the mesh is sharded; arrays come back by the SetupDG field
# names (column-major, maps 1-based Int64), ready for `Engine` / `CnsEngine` / `HexEngine` through `setup_fill`.
mutable struct Setup
    h::Ptr{Cvoid}
end
setup_error() = unsafe_string(ccall((:esdg_setup_last_error, LIB), Cstring, ()))
scheck(rc) = rc == 0 || error(setup_error())

"uniform_quad_mesh(Kx,Ky) (src/UniformQuadMesh.jl:25-50): (VX, VY, EToV) on [-1,1]^2, EToV (K x 4), 1-based"
function setup_uniform_quad_mesh(Kx, Ky)
    VX, VY = zeros((Kx + 1) * (Ky + 1)), zeros((Kx + 1) * (Ky + 1))
    EToV = zeros(Int64, Kx * Ky, 4)
    scheck(ccall((:esdg_setup_uniform_quad_mesh, LIB), Cint, (Cint, Cint, Ptr{Float64}, Ptr{Float64}, Ptr{Int64}), Kx, Ky, VX, VY, EToV))
    VX, VY, EToV
end
"uniform_hex_mesh(Kx,Ky,Kz) (src/UniformHexMesh.jl:25-80): (VX, VY, VZ, EToV), EToV (K x 8)"
function setup_uniform_hex_mesh(Kx, Ky, Kz)
    n = (Kx + 1) * (Ky + 1) * (Kz + 1)
    VX, VY, VZ = zeros(n), zeros(n), zeros(n)
    EToV = zeros(Int64, Kx * Ky * Kz, 8)
    scheck(ccall((:esdg_setup_uniform_hex_mesh, LIB), Cint, (Cint, Cint, Cint, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Int64}),
                 Kx, Ky, Kz, VX, VY, VZ, EToV))
    VX, VY, VZ, EToV
end
"init_reference_quad + init_mesh + periodic patch + driver operators for elements e_begin+1:e_end (0-based range; e_end <= 0: all)"
function setup_quad(N, formulation, VX, VY, EToV; periodic = true, e_begin = 0, e_end = 0)
    h = Ref{Ptr{Cvoid}}()
    scheck(ccall((:esdg_setup_quad, LIB), Cint,
                 (Cint, Cint, Ptr{Float64}, Ptr{Float64}, Int64, Ptr{Int64}, Int64, Cint, Int64, Int64, Ref{Ptr{Cvoid}}),
                 N, formulation, VX, VY, length(VX), EToV, size(EToV, 1), periodic, e_begin, e_end, h))
    Setup(h[])
end
"init_reference_hex + init_mesh (3D) + periodic patch + the operators of dg3D_euler_hex.jl:34-98"
function setup_hex(N, VX, VY, VZ, EToV; periodic = true, e_begin = 0, e_end = 0)
    h = Ref{Ptr{Cvoid}}()
    scheck(ccall((:esdg_setup_hex, LIB), Cint,
                 (Cint, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Int64, Ptr{Int64}, Int64, Cint, Int64, Int64, Ref{Ptr{Cvoid}}),
                 N, VX, VY, VZ, length(VX), EToV, size(EToV, 1), periodic, e_begin, e_end, h))
    Setup(h[])
end
"array by its SetupDG field name (\"Vq\", \"x\", \"rxJ\", \"Qrhskew\", ...): a Matrix{Float64} VIEW of library memory"
function setup_array(s::Setup, name)
    r, c = Ref{Int64}(), Ref{Int64}()
    p = ccall((:esdg_setup_array, LIB), Ptr{Float64}, (Ptr{Cvoid}, Cstring, Ref{Int64}, Ref{Int64}), s.h, name, r, c)
    p == C_NULL && error("no array $name: " * setup_error())
    unsafe_wrap(Array, p, (Int(r[]), Int(c[])))
end
"index map by name (\"mapM\", \"mapP\", \"mapB\", \"FToF\"), 1-based Int64"
function setup_map(s::Setup, name)
    n = Ref{Int64}()
    p = ccall((:esdg_setup_map, LIB), Ptr{Int64}, (Ptr{Cvoid}, Cstring, Ref{Int64}), s.h, name, n)
    p == C_NULL && error("no map $name: " * setup_error())
    unsafe_wrap(Array, p, Int(n[]))
end
"esdg_ops_t / esdg_mesh_t pointing into the set-up object, then esdg_create: an Engine without a Julia-side SetupDG"
function Engine(s::Setup, phys::PhysT; shard = NoShard)
    o, m = Ref{OpsT}(), Ref{MeshT}()
    scheck(ccall((:esdg_setup_fill, LIB), Cint, (Ptr{Cvoid}, Ref{OpsT}, Ref{MeshT}), s.h, o, m))
    keep = Any[s]
    mm = m[]
    if shard.nranks > 1                                        # esdg_setup_fill leaves one rank; patch the shard in
        f = shard_fields(shard, mm.K, keep)
        mm = MeshT(mm.K, mm.geo_ld, mm.rxJ, mm.sxJ, mm.ryJ, mm.syJ, mm.J, mm.wJq, mm.nxJ, mm.nyJ, mm.sJ, mm.mapP, mm.mapB, mm.NmapB,
                   mm.bkind, f[1], f[2], f[3], f[4], f[5], mm.vlid)
    end
    ctx = Ref{Ptr{Cvoid}}()
    GC.@preserve keep check(ccall((:esdg_create, LIB), Cint, (Ref{OpsT}, Ref{MeshT}, Ref{PhysT}, Ref{Ptr{Cvoid}}), o, mm, phys, ctx))
    finish_engine(ctx[], Int(mm.K), Int(o[].Np), keep)
end
function HexEngine(s::Setup; lf_scale = 0.0, shard = NoShard)
    o, m = Ref{HexOpsT}(), Ref{HexMeshT}()
    scheck(ccall((:esdg_setup_fill_hex, LIB), Cint, (Ptr{Cvoid}, Ref{HexOpsT}, Ref{HexMeshT}), s.h, o, m))
    keep = Any[s]
    mm = m[]
    if shard.nranks > 1
        f = shard_fields(shard, mm.K, keep)
        mm = HexMeshT(mm.K, mm.geo_ld, mm.rxJ, mm.sxJ, mm.txJ, mm.ryJ, mm.syJ, mm.tyJ, mm.rzJ, mm.szJ, mm.tzJ, mm.J, mm.wJq,
                      mm.nxJ, mm.nyJ, mm.nzJ, mm.sJ, mm.mapP, f[1], f[2], f[3], f[4], f[5])
    end
    p = PhysT(EULER_HEX_COLLOCATED, lf_scale, 1, 0, 0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0)
    ctx = Ref{Ptr{Cvoid}}()
    GC.@preserve keep check(ccall((:esdg_create_hex, LIB), Cint, (Ref{HexOpsT}, Ref{HexMeshT}, Ref{PhysT}, Ref{Ptr{Cvoid}}), o, mm, p, ctx))
    finish_engine(ctx[], Int(mm.K), Int(o[].Nq), keep)
end
function setup_destroy!(s::Setup)
    ccall((:esdg_setup_destroy, LIB), Cint, (Ptr{Cvoid},), s.h)
    s.h = C_NULL
end

# ---- one process per GPU: device selection and the library's RCCL transport (INTEGRATION.md section 4) -----------
device_count() = Int(ccall((:esdg_device_count, LIB), Cint, ()))
set_device(d) = check(ccall((:esdg_set_device, LIB), Cint, (Cint,), d))
"128 bytes of ncclUniqueId: create on ONE rank, hand to the others (MPI.Bcast!, a file, a socket), then comm_init! everywhere"
function comm_unique_id()
    id = zeros(UInt8, 128)
    check(ccall((:esdg_comm_unique_id, LIB), Cint, (Ptr{UInt8},), id))
    id
end
"attach the communicator (collective); afterwards rhs! / rhs_lsrk! / lsrk45_step! / dopri45_attempt! run the sharded schedule"
comm_init!(e::Engine, id::Vector{UInt8}, rank, nranks) =
    check(ccall((:esdg_comm_init, LIB), Cint, (Ptr{Cvoid}, Ptr{UInt8}, Cint, Cint), e.ctx, id, rank, nranks))
comm_size(e::Engine) = Int(ccall((:esdg_comm_size, LIB), Cint, (Ptr{Cvoid},), e.ctx))
comm_destroy!(e::Engine) = check(ccall((:esdg_comm_destroy, LIB), Cint, (Ptr{Cvoid},), e.ctx))
"sum (op = 0) / max (1) / min (2) of a few host doubles over the ranks: rhstest, the DOPRI error norm, dt"
function comm_allreduce(e::Engine, vals::Vector{Float64}; op = 0)
    v = copy(vals)
    check(ccall((:esdg_comm_allreduce, LIB), Cint, (Ptr{Cvoid}, Ptr{Float64}, Cint, Cint, Ptr{Cvoid}), e.ctx, v, length(v), op, C_NULL))
    v
end
"the two transport steps for hosts that drive the phases themselves (esdg_rhs_phase / esdg_halo_pack)"
halo_exchange!(e::Engine, phase) = check(ccall((:esdg_halo_exchange, LIB), Cint, (Ptr{Cvoid}, Cint, Ptr{Cvoid}), e.ctx, phase, C_NULL))
halo_wait!(e::Engine, phase) = check(ccall((:esdg_halo_wait, LIB), Cint, (Ptr{Cvoid}, Cint, Ptr{Cvoid}), e.ctx, phase, C_NULL))

# ---- literal drop-in with host arrays (PCIe-bound; validation only) -------------------------------------------
"`rhs(Q, md, ops, flux_fun, compute_rhstest)` of the Euler drivers with an engine in place of (md, ops): returns (rhsQ, rhstest)"
function rhs(e::Engine, Q, flux_fun = nothing, compute_rhstest = false)
    out = [similar(Q[1]) for _ in 1:e.nfld]
    qp, op = pointer.(collect(Q)), pointer.(out)
    GC.@preserve Q out check(ccall((:esdg_rhs_host, LIB), Cint, (Ptr{Cvoid}, Ptr{Ptr{Float64}}, Ptr{Ptr{Float64}}), e.ctx, qp, op))
    rhstest = 0.0
    if compute_rhstest
        upload!(e, Q)
        rhstest = rhs!(e; compute_rhstest = true)
    end
    Tuple(out), rhstest
end

# ---- the scripts' own signatures ---------------------------------------------------------------------------------
# A driver that deletes its inline `rhs` / `rhsRK!` definitions and says `using ESDGHip.Scripts` keeps every call site
# unchanged.  What the scripts read from globals (`Ef`, `mu`, `lambda`, `Pr`) is handed over once with `bind!`; engines are
# created on first use and cached per (md, ops) pair.  Host arrays in and out per call (PCIe-bound: for validation and
# for runs where the time loop is not worth porting; the device-resident loop is section 2 of INTEGRATION.md).
module Scripts
using ..ESDGHip
export bind!, rhs, rhsRK!, rhs_inviscid!, rhs_viscous!
const CTX = Dict{Symbol,Any}()
const ENGINES = IdDict{Any,Any}()
"globals of the driver the reference functions read: `bind!(rd = rd, Ef = Ef)` (Euler quad / hex), `bind!(rd = rd, mu = mu, lambda = lambda, Pr = Pr)` (CNS)"
bind!(; kw...) = (for (k, v) in kw; CTX[k] = v; end; nothing)
function engine_for(key, make)
    haskey(ENGINES, key) || (ENGINES[key] = make())
    ENGINES[key]
end
"`rhsQ, rhstest = rhs(Q, md, ops, flux_fun, compute_rhstest)` -- dg2D_euler_quad.jl:141 (7-tuple `ops`, 4 fields) and dg3D_euler_hex.jl:167 (5 fields)"
function rhs(Q, md, ops, flux_fun, compute_rhstest = false)
    e = engine_for((md, ops), () -> length(Q) == 5 ?
        ESDGHip.HexEngine(CTX[:rd], md, ops[1], ops[2], ops[3], ops[end-1], ops[end], CTX[:Ef]) :
        ESDGHip.Engine(CTX[:rd], md, ops, CTX[:Ef]))
    ESDGHip.rhs(e, Q, flux_fun, compute_rhstest)
end
"""
`rhsQ, rhstest, rhstest_visc = rhsRK!(Q, rd, md, Re, BCTYPE, ops, euler_fluxes, inviscid_dissp, viscous_dissp, work...)`
-- dg2D_CNS_cavity_optimized.jl:955; the work arrays are accepted and ignored.  Walls: `bind!(lid = ..., vlid = ...)`
as for `CnsEngine`.
"""
function rhsRK!(Q, rd, md, Re, BCTYPE, ops, flux_fun, inviscid_dissp, viscous_dissp, work...)
    e = engine_for((md, ops, BCTYPE), () -> ESDGHip.CnsEngine(rd, md, ops; Re = Re, mu = CTX[:mu], lambda = CTX[:lambda], Pr = CTX[:Pr],
        BCTYPE = BCTYPE, inviscid_dissp = inviscid_dissp, viscous_dissp = viscous_dissp,
        lid = get(CTX, :lid, nothing), vlid = get(CTX, :vlid, nothing)))
    rhsQ, _ = ESDGHip.rhs(e, Q)                     # esdg_rhs_host: the sum of both parts
    ESDGHip.upload!(e, Q)
    rhstest, rhstest_visc = ESDGHip.rhsRK!(e; compute_rhstest = true)
    rhsQ, rhstest, rhstest_visc
end
"the cached CNS engine of a mesh (created by `rhsRK!` / `rhs_inviscid!`, or here from `bind!(ops = ops, ...)`)"
function cns_engine(md, rd, ops, Re, BCTYPE, inviscid_dissp, viscous_dissp)
    engine_for((md, ops, BCTYPE), () -> ESDGHip.CnsEngine(rd, md, ops; Re = Re, mu = CTX[:mu], lambda = CTX[:lambda], Pr = CTX[:Pr],
        BCTYPE = BCTYPE, inviscid_dissp = inviscid_dissp, viscous_dissp = viscous_dissp,
        lid = get(CTX, :lid, nothing), vlid = get(CTX, :vlid, nothing)))
end
"one part of the right-hand side on host arrays: parts = 1 `rhs_inviscid!`, 2 `rhs_viscous!` (esdg_set_parts)"
function part_rhs(e, Q, parts)
    ESDGHip.check(ccall((:esdg_set_parts, ESDGHip.LIB), Cint, (Ptr{Cvoid}, Cint), e.ctx, parts))
    out = try
        first(ESDGHip.rhs(e, Q))
    finally
        ccall((:esdg_set_parts, ESDGHip.LIB), Cint, (Ptr{Cvoid}, Cint), e.ctx, 3)
    end
    out
end
"""
`rhsQ, rhstest = rhs_inviscid!(Q, md, ops, flux_fun, compute_rhstest, inviscid_dissp, VU, Qh, QF, QM, QP, Uf, UP, rhsQ, tmp, tmp2, lam, LFc)`
-- dg2D_CNS_cavity_optimized.jl:447; the twelve work arrays are accepted and ignored.  The script's function reads `rd`, `Re`
and `BCTYPE` from globals: `bind!(rd = rd, Re = Re, BCTYPE = BCTYPE, mu = mu, lambda = lambda, Pr = Pr)` once.
`rhstest` = sum(wJq .* v .* (Vq*rhsQ)) of this part (`:519-526`).
"""
function rhs_inviscid!(Q, md, ops, flux_fun, compute_rhstest, inviscid_dissp, work...)
    e = cns_engine(md, CTX[:rd], ops, CTX[:Re], get(CTX, :BCTYPE, 1), inviscid_dissp, get(CTX, :viscous_dissp, true))
    rhsQ = part_rhs(e, Q, 1)
    rhstest = 0.0
    if compute_rhstest
        ESDGHip.upload!(e, Q)
        ESDGHip.check(ccall((:esdg_set_parts, ESDGHip.LIB), Cint, (Ptr{Cvoid}, Cint), e.ctx, 1))
        rhstest = try ESDGHip.rhs!(e; compute_rhstest = true) finally
            ccall((:esdg_set_parts, ESDGHip.LIB), Cint, (Ptr{Cvoid}, Cint), e.ctx, 3) end
    end
    rhsQ, rhstest
end
"""
`visc_rhsQ, visc_test = rhs_viscous!(Q, md, rd, Re, BCTYPE, viscous_dissp, rhs, VU, VUx, VUy, sigma_x, sigma_y, penalization, Kxx, Kyy, Kxy)`
-- dg2D_CNS_cavity_optimized.jl:749; the ten work arrays are accepted and ignored.  The operator tuple is the one `rhsRK!` /
`rhs_inviscid!` was called with for this mesh, or `bind!(ops = ops)`.  `visc_test` is the second return of the script's
function, the boundary / penalty part of the viscous entropy balance (`:962-969` add it to `rhstest_visc`).
"""
function rhs_viscous!(Q, md, rd, Re, BCTYPE, viscous_dissp, work...)
    ops = nothing
    for k in keys(ENGINES)
        (k isa Tuple && length(k) == 3 && k[1] === md && k[3] == BCTYPE) && (ops = k[2])
    end
    ops === nothing && (ops = CTX[:ops])
    e = cns_engine(md, rd, ops, Re, BCTYPE, get(CTX, :inviscid_dissp, true), viscous_dissp)
    rhsQ = part_rhs(e, Q, 2)
    ESDGHip.upload!(e, Q)
    vt = zeros(1)
    ESDGHip.check(ccall((:esdg_viscous_entropy_test, ESDGHip.LIB), Cint, (Ptr{Cvoid}, Ptr{Float64}, Ptr{Float64}, Ptr{Cvoid}), e.ctx, e.Qd, vt, C_NULL))
    rhsQ, vt[1]
end
end # module Scripts

end # module
