# SetupDG.jl / UniformQuadMesh.jl / UniformHexMesh.jl stand-ins: the reference's module and function names
# (/root/reference/src/SetupDG.jl:33-36 `init_reference_quad`, `init_reference_hex`, `init_mesh`, `MeshData`, `RefElemData`;
# src/UniformQuadMesh.jl:25 `uniform_quad_mesh`; src/UniformHexMesh.jl:25 `uniform_hex_mesh`) as thin wrappers over the
# library's host-only set-up (esdg_setup_*, ESDGHip.setup_*), for meshes the reference's own set-up does not get to: its
# build_periodic_boundary_maps is O(Nbfaces^2) and connect_mesh sorts all faces.  A driver that keeps the reference's SetupDG
# needs none of this -- the engines take its arrays as they are (INTEGRATION.md sections 1-2).
#
#     push!(LOAD_PATH, "<repo>/julia"); using ESDGHip, SetupDG, UniformQuadMesh
#     VX, VY, EToV = uniform_quad_mesh(Kx, Ky)
#     rd = init_reference_quad(N)                          # Gauss collocation: init_reference_quad(N, gauss_quad(0,0,N)) is what the
#     md = init_mesh((VX, VY), EToV, rd)                   # Euler driver passes; the second argument only selects the formulation
#
# Un-run in this pipeline (no Julia); tests/test_abi.py checks statically that every ccall'd symbol exists in the header.
# One difference in kind: the reference builds the reference element first and the mesh second; the library builds both in
# one call, so `init_reference_*` only records the request and `init_mesh` fills `rd` and `md` (same field names, column-major
# Float64 matrices, 1-based Int64 maps -- views of library memory that live as long as `md.setup`).
module SetupDG

using ..ESDGHip: ESDGHip, Setup, setup_quad, setup_hex, setup_array, setup_map, EULER_COLLOCATED, CNS_MODAL

export init_reference_quad, init_reference_hex, init_mesh, MeshData, RefElemData, make_periodic!

mutable struct RefElemData
    N::Int; dim::Int; formulation::Int
    V1; r; s; t; rq; sq; tq; rf; sf; tf; wq; wf; nrJ; nsJ; ntJ
    Dr; Ds; Dt; M; Vq; Vf; Pq; LIFT
    RefElemData(N, dim, formulation) = (rd = new(); rd.N = N; rd.dim = dim; rd.formulation = formulation; rd)
end

mutable struct MeshData
    VX; VY; VZ; K::Int; EToV; FToF
    x; y; z; xf; yf; zf; xq; yq; zq; wJq
    mapM; mapP; mapB
    rxJ; sxJ; txJ; ryJ; syJ; tyJ; rzJ; szJ; tzJ; J
    nxJ; nyJ; nzJ; sJ
    setup::Setup          # owns the memory every field above points into
    MeshData() = new()
end

"init_reference_quad(N [, quad_rule]) (SetupDG.jl:218-277): with a Gauss rule of degree N the collocated Euler operators
(dg2D_euler_quad.jl:47-91), without one the modal CNS operators (dg2D_CNS_cavity_optimized.jl:60-105)"
init_reference_quad(N, quad_rule_vol = nothing) = RefElemData(N, 2, quad_rule_vol === nothing ? CNS_MODAL : EULER_COLLOCATED)
"init_reference_hex(N [, quad_rule]) (SetupDG.jl:321-398): the collocated operators of dg3D_euler_hex.jl:34-98"
init_reference_hex(N, quad_rule_vol = nothing) = RefElemData(N, 3, 3)

getf(s, name) = try setup_array(s, name) catch; nothing end

"init_mesh((VX,VY[,VZ]), EToV, rd; periodic = true, e_begin = 0, e_end = 0) (SetupDG.jl:100-140, 402-470 followed by the drivers'
build_periodic_boundary_maps! patch): fills rd and returns md.  e_begin / e_end: 0-based element range of a sharded run."
function init_mesh(VXYZ, EToV, rd::RefElemData; periodic = true, e_begin = 0, e_end = 0)
    s = rd.dim == 2 ? setup_quad(rd.N, rd.formulation, VXYZ[1], VXYZ[2], EToV; periodic = periodic, e_begin = e_begin, e_end = e_end) :
                      setup_hex(rd.N, VXYZ[1], VXYZ[2], VXYZ[3], EToV; periodic = periodic, e_begin = e_begin, e_end = e_end)
    for f in (:V1, :r, :s, :t, :rq, :sq, :tq, :rf, :sf, :tf, :nrJ, :nsJ, :ntJ, :Dr, :Ds, :Dt, :M, :Vq, :Vf, :Pq, :LIFT)
        setfield!(rd, f, getf(s, String(f)))
    end
    rd.wq = vec(setup_array(s, "wq")); rd.wf = vec(setup_array(s, "wf"))
    md = MeshData()
    md.setup = s
    md.VX, md.VY, md.VZ = VXYZ[1], VXYZ[2], length(VXYZ) > 2 ? VXYZ[3] : nothing
    md.EToV = EToV
    for f in (:x, :y, :z, :xf, :yf, :zf, :xq, :yq, :zq, :wJq, :rxJ, :sxJ, :txJ, :ryJ, :syJ, :tyJ, :rzJ, :szJ, :tzJ, :J, :nxJ, :nyJ, :nzJ, :sJ)
        setfield!(md, f, getf(s, String(f)))
    end
    md.K = size(md.J, 2)
    md.mapM, md.mapP, md.mapB = setup_map(s, "mapM"), setup_map(s, "mapP"), setup_map(s, "mapB")
    md.FToF = setup_map(s, "FToF")
    md
end

"the periodic patch is part of init_mesh(...; periodic = true) here; kept so that driver lines calling it still parse"
make_periodic!(md, rd) = md

end # module SetupDG

module UniformQuadMesh
using ..ESDGHip: setup_uniform_quad_mesh
export uniform_quad_mesh
"uniform_quad_mesh(Kx, Ky) (src/UniformQuadMesh.jl:25-50)"
uniform_quad_mesh(Kx, Ky) = setup_uniform_quad_mesh(Kx, Ky)
end

module UniformHexMesh
using ..ESDGHip: setup_uniform_hex_mesh
export uniform_hex_mesh
"uniform_hex_mesh(Kx, Ky, Kz) (src/UniformHexMesh.jl:25-80)"
uniform_hex_mesh(Kx, Ky, Kz) = setup_uniform_hex_mesh(Kx, Ky, Kz)
end
