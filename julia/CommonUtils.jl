# CommonUtils.jl stand-in: the names the reference's drivers take from its CommonUtils module
# (/root/reference/src/CommonUtils.jl:14-24), for a driver that switches to the SetupDG stand-in of this directory:
#
#   rk45_coeffs()                        dg2D_euler_quad.jl:94, dg3D_euler_hex.jl:115, cavity_optimized.jl:39  (Carpenter-Kennedy LSRK45)
#   eye(n), speye(n), unzip(a), meshgrid dg2D_euler_quad.jl:76 `[eye(length(wq)); Ef]`; plain Julia
#   geometric_factors(x,y,z,Dr,Ds,Dt)    dg3D_euler_hex.jl:81 (also the 2D method); plain Julia on the driver's own matrices
#   build_periodic_boundary_maps(...)    dg2D_euler_quad.jl:42, dg3D_euler_hex.jl:64: `init_mesh(...; periodic = true)` of the
#                                        SetupDG stand-in has already made mapP periodic, so this returns mapP[mapB] -- what the
#                                        driver's next line (`mapP[mapB] = mapPB`) writes back
#   connect_mesh, build_node_maps        part of the library's set-up (esdg_setup_quad / esdg_setup_hex); not separately callable
#
# Un-run in this pipeline (no Julia).  A driver that keeps the reference's own CommonUtils needs none of this.
module CommonUtils

using LinearAlgebra, SparseArrays

export meshgrid, geometric_factors, build_periodic_boundary_maps, build_periodic_boundary_maps!, rk45_coeffs, unzip, eye, speye

unzip(a) = map(x -> getfield.(a, x), fieldnames(eltype(a)))
eye(n) = Matrix{Float64}(I, n, n)
speye(n) = sparse(1.0I, n, n)

"MATLAB-style meshgrid: X varies along columns, Y along rows"
function meshgrid(vx::AbstractVector, vy::AbstractVector)
    X = [x for _ in vy, x in vx]
    Y = [y for y in vy, _ in vx]
    return X, Y
end
meshgrid(v::AbstractVector) = meshgrid(v, v)

"five-stage fourth-order low-storage Runge-Kutta coefficients of Carpenter and Kennedy: (rk4a, rk4b, rk4c)"
function rk45_coeffs()
    rk4a = [0.0, -567301805773.0 / 1357537059087.0, -2404267990393.0 / 2016746695238.0,
            -3550918686646.0 / 2091501179385.0, -1275806237668.0 / 842570457699.0]
    rk4b = [1432997174477.0 / 9575080441755.0, 5161836677717.0 / 13612068292357.0, 1720146321549.0 / 2090206949498.0,
            3134564353537.0 / 4481467310338.0, 2277821191437.0 / 14882151754819.0]
    rk4c = [0.0, 1432997174477.0 / 9575080441755.0, 2526269341429.0 / 6820363962896.0, 2006345519317.0 / 3224310063776.0,
            2802321613138.0 / 2924317926251.0, 1.0]
    return rk4a, rk4b, rk4c
end

"2D: (rxJ, sxJ, ryJ, syJ, J) of the mapping (r,s) -> (x,y)"
function geometric_factors(x, y, Dr, Ds)
    xr, xs, yr, ys = Dr * x, Ds * x, Dr * y, Ds * y
    J = @. xr * ys - xs * yr
    return ys, -yr, -xs, xr, J
end

"""
3D: (rxJ, sxJ, txJ, ryJ, syJ, tyJ, rzJ, szJ, tzJ, J) in the curl-conservative form of Kopriva (2006), which is what the
reference's `geometric_factors` computes and what keeps the discrete metric identities -- free-stream preservation and entropy
conservation -- on the curved mapping of dg3D_euler_hex.jl (plain cross products of the Jacobian's columns agree with it on
affine meshes only).  Same operations as `setup_dg.geometric_factors_3d` of the Python host, whose free-stream test on the
curved mapping stands in for this un-run file (tests/test_hex_cpu.py).
"""
function geometric_factors(x, y, z, Dr, Ds, Dt)
    function curl(a, b)
        Fr, Fs, Ft = (Dr * a) .* b, (Ds * a) .* b, (Dt * a) .* b
        return Dt * Fs - Ds * Ft, Dr * Ft - Dt * Fr, Ds * Fr - Dr * Fs
    end
    rxJ, sxJ, txJ = curl(y, z)
    ryJ, syJ, tyJ = (-).(curl(x, z))
    rzJ, szJ, tzJ = (-).(curl(y, x))
    xr, xs, xt = Dr * x, Ds * x, Dt * x
    yr, ys, yt = Dr * y, Ds * y, Dt * y
    zr, zs, zt = Dr * z, Ds * z, Dt * z
    J = @. xr * (ys * zt - zs * yt) - yr * (xs * zt - zs * xt) + zr * (xs * yt - ys * xt)
    return rxJ, sxJ, txJ, ryJ, syJ, tyJ, rzJ, szJ, tzJ, J
end

"2D / 3D: the partners of the boundary face nodes across the periodic box.  With the SetupDG stand-in mapP is periodic already."
build_periodic_boundary_maps(xf, yf, LX, LY, NfacesTotal, mapM, mapP, mapB) = mapP[mapB]
build_periodic_boundary_maps(xf, yf, zf, LX, LY, LZ, NfacesTotal, mapM, mapP, mapB) = mapP[mapB]
build_periodic_boundary_maps!(md, rd, args...) = md

end # module CommonUtils
