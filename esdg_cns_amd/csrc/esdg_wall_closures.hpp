// esdg_wall_closures.hpp -- the boundary closures of the CNS drivers at a boundary face node, shared by the tensor kernels
// (kt2_sigma, kt2_rhs, kt3_rhs) and the generic pair-list kernels (esdg_kernels.hip): exterior entropy variables and the
// prescribed stress jump.  (The inviscid mirror state -- impose_BCs_inviscid!, cavity :157-176 -- is three lines at its call sites.)
#pragma once
#include "esdg_dev.hpp"
#include "esdg_devmath.hpp"

namespace esdg {
namespace t2 {
using devmath::rcp_refined;

// Wall closures (meshes with boundary nodes, M.bc != null; bc: 1 wall, 2 lid, 3 Dirichlet inflow, 4 copy).
// wall_exterior_v: exterior projected entropy variables (v2,v3,v4) at a boundary face node from the own ones
// (impose_BCs_entropyvars! cavity :178-216; dg2D_CNS_modalESDG.jl:187-203); gn = (nxJ, nyJ, sJ) of the face.
__device__ __forceinline__ void wall_exterior_v(const double* vf, int bc, double vlid, const double* gn, const Phys& ph, double* vP) {
  if (bc >= 3) {
#pragma unroll
    for (int c = 0; c < 3; ++c) vP[c] = bc == 3 ? ph.inflow_vv[c] : vf[c];
  } else if (ph.BCTYPE == 1) {                            // adiabatic no-slip (vlid: lid velocity at this node)
    vP[0] = bc == 2 ? -vf[0] - 2 * vlid * vf[2] : -vf[0];
    vP[1] = -vf[1];
    vP[2] = vf[2];
  } else if (ph.BCTYPE == 2) {                            // isothermal
    const double theta = 1.0 / (0.3 * 0.3) / 1.4 / 0.4;
    vP[0] = bc == 2 ? 2.0 / theta - vf[0] : -vf[0];
    vP[1] = -vf[1];
    vP[2] = -2.0 / theta - vf[2];
  } else {                                                // slip / reflective
    const double is = rcp_refined(gn[2]);
    const double nx = gn[0] * is, ny = gn[1] * is;
    const double vn = vf[0] * nx + vf[1] * ny;
    vP[0] = vf[0] - 2 * vn * nx;
    vP[1] = vf[1] - 2 * vn * ny;
    vP[2] = vf[2];
  }
}

// wall_stress_jump: the stress jump impose_BCs_stress! (:218-262; modalESDG :205-216) prescribes at a boundary face node,
// from the own face values of sigma_x (fx), sigma_y (fy) and their normal component sn
__device__ __forceinline__ void wall_stress_jump(const double* sn, const double* fx, const double* fy, int bc, double vlid, const double* gn,
                                                 const Phys& ph, double* sj) {
  sj[0] = 0.0; sj[1] = 0.0; sj[2] = 0.0;
  if (bc >= 3 || ph.BCTYPE == 2) return;
  if (ph.BCTYPE == 1) {
    sj[2] = bc == 2 ? -sn[2] + vlid * sn[0] : -sn[2];
  } else {
    const double is = rcp_refined(gn[2]);
    const double n1 = gn[0] * is, n2 = gn[1] * is;
    const double snx = fx[0] * n1 + fx[1] * n2, sny = fy[0] * n1 + fy[1] * n2;
    sj[0] = .5 * ((-2 * fx[0] + 2 * n1 * snx) * gn[0] + (-2 * fy[0] + 2 * n1 * sny) * gn[1]);
    sj[1] = .5 * ((-2 * fx[1] + 2 * n2 * snx) * gn[0] + (-2 * fy[1] + 2 * n2 * sny) * gn[1]);
    sj[2] = -sn[2];
  }
}

}  // namespace t2
}  // namespace esdg
