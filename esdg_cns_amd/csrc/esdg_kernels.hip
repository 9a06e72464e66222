// esdg_kernels.hip -- hand-written gfx950 (MI355X / CDNA4) kernels for the explicit-RK
// right-hand side of the entropy-stable DG Euler / compressible Navier-Stokes solvers.
//
// Algorithm (what, not how) follows the reference drivers:
//   rhs            examples/dg2D_euler_quad.jl:141-194          (collocated Euler)
//   rhs_inviscid!  examples/CompressibleNS/dg2D_CNS_cavity_optimized.jl:447-528
//   rhs_viscous!   ...cavity_optimized.jl:749-849  (dg_grad! :548, viscous_matrices! :613, dg_div! :590)
// but is re-organised for the GPU: every operator apply happens at the tensor-Gauss
// collocation nodes with the sparse operators derived in esdg_api.hip (the modal path is
// Vq -> collocated core -> Pq), flux differencing runs over the list of unordered node pairs
// with a non-zero SBP weight (200 per element at N=4 instead of the 825 the reference loop
// visits), and the only inter-element coupling goes through three compact face-trace buffers
// (A_U, A_v, B) that neighbours gather through mapP.
//
// Work decomposition: one workgroup = 256 threads = 4 waves processes E elements (E chosen per
// degree so that E*pairs ~ k*256); all per-element data lives in LDS between stages; items
// (nodes / face nodes / pairs) x elements are spread over the 256 lanes.  HBM accesses are
// coalesced: a state block of E elements is E*Np contiguous doubles per field.
#include "esdg_dev.hpp"
#include "esdg_wall_closures.hpp"

namespace esdg {

#define ESDG_TPB 256

// elements per workgroup, per N1 = N+1
template <int N1> struct EPB { static constexpr int v = 1; };
template <> struct EPB<2> { static constexpr int v = 16; };
template <> struct EPB<3> { static constexpr int v = 9; };
template <> struct EPB<4> { static constexpr int v = 6; };
template <> struct EPB<5> { static constexpr int v = 4; };
template <> struct EPB<6> { static constexpr int v = 2; };
template <> struct EPB<7> { static constexpr int v = 2; };
template <> struct EPB<8> { static constexpr int v = 1; };

bool supported_degree(int N1) { return N1 >= 2 && N1 <= 8; }

// ---------------------------------------------------------------------------------------------
// pointwise physics (examples/EntropyStableEuler/*.jl), gamma = 1.4
// ---------------------------------------------------------------------------------------------
// logmean.jl:14-28 -- including the reference's series branch (whose coefficients are those of
// the gamma=1.4 polytropic mean, not of the log-mean: reproduced as is).
__device__ __forceinline__ double logmean(double aL, double aR, double logL, double logR) {
  const double da = aR - aL;
  const double aavg = .5 * (aR + aL);
  const double f = da / aavg;
  const double v = f * f;
  if (fabs(f) < 1e-4) return aavg * (1 + v * (-.2 - v * (.0512 - v * 0.026038857142857)));
  return -da / (logL - logR);
}

// euler_fluxes.jl:23-48.  qL/qR = (rho,u,v,beta,log rho,log beta).  GM1 = gamma-1.
template <bool MODAL>
__device__ __forceinline__ void ec_flux(const double* qL, const double* qR, double* Fx, double* Fy) {
  constexpr double GM1 = MODAL ? 0.4 : (1.4 - 1);
  const double rholog = logmean(qL[0], qR[0], qL[4], qR[4]);
  const double betalog = logmean(qL[3], qR[3], qL[5], qR[5]);
  const double rhoavg = .5 * (qL[0] + qR[0]);
  const double uavg = .5 * (qL[1] + qR[1]);
  const double vavg = .5 * (qL[2] + qR[2]);
  const double unorm = qL[1] * qR[1] + qL[2] * qR[2];
  const double pa = rhoavg / (qL[3] + qR[3]);
  const double f4aux = rholog / (2 * GM1 * betalog) + pa + .5 * rholog * unorm;
  Fx[0] = rholog * uavg;
  Fx[1] = Fx[0] * uavg + pa;
  Fx[2] = Fx[0] * vavg;
  Fx[3] = f4aux * uavg;
  Fy[0] = rholog * vavg;
  Fy[1] = Fx[2];
  Fy[2] = Fy[0] * vavg + pa;
  Fy[3] = f4aux * vavg;
}

// euler_variables.jl:79-92 / cavity_optimized.jl:461-467.  Also returns log(rho).
// s = log((g-1) rhoe / rho^g) is evaluated as log((g-1) rhoe) - g log(rho): no pow.
template <bool MODAL>
__device__ __forceinline__ void v_of_u(const double* U, double* V, double& lrho) {
  constexpr double GM1 = MODAL ? 0.4 : (1.4 - 1);
  constexpr double GP1 = MODAL ? 2.4 : (1.4 + 1);
  const double rhoe = U[3] - .5 * (U[1] * U[1] + U[2] * U[2]) / U[0];
  lrho = log(U[0]);
  const double sU = log(GM1 * rhoe) - 1.4 * lrho;
  V[0] = (-U[3] + rhoe * (GP1 - sU)) / rhoe;
  V[1] = U[1] / rhoe;
  V[2] = U[2] / rhoe;
  V[3] = -U[0] / rhoe;
}

// euler_variables.jl:95-120 / cavity_optimized.jl:473-478:
// rhoe(v) = ((g-1)/(-v4)^g)^(1/(g-1)) exp(-s/(g-1)),  s = g - v1 + |vU|^2/(2 v4)
// evaluated as exp((log(g-1) - g log(-v4) - s)/(g-1)): one log + one exp instead of two pow.
template <bool MODAL>
__device__ __forceinline__ void u_of_v(const double* V, double* U) {
  constexpr double GM1 = MODAL ? 0.4 : (1.4 - 1);
  const double vUnorm = V[1] * V[1] + V[2] * V[2];
  const double s = 1.4 - V[0] + vUnorm / (2 * V[3]);
  const double rhoeV = exp((log(GM1) - 1.4 * log(-V[3]) - s) / GM1);
  U[0] = rhoeV * (-V[3]);
  U[1] = rhoeV * V[1];
  U[2] = rhoeV * V[2];
  U[3] = rhoeV * (1 - vUnorm / (2 * V[3]));
}

// conservative -> (rho,u,v,beta,log rho,log beta): betafun euler_variables.jl:30-48 / cavity :484
template <bool MODAL>
__device__ __forceinline__ void prim_logs(const double* U, double* q) {
  constexpr double GM1 = MODAL ? 0.4 : (1.4 - 1);
  q[0] = U[0];
  q[1] = U[1] / U[0];
  q[2] = U[2] / U[0];
  double beta;
  if (MODAL)
    beta = U[0] / (2 * 0.4 * (U[3] - .5 * (U[1] * U[1] + U[2] * U[2]) / U[0]));
  else
    beta = U[0] / (2 * (GM1 * (U[3] - .5 * ((U[1] * U[1] + U[2] * U[2]) / U[0]))));
  q[3] = beta;
  q[4] = log(U[0]);
  q[5] = log(beta);
}

// wavespeed, euler_variables.jl:7-10 (note sqrt(|u_n|), quirk Q1) / cavity_optimized.jl:507
template <bool MODAL>
__device__ __forceinline__ double lf_lambda(const double* U, double nxJ, double nyJ, double sJ) {
  const double rhoUn = (U[1] * nxJ + U[2] * nyJ) / sJ;
  if (MODAL) return fabs(sqrt(fabs(rhoUn / U[0])) + sqrt(1.4 * 0.4 * (U[3] - .5 * rhoUn * rhoUn / U[0]) / U[0]));
  const double p = (1.4 - 1) * (U[3] - .5 * (rhoUn * rhoUn) / U[0]);
  return fabs(sqrt(fabs(rhoUn / U[0])) + sqrt(1.4 * p / U[0]));
}

// viscous_matrices! + sigma accumulation, cavity_optimized.jl:613-645, 786-801
// (lam is the value AFTER the reference's `let lam = -lam` flip).  v = entropy vars at the node,
// tx/ty = d(v2,v3,v4)/dx, /dy.  Output rows 2..4 of sigma_x, sigma_y.
__device__ __forceinline__ void viscous_stress(const double* v, const double* tx, const double* ty, double lam,
                                               double mu, double Pr, double* sx, double* sy) {
  const double v2 = v[1], v3 = v[2], v4 = v[3];
  const double inv = 1 / (v4 * v4 * v4);
  const double l2m = lam + 2.0 * mu;
  const double v44 = v4 * v4;
  const double Kxx22 = inv * -l2m * v44, Kxx24 = inv * l2m * v2 * v4, Kxx33 = inv * -mu * v44,
               Kxx34 = inv * mu * v3 * v4,
               Kxx44 = inv * -(l2m * (v2 * v2) + mu * (v3 * v3) - 1.4 * mu * v4 / Pr);
  const double Kxy23 = inv * -lam * v44, Kxy24 = inv * lam * v3 * v4, Kxy32 = inv * -mu * v44,
               Kxy34 = inv * mu * v2 * v4, Kxy42 = inv * mu * v3 * v4, Kxy43 = inv * lam * v2 * v4,
               Kxy44 = inv * (lam + mu) * (-v2 * v3);
  const double Kyy22 = inv * -mu * v44, Kyy24 = inv * mu * v2 * v4, Kyy33 = inv * -l2m * v44,
               Kyy34 = inv * l2m * v3 * v4,
               Kyy44 = inv * -(l2m * (v3 * v3) + mu * (v2 * v2) - 1.4 * mu * v4 / Pr);
  // sigma_x[row] = sum_col Kxx[row,col] tx[col] + Kxy[row,col] ty[col]
  sx[0] = Kxx22 * tx[0] + Kxx24 * tx[2] + Kxy23 * ty[1] + Kxy24 * ty[2];
  sx[1] = Kxx33 * tx[1] + Kxx34 * tx[2] + Kxy32 * ty[0] + Kxy34 * ty[2];
  sx[2] = Kxx24 * tx[0] + Kxx34 * tx[1] + Kxx44 * tx[2] + Kxy42 * ty[0] + Kxy43 * ty[1] + Kxy44 * ty[2];
  // sigma_y[row] = sum_col Kxy[col,row] tx[col] + Kyy[row,col] ty[col]
  sy[0] = Kxy32 * tx[1] + Kxy42 * tx[2] + Kyy22 * ty[0] + Kyy24 * ty[2];
  sy[1] = Kxy23 * tx[0] + Kxy43 * tx[2] + Kyy33 * ty[1] + Kyy34 * ty[2];
  sy[2] = Kxy24 * tx[0] + Kxy34 * tx[1] + Kxy44 * tx[2] + Kyy24 * ty[0] + Kyy34 * ty[1] + Kyy44 * ty[2];
}

// ---------------------------------------------------------------------------------------------
// shared stages
// ---------------------------------------------------------------------------------------------
// Load the state block of nE elements (4 fields) into LDS, layout [e][f][node].
template <int Np>
__device__ __forceinline__ void load_state(const double* __restrict__ Q, int64_t K, int64_t e0, int nE,
                                           double* __restrict__ dst) {
  const int tid = threadIdx.x;
  const int per = nE * Np;
  for (int idx = tid; idx < 4 * per; idx += ESDG_TPB) {
    const int f = idx / per, r = idx - f * per;
    const int e = r / Np, k = r - e * Np;
    dst[(e * 4 + f) * Np + k] = Q[(int64_t)f * K * Np + e0 * Np + r];
  }
}

// Uq = Vq * Qn (modal) -- dense (Nq x Np) apply per element/field.
template <int Nq, int Np>
__device__ __forceinline__ void interp_to_quad(const Tables& T, int nE, const double* __restrict__ sQn,
                                               double* __restrict__ sU) {
  const int tid = threadIdx.x;
  for (int idx = tid; idx < nE * 4 * Nq; idx += ESDG_TPB) {
    const int ef = idx / Nq, q = idx - ef * Nq;
    const double* x = sQn + ef * Np;
    const double* a = T.Vq + q * Np;
    double s = 0.0;
#pragma unroll
    for (int k = 0; k < Np; ++k) s += a[k] * x[k];
    sU[ef * Nq + q] = s;
  }
}

// ---------------------------------------------------------------------------------------------
// phase 0: entropy projection to the faces -> trace buffers A_U (rho,rhou,rhov,E,lam), A_v (v2..v4)
// (euler_quad.jl:149-151,162-164 ; cavity_optimized.jl:459-478, 501-507, 763-775)
// ---------------------------------------------------------------------------------------------
template <int N1, bool MODAL, bool VISC>
__global__ __launch_bounds__(ESDG_TPB) void k_project(Tables T, MeshDev M, Phys ph, const double* __restrict__ Q,
                                                      double* __restrict__ A_U, double* __restrict__ A_v) {
  constexpr int Nq = N1 * N1, Np = Nq, Nfq = 4 * N1, E = EPB<N1>::v;
  __shared__ double sU[E * 4 * Nq];
  __shared__ double sV[E * 4 * Nq];
  const int tid = threadIdx.x;
  const int64_t e0 = (int64_t)blockIdx.x * E;
  const int nE = (int)min((int64_t)E, M.K - e0);

  load_state<Np>(Q, M.K, e0, nE, MODAL ? sV : sU);
  __syncthreads();
  if (MODAL) {
    interp_to_quad<Nq, Np>(T, nE, sV, sU);
    __syncthreads();
  }
  for (int idx = tid; idx < nE * Nq; idx += ESDG_TPB) {
    const int e = idx / Nq, q = idx - e * Nq;
    double U[4], V[4], lr;
#pragma unroll
    for (int c = 0; c < 4; ++c) U[c] = sU[(e * 4 + c) * Nq + q];
    v_of_u<MODAL>(U, V, lr);
#pragma unroll
    for (int c = 0; c < 4; ++c) sV[(e * 4 + c) * Nq + q] = V[c];
  }
  __syncthreads();
  for (int idx = tid; idx < nE * Nfq; idx += ESDG_TPB) {
    const int e = idx / Nfq, fn = idx - e * Nfq;
    double V[4] = {0, 0, 0, 0};
    for (int t = 0; t < T.wEf; ++t) {
      const double a = T.Ef_val[fn * T.wEf + t];
      const int col = T.Ef_idx[fn * T.wEf + t];
#pragma unroll
      for (int c = 0; c < 4; ++c) V[c] += a * sV[(e * 4 + c) * Nq + col];
    }
    double U[4];
    u_of_v<MODAL>(V, U);
    const double* g = M.fnrm + ((e0 + e) * Nfq + fn) * 3   /* per-node (nxJ, nyJ, sJ), MeshDev::fnrm */;
    const double lam = lf_lambda<MODAL>(U, g[0], g[1], g[2]);
    const int64_t n = (e0 + e) * Nfq + fn;
    double* a = A_U + n * AU_NC;
    a[0] = U[0]; a[1] = U[1]; a[2] = U[2]; a[3] = U[3]; a[4] = lam;
    if (VISC) {
      double* b = A_v + n * AV_NC;
      b[0] = V[1]; b[1] = V[2]; b[2] = V[3];
    }
  }
}

// ---------------------------------------------------------------------------------------------
// viscous helpers used by phase 1 (sigma) and phase 2 (rhs)
// ---------------------------------------------------------------------------------------------
// Face stage: projected entropy variables at own face nodes (Ef*VU), neighbour values from A_v,
// half-jumps .5*(vP - vf) -> sDv[e][c][fn], penalty tau*(vP - vf) -> sPen (optional).
template <int N1, bool WITH_PEN>
__device__ __forceinline__ void visc_face_jumps(const Tables& T, const MeshDev& M, const Phys& ph, int64_t e0, int nE,
                                                const double* __restrict__ sV, const double* __restrict__ A_v,
                                                double* __restrict__ sDv, double* __restrict__ sPen) {
  constexpr int Nq = N1 * N1, Nfq = 4 * N1;
  for (int idx = threadIdx.x; idx < nE * Nfq; idx += ESDG_TPB) {
    const int e = idx / Nfq, fn = idx - e * Nfq;
    double vf[3] = {0, 0, 0};
    for (int t = 0; t < T.wEf; ++t) {
      const double a = T.Ef_val[fn * T.wEf + t];
      const int col = T.Ef_idx[fn * T.wEf + t];
#pragma unroll
      for (int c = 0; c < 3; ++c) vf[c] += a * sV[(e * 4 + c + 1) * Nq + col];
    }
    const int64_t n = (e0 + e) * Nfq + fn;
    const double* vp = A_v + (int64_t)M.mapP[n] * AV_NC;
    double vP[3] = {vp[0], vp[1], vp[2]};
    const int bc = M.bc ? M.bc[n] : 0;
    // boundary node: exterior values by the wall closure (impose_BCs_entropyvars!, cavity :178-216; modalESDG :187-203)
    if (bc) t2::wall_exterior_v(vf, bc, M.vlid ? M.vlid[n] : 1.0, M.fnrm + n * 3, ph, vP);
    const double tau = -1 / ph.Re / vf[2];
    double pen[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const double dv = vP[c] - vf[c];
      sDv[(e * 3 + c) * Nfq + fn] = .5 * dv;
      pen[c] = tau * dv;
    }
    if (WITH_PEN && bc) {   // third component overridden at boundary nodes (:827-837; the form kt3_rhs uses)
      const double dV[3] = {vP[0] - vf[0], vP[1] - vf[1], vP[2] - vf[2]};
      double sq = .5 * (vP[0] + vf[0]) * dV[0] + .5 * (vP[1] + vf[1]) * dV[1];
      if (ph.BCTYPE != 1) sq += dV[2] * dV[2] * .5;
      pen[2] = -tau * sq / vf[2];
    }
    if (WITH_PEN) {
#pragma unroll
      for (int c = 0; c < 3; ++c) sPen[(e * 3 + c) * Nfq + fn] = pen[c];
    }
  }
}

// Volume stage: BR1 gradient of v2..v4 at the Gauss nodes (dg_grad! :548-569 in collocated form),
// then sigma = K(v) grad v (:786-801) -> sS[e][0..2]=sigma_x rows 2..4, [3..5]=sigma_y rows 2..4.
template <int N1>
__device__ __forceinline__ void visc_sigma(const Tables& T, const MeshDev& M, const Phys& ph, int64_t e0, int nE,
                                           const double* __restrict__ sV, const double* __restrict__ sDv,
                                           double* __restrict__ sS) {
  constexpr int Nq = N1 * N1, Nfq = 4 * N1;
  for (int idx = threadIdx.x; idx < nE * Nq; idx += ESDG_TPB) {
    const int e = idx / Nq, q = idx - e * Nq;
    const double* g = M.geo + (e0 + e) * GEO_STRIDE;
    const double rx = g[0], sx_ = g[1], ry = g[2], sy_ = g[3], J = g[4];
    double vr[3] = {0, 0, 0}, vs[3] = {0, 0, 0};
    for (int t = 0; t < T.wD; ++t) {
      const double ar = T.Dr_val[q * T.wD + t], as = T.Ds_val[q * T.wD + t];
      const int cr = T.Dr_idx[q * T.wD + t], cs = T.Ds_idx[q * T.wD + t];
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        vr[c] += ar * sV[(e * 4 + c + 1) * Nq + cr];
        vs[c] += as * sV[(e * 4 + c + 1) * Nq + cs];
      }
    }
    double lx[3] = {0, 0, 0}, ly[3] = {0, 0, 0};
    for (int t = 0; t < T.wLf; ++t) {
      const double a = T.Lf_val[q * T.wLf + t];
      const int fn = T.Lf_idx[q * T.wLf + t];
      const double* gn = M.fnrm + ((e0 + e) * Nfq + fn) * 3;
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        const double d = sDv[(e * 3 + c) * Nfq + fn];
        lx[c] += a * (d * gn[0]);
        ly[c] += a * (d * gn[1]);
      }
    }
    double tx[3], ty[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      tx[c] = ((rx * vr[c] + sx_ * vs[c]) + lx[c]) / J;
      ty[c] = ((ry * vr[c] + sy_ * vs[c]) + ly[c]) / J;
    }
    const double v[4] = {sV[(e * 4 + 0) * Nq + q], sV[(e * 4 + 1) * Nq + q], sV[(e * 4 + 2) * Nq + q],
                         sV[(e * 4 + 3) * Nq + q]};
    double sgx[3], sgy[3];
    viscous_stress(v, tx, ty, -ph.lambda, ph.mu, ph.Pr, sgx, sgy);
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      sS[(e * 6 + c) * Nq + q] = sgx[c];
      sS[(e * 6 + 3 + c) * Nq + q] = sgy[c];
    }
  }
}

// own normal stress at a face node: (Ef*sigma_x)*nxJ + (Ef*sigma_y)*nyJ, rows 2..4
template <int N1>
__device__ __forceinline__ void face_normal_stress(const Tables& T, const double* __restrict__ sS, int e, int fn,
                                                   double nxJ, double nyJ, double* sn, double* fx, double* fy) {
  constexpr int Nq = N1 * N1;
#pragma unroll
  for (int c = 0; c < 3; ++c) { fx[c] = 0.0; fy[c] = 0.0; }
  for (int t = 0; t < T.wEf; ++t) {
    const double a = T.Ef_val[fn * T.wEf + t];
    const int col = T.Ef_idx[fn * T.wEf + t];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      fx[c] += a * sS[(e * 6 + c) * Nq + col];
      fy[c] += a * sS[(e * 6 + 3 + c) * Nq + col];
    }
  }
#pragma unroll
  for (int c = 0; c < 3; ++c) sn[c] = fx[c] * nxJ + fy[c] * nyJ;
}

// ---------------------------------------------------------------------------------------------
// phase 1 (CNS only): sigma = K(v) grad v, normal stress traces -> B   (cavity_optimized.jl:763-813)
// ---------------------------------------------------------------------------------------------
template <int N1>
__global__ __launch_bounds__(ESDG_TPB) void k_sigma(Tables T, MeshDev M, Phys ph, const double* __restrict__ Q,
                                                    const double* __restrict__ A_v, double* __restrict__ B) {
  constexpr int Nq = N1 * N1, Np = Nq, Nfq = 4 * N1, E = EPB<N1>::v;
  __shared__ double sU[E * 4 * Nq];
  __shared__ double sV[E * 4 * Nq];
  __shared__ double sDv[E * 3 * Nfq];
  __shared__ double sS[E * 6 * Nq];
  const int tid = threadIdx.x;
  const int64_t e0 = (int64_t)blockIdx.x * E;
  const int nE = (int)min((int64_t)E, M.K - e0);

  load_state<Np>(Q, M.K, e0, nE, sV);
  __syncthreads();
  interp_to_quad<Nq, Np>(T, nE, sV, sU);
  __syncthreads();
  for (int idx = tid; idx < nE * Nq; idx += ESDG_TPB) {
    const int e = idx / Nq, q = idx - e * Nq;
    double U[4], V[4], lr;
#pragma unroll
    for (int c = 0; c < 4; ++c) U[c] = sU[(e * 4 + c) * Nq + q];
    v_of_u<true>(U, V, lr);
#pragma unroll
    for (int c = 0; c < 4; ++c) sV[(e * 4 + c) * Nq + q] = V[c];
  }
  __syncthreads();
  visc_face_jumps<N1, false>(T, M, ph, e0, nE, sV, A_v, sDv, nullptr);
  __syncthreads();
  visc_sigma<N1>(T, M, ph, e0, nE, sV, sDv, sS);
  __syncthreads();
  for (int idx = tid; idx < nE * Nfq; idx += ESDG_TPB) {
    const int e = idx / Nfq, fn = idx - e * Nfq;
    const double* gn = M.fnrm + ((e0 + e) * Nfq + fn) * 3   /* per-node (nxJ, nyJ, sJ), MeshDev::fnrm */;
    double sn[3], fx[3], fy[3];
    face_normal_stress<N1>(T, sS, e, fn, gn[0], gn[1], sn, fx, fy);
    double* b = B + ((e0 + e) * Nfq + fn) * B_NC;
    b[0] = sn[0]; b[1] = sn[1]; b[2] = sn[2];
  }
}

// ---------------------------------------------------------------------------------------------
// last phase: surface + volume flux differencing (+ viscous divergence and penalty) -> rhs
// ---------------------------------------------------------------------------------------------
template <int N1, bool MODAL, bool VISC>
__global__ __launch_bounds__(ESDG_TPB) void k_rhs(Tables T, MeshDev M, Phys ph, const double* __restrict__ Q,
                                                  const double* __restrict__ A_U, const double* __restrict__ A_v,
                                                  const double* __restrict__ B, double* __restrict__ rhs) {
  constexpr int Nq = N1 * N1, Np = Nq, Nfq = 4 * N1, Nh = Nq + Nfq, E = EPB<N1>::v;
  constexpr int PMAX = N1 * N1 * (N1 + 3);
  extern __shared__ __align__(16) double lds[];
  double* sU = lds;                      // [E][4][Nq]  Uq, later the accumulated collocated rhs (sR)
  double* sV = sU + E * 4 * Nq;          // [E][4][Nq]  nodal staging, then entropy variables
  double* sQh = sV + E * 4 * Nq;         // [E][Nh][6]
  double* sFl = sQh + E * Nh * 6;        // [E][4][Nfq]
  double* sQF = sFl + E * 4 * Nfq;       // [E][4][Nh]
  double* sPV = sQF + E * 4 * Nh;        // [E][P][4]; reused by the viscous stages
  double* sR = sU;
  const int tid = threadIdx.x;
  const int P = T.P;
  const int64_t e0 = (int64_t)blockIdx.x * E;
  const int nE = (int)min((int64_t)E, M.K - e0);

  load_state<Np>(Q, M.K, e0, nE, MODAL ? sV : sU);
  __syncthreads();
  if (MODAL) {
    interp_to_quad<Nq, Np>(T, nE, sV, sU);
    __syncthreads();
  }
  // volume nodes: primitives + logs (and entropy variables for the viscous part)
  for (int idx = tid; idx < nE * Nq; idx += ESDG_TPB) {
    const int e = idx / Nq, q = idx - e * Nq;
    double U[4], qh[6];
#pragma unroll
    for (int c = 0; c < 4; ++c) U[c] = sU[(e * 4 + c) * Nq + q];
    prim_logs<MODAL>(U, qh);
    double* d = sQh + (e * Nh + q) * 6;
#pragma unroll
    for (int c = 0; c < 6; ++c) d[c] = qh[c];
    if (VISC) {
      double V[4], lr;
      v_of_u<MODAL>(U, V, lr);
#pragma unroll
      for (int c = 0; c < 4; ++c) sV[(e * 4 + c) * Nq + q] = V[c];
    }
  }
  // face nodes: own + neighbour traces, interface flux (euler_quad.jl:158-169 / update_flux! :308-324)
  for (int idx = tid; idx < nE * Nfq; idx += ESDG_TPB) {
    const int e = idx / Nfq, fn = idx - e * Nfq;
    const int64_t n = (e0 + e) * Nfq + fn;
    const double* aM = A_U + n * AU_NC;
    const double* aP = A_U + (int64_t)M.mapP[n] * AU_NC;
    const double UM[4] = {aM[0], aM[1], aM[2], aM[3]};
    const double UP[4] = {aP[0], aP[1], aP[2], aP[3]};
    double lamM = aM[4], lamP = aP[4];
    double qM[6], qP[6];
    prim_logs<MODAL>(UM, qM);
    prim_logs<MODAL>(UP, qP);
    double* d = sQh + (e * Nh + Nq + fn) * 6;
#pragma unroll
    for (int c = 0; c < 6; ++c) d[c] = qM[c];
    const double* gn = M.fnrm + ((e0 + e) * Nfq + fn) * 3   /* per-node (nxJ, nyJ, sJ), MeshDev::fnrm */;
    const int bc = M.bc ? M.bc[n] : 0;
    if (bc >= 3) {   // shock-tube closures (dg2D_CNS_modalESDG.jl:168-185): Dirichlet state / copy, lam = lamP = 0
#pragma unroll
      for (int c = 0; c < 6; ++c) qP[c] = bc == 3 ? ph.inflow_q[c] : qM[c];
      lamM = 0.0; lamP = 0.0;
    } else if (bc) {   // wall: mirror state rho+ = rho, beta+ = beta, u+ = u - 2 (u.n) n  (impose_BCs_inviscid!, cavity :157-176)
      const double nx = gn[0] / gn[2], ny = gn[1] / gn[2];
      const double un = qM[1] * nx + qM[2] * ny;
#pragma unroll
      for (int c = 0; c < 6; ++c) qP[c] = qM[c];
      qP[1] = qM[1] - 2 * un * nx;
      qP[2] = qM[2] - 2 * un * ny;
      lamP = lamM;
    }
    double Fx[4], Fy[4];
    ec_flux<MODAL>(qM, qP, Fx, Fy);
    const double LFc = ph.lf_scale * fmax(lamM, lamP) * gn[2];
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      double f = Fx[c] * gn[0] + Fy[c] * gn[1];
      // (the LF jump is Uf[mapP] - Uf, which vanishes at boundary nodes: mapP = self, cavity :511-513)
      if (ph.inviscid_dissp && !bc) f -= LFc * (UP[c] - UM[c]);
      sFl[(e * 4 + c) * Nfq + fn] = f;
    }
  }
  __syncthreads();
  // flux differencing over the non-zero pairs (sparse_hadamard_sum :102-138 / flux_differencing! :326-348)
  for (int idx = tid; idx < nE * P; idx += ESDG_TPB) {
    const int e = idx / P, p = idx - e * P;
    const int i = T.pair_ij[2 * p], j = T.pair_ij[2 * p + 1];
    const double cr = T.pair_c[2 * p], cs = T.pair_c[2 * p + 1];
    const double* g = M.geo + (e0 + e) * GEO_STRIDE;
    const double cx = g[0] * cr + g[1] * cs;
    const double cy = g[2] * cr + g[3] * cs;
    double qi[6], qj[6];
    const double* di = sQh + (e * Nh + i) * 6;
    const double* dj = sQh + (e * Nh + j) * 6;
#pragma unroll
    for (int c = 0; c < 6; ++c) { qi[c] = di[c]; qj[c] = dj[c]; }
    double Fx[4], Fy[4];
    ec_flux<MODAL>(qi, qj, Fx, Fy);
    double* o = sPV + ((size_t)e * PMAX + p) * 4;
#pragma unroll
    for (int c = 0; c < 4; ++c) o[c] = 2 * (cx * Fx[c] + cy * Fy[c]);
  }
  __syncthreads();
  // gather pair values to rows: QF[i] = sum(+val of pairs (i,.)) - sum(val of pairs (.,i))
  for (int idx = tid; idx < nE * Nh; idx += ESDG_TPB) {
    const int e = idx / Nh, i = idx - e * Nh;
    double a[4] = {0, 0, 0, 0};
    for (int t = T.inc_ptr[i]; t < T.inc_ptr[i + 1]; ++t) {
      const unsigned w = T.inc[t];
      const double* o = sPV + ((size_t)e * PMAX + (w & 0x7fff)) * 4;
      if (w & 0x8000) {
#pragma unroll
        for (int c = 0; c < 4; ++c) a[c] -= o[c];
      } else {
#pragma unroll
        for (int c = 0; c < 4; ++c) a[c] += o[c];
      }
    }
#pragma unroll
    for (int c = 0; c < 4; ++c) sQF[(e * 4 + c) * Nh + i] = a[c];
  }
  __syncthreads();
  // collocated rhs: -(PhC*QF + LfC*flux)/J   (euler_quad.jl:170-184 / cavity :514-518)
  for (int idx = tid; idx < nE * Nq; idx += ESDG_TPB) {
    const int e = idx / Nq, q = idx - e * Nq;
    double a[4] = {0, 0, 0, 0};
    for (int t = 0; t < T.wPh; ++t) {
      const double w = T.Ph_val[q * T.wPh + t];
      const int col = T.Ph_idx[q * T.wPh + t];
#pragma unroll
      for (int c = 0; c < 4; ++c) a[c] += w * sQF[(e * 4 + c) * Nh + col];
    }
    for (int t = 0; t < T.wLf; ++t) {
      const double w = T.Lf_val[q * T.wLf + t];
      const int col = T.Lf_idx[q * T.wLf + t];
#pragma unroll
      for (int c = 0; c < 4; ++c) a[c] += w * sFl[(e * 4 + c) * Nfq + col];
    }
    const double J = M.geo[(e0 + e) * GEO_STRIDE + 4];
#pragma unroll
    for (int c = 0; c < 4; ++c) sR[(e * 4 + c) * Nq + q] = -a[c] / J;
  }
  __syncthreads();
  if (VISC) {
    double* sDv = sPV;                    // [E][3][Nfq]
    double* sPen = sDv + E * 3 * Nfq;     // [E][3][Nfq]
    double* sSj = sPen + E * 3 * Nfq;     // [E][3][Nfq]
    double* sS = sSj + E * 3 * Nfq;       // [E][6][Nq]
    visc_face_jumps<N1, true>(T, M, ph, e0, nE, sV, A_v, sDv, sPen);
    __syncthreads();
    visc_sigma<N1>(T, M, ph, e0, nE, sV, sDv, sS);
    __syncthreads();
    // stress jumps .5*((sxP-sxf)*nxJ + (syP-syf)*nyJ) with the neighbour's normal stress from B
    // (the neighbour's outward normal is minus ours), dg_div! :606
    for (int idx = tid; idx < nE * Nfq; idx += ESDG_TPB) {
      const int e = idx / Nfq, fn = idx - e * Nfq;
      const double* gn = M.fnrm + ((e0 + e) * Nfq + fn) * 3   /* per-node (nxJ, nyJ, sJ), MeshDev::fnrm */;
      double sn[3], fx[3], fy[3];
      face_normal_stress<N1>(T, sS, e, fn, gn[0], gn[1], sn, fx, fy);
      const int64_t n = (e0 + e) * Nfq + fn;
      const double* bp = B + (int64_t)M.mapP[n] * B_NC;
      const int bc = M.bc ? M.bc[n] : 0;
      double sj[3] = {.5 * (-bp[0] - sn[0]), .5 * (-bp[1] - sn[1]), .5 * (-bp[2] - sn[2])};
      // boundary node: the jump impose_BCs_stress! prescribes (cavity :218-262; modalESDG :205-216)
      if (bc) t2::wall_stress_jump(sn, fx, fy, bc, M.vlid ? M.vlid[n] : 1.0, gn, ph, sj);
#pragma unroll
      for (int c = 0; c < 3; ++c) sSj[(e * 3 + c) * Nfq + fn] = sj[c];
    }
    __syncthreads();
    // divergence + penalty (dg_div! :590-611, penalty :817-845; the penalty is NOT scaled by 1/J, quirk Q3)
    for (int idx = tid; idx < nE * Nq; idx += ESDG_TPB) {
      const int e = idx / Nq, q = idx - e * Nq;
      const double* g = M.geo + (e0 + e) * GEO_STRIDE;
      double dxr[3] = {0, 0, 0}, dxs[3] = {0, 0, 0}, dyr[3] = {0, 0, 0}, dys[3] = {0, 0, 0};
      for (int t = 0; t < T.wD; ++t) {
        const double ar = T.Dr_val[q * T.wD + t], as = T.Ds_val[q * T.wD + t];
        const int cr = T.Dr_idx[q * T.wD + t], cs = T.Ds_idx[q * T.wD + t];
#pragma unroll
        for (int c = 0; c < 3; ++c) {
          dxr[c] += ar * sS[(e * 6 + c) * Nq + cr];
          dxs[c] += as * sS[(e * 6 + c) * Nq + cs];
          dyr[c] += ar * sS[(e * 6 + 3 + c) * Nq + cr];
          dys[c] += as * sS[(e * 6 + 3 + c) * Nq + cs];
        }
      }
      double sf[3] = {0, 0, 0}, pn[3] = {0, 0, 0};
      for (int t = 0; t < T.wLf; ++t) {
        const double w = T.Lf_val[q * T.wLf + t];
        const int col = T.Lf_idx[q * T.wLf + t];
#pragma unroll
        for (int c = 0; c < 3; ++c) {
          sf[c] += w * sSj[(e * 3 + c) * Nfq + col];
          pn[c] += w * sPen[(e * 3 + c) * Nfq + col];
        }
      }
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        double r = ((g[0] * dxr[c] + g[1] * dxs[c] + g[2] * dyr[c] + g[3] * dys[c]) + sf[c]) / g[4];
        if (ph.viscous_dissp) r += pn[c];
        sR[(e * 4 + c + 1) * Nq + q] += r;
      }
    }
    __syncthreads();
  }
  // back to nodal coefficients (modal: Pq) and store, coalesced per field
  {
    const int per = nE * Np;
    for (int idx = tid; idx < 4 * per; idx += ESDG_TPB) {
      const int f = idx / per, r = idx - f * per;
      const int e = r / Np, k = r - e * Np;
      double s;
      if (MODAL) {
        const double* a = T.Pq + k * Nq;
        const double* x = sR + (e * 4 + f) * Nq;
        s = 0.0;
#pragma unroll
        for (int q = 0; q < Nq; ++q) s += a[q] * x[q];
      } else {
        s = sR[(e * 4 + f) * Nq + k];
      }
      rhs[(int64_t)f * M.K * Np + e0 * Np + r] = s;
    }
  }
}

template <int N1>
constexpr size_t rhs_lds_bytes() {
  constexpr int Nq = N1 * N1, Nfq = 4 * N1, Nh = Nq + Nfq, E = EPB<N1>::v, PMAX = N1 * N1 * (N1 + 3);
  size_t inv = (size_t)E * (8 * Nq + 6 * Nh + 4 * Nfq + 4 * Nh + 4 * PMAX);
  size_t visc = (size_t)E * (8 * Nq + 6 * Nh + 4 * Nfq + 4 * Nh + 9 * Nfq + 6 * Nq);
  return sizeof(double) * (inv > visc ? inv : visc);
}

// ---------------------------------------------------------------------------------------------
// small utility kernels
// ---------------------------------------------------------------------------------------------
// halo pack: dst[s][c] = src[list[s]][c]
__global__ void k_pack(const double* __restrict__ src, int ncomp, const int32_t* __restrict__ list, int64_t n,
                       double* __restrict__ dst) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n * ncomp) {
    const int64_t s = i / ncomp;
    const int c = (int)(i - s * ncomp);
    dst[i] = src[(int64_t)list[s] * ncomp + c];
  }
}

// LSRK stage (src/CommonUtils.jl:29-49, loop euler_quad.jl:204-205)
__global__ void k_lsrk(double* __restrict__ Q, double* __restrict__ resQ, const double* __restrict__ rhs, double a,
                       double b, double dt, int64_t n) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const double r = __builtin_fma(a, resQ[i], dt * rhs[i]);   // same rounding sequence as the fused form in kt_rhs
    resQ[i] = r;
    Q[i] = __builtin_fma(b, r, Q[i]);
  }
}

struct StagePtrs { const double* k[8]; double c[8]; };

__global__ void k_axpy_stages(double* __restrict__ y, const double* __restrict__ x0, StagePtrs sp, int ns, double dt,
                              int64_t n) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    double k = 0.0;
    for (int s = 0; s < ns; ++s) k = __builtin_fma(sp.c[s], sp.k[s][i], k);   // (explicit: the fused form in kt3_rhs repeats this chain)
    y[i] = __builtin_fma(dt, k, x0[i]);
  }
}

__device__ __forceinline__ double block_sum(double v) {
  __shared__ double red[ESDG_TPB / 64];
  for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  __syncthreads();
  if (lane == 0) red[w] = v;
  __syncthreads();
  double s = 0.0;
  if (threadIdx.x == 0)
    for (int i = 0; i < ESDG_TPB / 64; ++i) s += red[i];
  return s;
}

// The Hairer norm's numerator in ONE summation order, whoever forms the terms (round 5; the order used to follow the launch that
// produced the partials).  Node i of the `nodes` per field carries t_i = sum over its nfld fields, in field order and as one fma
// chain, of (|e| / (tol (1 + |x|)))^2; consecutive runs of ESDG_ERR_CHUNK nodes are summed by one workgroup -- thread t adds
// nodes t, t + 256, ... of its run in that order, then the 256 thread sums meet in a fixed tree -- and k_sum adds the runs' sums
// in its own fixed order.  The order is a function of (nodes, nfld) alone: the fused attempt (kt3_rhs / kh_rhs_l store t_i,
// k_chunk_sum adds them), the attempt from the building blocks (k_dopri_err forms t_i and adds them in flight) and a sharded
// context's pieces give the same bits.
__device__ __forceinline__ double chunk_tree(double a) {
  __shared__ double red[ESDG_TPB];
  red[threadIdx.x] = a;
  __syncthreads();
  for (int w = ESDG_TPB / 2; w > 0; w >>= 1) {
    if ((int)threadIdx.x < w) red[threadIdx.x] += red[threadIdx.x + w];
    __syncthreads();
  }
  return red[0];
}

__global__ __launch_bounds__(ESDG_TPB) void k_dopri_err(const double* __restrict__ Q, StagePtrs sp, int ns, double tol,
                                                        int64_t nodes, int nfld, double* __restrict__ chunk) {
  double acc = 0.0;
  const int64_t base = (int64_t)blockIdx.x * ESDG_ERR_CHUNK;
  for (int j = 0; j < ESDG_ERR_CHUNK / ESDG_TPB; ++j) {
    const int64_t i = base + (int64_t)j * ESDG_TPB + threadIdx.x;
    if (i < nodes) {
      double t = 0.0;
      for (int f = 0; f < nfld; ++f) {
        const int64_t idx = (int64_t)f * nodes + i;
        double e = 0.0;
        for (int s = 0; s < ns; ++s) e = __builtin_fma(sp.c[s], sp.k[s][idx], e);
        const double sc = fabs(e) / (tol * (1 + fabs(Q[idx])));
        t = __builtin_fma(sc, sc, t);   // (the chain the fused kernels run per node: kt3_rhs / kh_rhs_l STG epilogues)
      }
      asm volatile("" : "+v"(t));       // (a finished term, as the fused kernels store it: no contraction with the running sum)
      acc += t;
    }
  }
  const double s = chunk_tree(acc);
  if (threadIdx.x == 0) chunk[blockIdx.x] = s;
}

__global__ __launch_bounds__(ESDG_TPB) void k_chunk_sum(const double* __restrict__ x, int64_t n, double* __restrict__ chunk) {
  double acc = 0.0;
  const int64_t base = (int64_t)blockIdx.x * ESDG_ERR_CHUNK;
  for (int j = 0; j < ESDG_ERR_CHUNK / ESDG_TPB; ++j) {
    const int64_t i = base + (int64_t)j * ESDG_TPB + threadIdx.x;
    if (i < n) acc += x[i];
  }
  const double s = chunk_tree(acc);
  if (threadIdx.x == 0) chunk[blockIdx.x] = s;
}

// rhstest = sum(wJq .* v(Uq) .* (Vq*rhs))  (euler_quad.jl:186-191; cavity_optimized.jl:958-966 with
// Vq*Pq = I on quads).  One partial per block.
template <int N1, bool MODAL>
__global__ __launch_bounds__(ESDG_TPB) void k_rhstest(Tables T, MeshDev M, const double* __restrict__ Q,
                                                      const double* __restrict__ rhs, double* __restrict__ partial) {
  constexpr int Nq = N1 * N1, Np = Nq;
  double acc = 0.0;
  const int64_t total = M.K * Nq;
  for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
       idx += (int64_t)gridDim.x * blockDim.x) {
    const int64_t e = idx / Nq;
    const int q = (int)(idx - e * Nq);
    double U[4], R[4];
    for (int f = 0; f < 4; ++f) {
      const double* x = Q + (int64_t)f * M.K * Np + e * Np;
      const double* r = rhs + (int64_t)f * M.K * Np + e * Np;
      if (MODAL) {
        double s = 0.0, t = 0.0;
        for (int k = 0; k < Np; ++k) {
          s += T.Vq[q * Np + k] * x[k];
          t += T.Vq[q * Np + k] * r[k];
        }
        U[f] = s;
        R[f] = t;
      } else {
        U[f] = x[q];
        R[f] = r[q];
      }
    }
    double V[4], lr;
    v_of_u<MODAL>(U, V, lr);
    const double w = M.wJq[e * Nq + q];
    acc += w * (V[0] * R[0] + V[1] * R[1] + V[2] * R[2] + V[3] * R[3]);
  }
  const double s = block_sum(acc);
  if (threadIdx.x == 0) partial[blockIdx.x] = s;
}

// ---------------------------------------------------------------------------------------------
// launchers
// ---------------------------------------------------------------------------------------------
#define ESDG_DISPATCH_N1(N1v, BODY)                    \
  switch (N1v) {                                       \
    case 2: { constexpr int N1 = 2; BODY; } break;     \
    case 3: { constexpr int N1 = 3; BODY; } break;     \
    case 4: { constexpr int N1 = 4; BODY; } break;     \
    case 5: { constexpr int N1 = 5; BODY; } break;     \
    case 6: { constexpr int N1 = 6; BODY; } break;     \
    case 7: { constexpr int N1 = 7; BODY; } break;     \
    case 8: { constexpr int N1 = 8; BODY; } break;     \
    default: return (int)hipErrorInvalidValue;         \
  }

static inline int nblocks_for(int64_t K, int E) { return (int)((K + E - 1) / E); }

int launch_project(const Tables& T, const MeshDev& M, const Phys& ph, const double* Q, double* A_U, double* A_v,
                   hipStream_t s) {
  if (M.K == 0) return 0;
  const bool modal = ph.formulation != 0, visc = ph.formulation == 1;
  ESDG_DISPATCH_N1(T.N1, {
    const int nb = nblocks_for(M.K, EPB<N1>::v);
    if (!modal)
      hipLaunchKernelGGL((k_project<N1, false, false>), dim3(nb), dim3(ESDG_TPB), 0, s, T, M, ph, Q, A_U, A_v);
    else if (visc)
      hipLaunchKernelGGL((k_project<N1, true, true>), dim3(nb), dim3(ESDG_TPB), 0, s, T, M, ph, Q, A_U, A_v);
    else
      hipLaunchKernelGGL((k_project<N1, true, false>), dim3(nb), dim3(ESDG_TPB), 0, s, T, M, ph, Q, A_U, A_v);
  });
  return (int)hipGetLastError();
}

int launch_sigma(const Tables& T, const MeshDev& M, const Phys& ph, const double* Q, const double* A_v, double* B,
                 hipStream_t s) {
  if (M.K == 0) return 0;
  ESDG_DISPATCH_N1(T.N1, {
    const int nb = nblocks_for(M.K, EPB<N1>::v);
    hipLaunchKernelGGL((k_sigma<N1>), dim3(nb), dim3(ESDG_TPB), 0, s, T, M, ph, Q, A_v, B);
  });
  return (int)hipGetLastError();
}

template <typename KernelT>
static int set_dyn_lds(KernelT k, size_t bytes) {
  return (int)hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize,
                                  (int)bytes);
}

int launch_rhs(const Tables& T, const MeshDev& M, const Phys& ph, const double* Q, const double* A_U,
               const double* A_v, const double* B, double* rhs, hipStream_t s) {
  if (M.K == 0) return 0;
  const bool modal = ph.formulation != 0, visc = ph.formulation == 1;
  ESDG_DISPATCH_N1(T.N1, {
    const int nb = nblocks_for(M.K, EPB<N1>::v);
    const size_t lds = rhs_lds_bytes<N1>();
    if (!modal) {
      static int once = set_dyn_lds(k_rhs<N1, false, false>, lds);
      (void)once;
      hipLaunchKernelGGL((k_rhs<N1, false, false>), dim3(nb), dim3(ESDG_TPB), lds, s, T, M, ph, Q, A_U, A_v, B, rhs);
    } else if (visc) {
      static int once = set_dyn_lds(k_rhs<N1, true, true>, lds);
      (void)once;
      hipLaunchKernelGGL((k_rhs<N1, true, true>), dim3(nb), dim3(ESDG_TPB), lds, s, T, M, ph, Q, A_U, A_v, B, rhs);
    } else {
      static int once = set_dyn_lds(k_rhs<N1, true, false>, lds);
      (void)once;
      hipLaunchKernelGGL((k_rhs<N1, true, false>), dim3(nb), dim3(ESDG_TPB), lds, s, T, M, ph, Q, A_U, A_v, B, rhs);
    }
  });
  return (int)hipGetLastError();
}

int launch_pack(const double* src, int ncomp, const int32_t* list, int64_t n, double* dst, hipStream_t s) {
  if (n == 0) return 0;
  const int64_t tot = n * ncomp;
  hipLaunchKernelGGL(k_pack, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, s, src, ncomp, list, n, dst);
  return (int)hipGetLastError();
}

int launch_rhstest(const Tables& T, const MeshDev& M, const Phys& ph, const double* Q, const double* rhs,
                   double* partial, int nblocks, hipStream_t s) {
  const bool modal = ph.formulation != 0;
  if (T.N1 == 9 || T.N1 == 10) {   // (degrees the tensor kernels of rounds 2-4 serve beyond this file's own range)
    if (T.N1 == 9) {
      if (modal) hipLaunchKernelGGL((k_rhstest<9, true>), dim3(nblocks), dim3(ESDG_TPB), 0, s, T, M, Q, rhs, partial);
      else hipLaunchKernelGGL((k_rhstest<9, false>), dim3(nblocks), dim3(ESDG_TPB), 0, s, T, M, Q, rhs, partial);
    } else {
      if (modal) hipLaunchKernelGGL((k_rhstest<10, true>), dim3(nblocks), dim3(ESDG_TPB), 0, s, T, M, Q, rhs, partial);
      else hipLaunchKernelGGL((k_rhstest<10, false>), dim3(nblocks), dim3(ESDG_TPB), 0, s, T, M, Q, rhs, partial);
    }
    return (int)hipGetLastError();
  }
  ESDG_DISPATCH_N1(T.N1, {
    if (modal)
      hipLaunchKernelGGL((k_rhstest<N1, true>), dim3(nblocks), dim3(ESDG_TPB), 0, s, T, M, Q, rhs, partial);
    else
      hipLaunchKernelGGL((k_rhstest<N1, false>), dim3(nblocks), dim3(ESDG_TPB), 0, s, T, M, Q, rhs, partial);
  });
  return (int)hipGetLastError();
}

static inline int stream_blocks(int64_t n) {
  int64_t b = (n + 255) / 256;
  return (int)(b > 4096 ? 4096 : (b < 1 ? 1 : b));
}

int launch_lsrk(double* Q, double* resQ, const double* rhs, double a, double b, double dt, int64_t n, hipStream_t s) {
  hipLaunchKernelGGL(k_lsrk, dim3(stream_blocks(n)), dim3(256), 0, s, Q, resQ, rhs, a, b, dt, n);
  return (int)hipGetLastError();
}

int launch_axpy_stages(double* y, const double* x0, const double* const* k, const double* coef, int ns, double dt,
                       int64_t n, hipStream_t s) {
  if (ns > 8) return (int)hipErrorInvalidValue;
  StagePtrs sp;
  for (int i = 0; i < ns; ++i) { sp.k[i] = k[i]; sp.c[i] = coef[i]; }
  hipLaunchKernelGGL(k_axpy_stages, dim3(stream_blocks(n)), dim3(256), 0, s, y, x0, sp, ns, dt, n);
  return (int)hipGetLastError();
}

// admissibility of a state: per-block minima of rho and of p = (gamma-1)(E - |rhoU|^2/(2 rho)) over the n nodes of a
// stacked state [nfld][n] (nfld = 4: 2D, 5: hex).  The reference throws a DomainError from log/sqrt instead.
__global__ void k_min_rho_p(const double* __restrict__ Q, int nfld, int64_t n, double* __restrict__ partial) {
  __shared__ double r0[256], r1[256];
  double mr = 1e300, mp = 1e300;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const double rho = Q[i];
    double m2 = 0.0;
    for (int c = 1; c < nfld - 1; ++c) m2 += Q[(int64_t)c * n + i] * Q[(int64_t)c * n + i];
    const double p = 0.4 * (Q[(int64_t)(nfld - 1) * n + i] - .5 * m2 / rho);
    mr = fmin(mr, rho == rho ? rho : -1e300);      // NaN counts as inadmissible
    mp = fmin(mp, p == p ? p : -1e300);
  }
  r0[threadIdx.x] = mr; r1[threadIdx.x] = mp;
  __syncthreads();
  for (int w = 128; w > 0; w >>= 1) {
    if ((int)threadIdx.x < w) { r0[threadIdx.x] = fmin(r0[threadIdx.x], r0[threadIdx.x + w]); r1[threadIdx.x] = fmin(r1[threadIdx.x], r1[threadIdx.x + w]); }
    __syncthreads();
  }
  if (threadIdx.x == 0) { partial[2 * blockIdx.x] = r0[0]; partial[2 * blockIdx.x + 1] = r1[0]; }
}

int launch_min_rho_p(const double* Q, int nfld, int64_t n, double* partial, int nblocks, hipStream_t s) {
  hipLaunchKernelGGL(k_min_rho_p, dim3(nblocks), dim3(256), 0, s, Q, nfld, n, partial);
  return (int)hipGetLastError();
}

__global__ __launch_bounds__(1024) void k_sum(const double* __restrict__ x, int64_t n, double* __restrict__ out) {
  __shared__ double red[1024];
  double a = 0.0;
  for (int64_t i = threadIdx.x; i < n; i += 1024) a += x[i];
  red[threadIdx.x] = a;
  __syncthreads();
  for (int w = 512; w > 0; w >>= 1) {
    if ((int)threadIdx.x < w) red[threadIdx.x] += red[threadIdx.x + w];
    __syncthreads();
  }
  if (threadIdx.x == 0) out[0] = red[0];
}

int launch_sum(const double* x, int64_t n, double* out, hipStream_t s) {
  hipLaunchKernelGGL(k_sum, dim3(1), dim3(1024), 0, s, x, n, out);
  return (int)hipGetLastError();
}

int launch_dopri_err(const double* Q, const double* const* k, const double* coefE, int ns, double tol, int64_t nodes, int nfld,
                     double* chunk, hipStream_t s) {
  if (ns > 8 || nfld < 1) return (int)hipErrorInvalidValue;
  StagePtrs sp;
  for (int i = 0; i < ns; ++i) { sp.k[i] = k[i]; sp.c[i] = coefE[i]; }
  hipLaunchKernelGGL(k_dopri_err, dim3((unsigned)err_chunks(nodes)), dim3(ESDG_TPB), 0, s, Q, sp, ns, tol, nodes, nfld, chunk);
  return (int)hipGetLastError();
}

int launch_chunk_sum(const double* x, int64_t n, double* chunk, hipStream_t s) {
  hipLaunchKernelGGL(k_chunk_sum, dim3((unsigned)err_chunks(n)), dim3(ESDG_TPB), 0, s, x, n, chunk);
  return (int)hipGetLastError();
}

}  // namespace esdg
