// Error functionals the drivers print after (or during) a run, evaluated on the device so that a convergence study
// never moves the state to the host:
//   * L2 error against an exact solution with a finer quadrature (examples/dg2D_euler_quad.jl:214-233),
//   * the exact travelling viscous shock of Becker by bisection (examples/CompressibleNS/dg2D_CNS_modalESDG.jl:545-579)
//     and the nodal L1 / Linf errors of that driver (:745-771),
//   * the boundary-velocity error of the lid-driven cavity (examples/CompressibleNS/dg2D_CNS_convergence_test.jl:1055-1080).
// None of this is on the RHS hot path: plain grid-stride kernels, one work item per evaluation node, block partials
// summed on the host in block order (deterministic for a fixed launch shape).
#include <hip/hip_runtime.h>

#include "esdg_dev.hpp"
#include "esdg_devmath.hpp"

namespace esdg {

namespace {

constexpr int ERR_TPB = 256;
constexpr double GAM = 1.4;

// vortex(x,y,t) of examples/EntropyStableEuler/EntropyStableEuler.jl:21-35 (x0 = 5, y0 = 0, beta = 5) as conservative
// variables (primitive_to_conservative, euler_variables.jl:15-24)
__device__ inline void exact_vortex(double x, double y, double t, double* U) {
  const double x0 = 5.0, y0 = 0.0, beta = 5.0, pi = 3.14159265358979323846;
  const double r2 = (x - x0 - t) * (x - x0 - t) + (y - y0) * (y - y0);
  const double be = beta * exp(1.0 - r2);
  const double u = 1.0 - be * (y - y0) / (2.0 * pi);
  const double v = be * (x - x0 - t) / (2.0 * pi);
  const double rho = pow(1.0 - (1.0 / (8.0 * GAM * pi * pi)) * (GAM - 1.0) / 2.0 * be * be, 1.0 / (GAM - 1.0));
  const double p = pow(rho, GAM);
  U[0] = rho; U[1] = rho * u; U[2] = rho * v;
  U[3] = p / (GAM - 1.0) + .5 * rho * (u * u + v * v);
}

// par = (v_0, v_1, v_01, m_0, L_k = kappa/m_0/cv, v_inf): dg2D_CNS_modalESDG.jl:545-579, max_iter = 100, tol = 1e-14
__device__ inline void exact_becker(double x, double t, const double* par, double* U) {
  const double v0 = par[0], v1 = par[1], v01 = par[2], m0 = par[3], Lk = par[4], vinf = par[5];
  const double xs = x - vinf * t;
  const double c = 2.0 * Lk / (GAM + 1.0), a0 = v0 / (v0 - v1), a1 = v1 / (v0 - v1);
  auto f = [&](double v) { return -xs + c * (a0 * log((v0 - v) / (v0 - v01)) - a1 * log((v - v1) / (v01 - v1))); };
  auto sgn = [](double z) { return z > 0.0 ? 1 : (z < 0.0 ? -1 : 0); };
  double vL = v1, vR = v0, vn = .5 * (vL + vR);
  for (int it = 0; it < 100; ++it) {
    vn = .5 * (vL + vR);
    const double fn = f(vn);
    if (fabs(fn) < 1e-14) break;
    if (sgn(f(vL)) == sgn(fn)) vL = vn; else vR = vn;
  }
  const double u = vn, rho = m0 / u;
  const double e = 1.0 / (2.0 * GAM) * ((GAM + 1.0) / (GAM - 1.0) * v01 * v01 - u * u);
  U[0] = rho; U[1] = rho * (vinf + u); U[2] = 0.0;
  U[3] = rho * (e + .5 * (vinf + u) * (vinf + u));
}

__device__ inline void exact_state(int kind, double x, double y, double t, const double* par, double* U) {
  if (kind == 0) exact_vortex(x, y, t, U); else exact_becker(x, t, par, U);
}

struct BeckerPar { double v[6]; };

template <int NV>
__device__ inline void block_sum(double* val, double* partial) {
  __shared__ double red[NV][ERR_TPB];
#pragma unroll
  for (int c = 0; c < NV; ++c) red[c][threadIdx.x] = val[c];
  __syncthreads();
  for (int w = ERR_TPB / 2; w > 0; w >>= 1) {
    if ((int)threadIdx.x < w)
#pragma unroll
      for (int c = 0; c < NV; ++c) red[c][threadIdx.x] += red[c][threadIdx.x + w];
    __syncthreads();
  }
  if (threadIdx.x == 0)
#pragma unroll
    for (int c = 0; c < NV; ++c) partial[(size_t)blockIdx.x * NV + c] = red[c][0];
}

// sum over elements and error-quadrature nodes of wq2*(Vq2*J) * (Vq2*Q_f - Qex_f(Vq2*x, Vq2*y, t))^2, per field
__global__ void k_err_l2(ErrDev E, const double* __restrict__ Q, int kind, BeckerPar par, double t,
                         double* __restrict__ partial) {
  double acc[4] = {0, 0, 0, 0};
  const int64_t n = E.K * E.Nq2, KN = E.K * E.Np;
  for (int64_t i = (int64_t)blockIdx.x * ERR_TPB + threadIdx.x; i < n; i += (int64_t)gridDim.x * ERR_TPB) {
    const int64_t e = i / E.Nq2;
    const int q = (int)(i - e * E.Nq2);
    const double* row = E.Vq2 + (size_t)q * E.Np;
    const int64_t b = e * E.Np;
    double x = 0, y = 0, J = 0, U[4] = {0, 0, 0, 0};
    for (int j = 0; j < E.Np; ++j) {
      const double w = row[j];
      x += w * E.x[b + j]; y += w * E.y[b + j]; J += w * E.J[b + j];
#pragma unroll
      for (int f = 0; f < 4; ++f) U[f] += w * Q[f * KN + b + j];
    }
    double X[4];
    exact_state(kind, x, y, t, par.v, X);
    const double wJ = E.wq2[q] * J;
#pragma unroll
    for (int f = 0; f < 4; ++f) acc[f] += wJ * (U[f] - X[f]) * (U[f] - X[f]);
  }
  block_sum<4>(acc, partial);
}

// nodal errors of dg2D_CNS_modalESDG.jl:745-771 for the fields rho, rho*u, E: per block
// (sum|d|, sum|q|) x 3 and (max|d|, max|q|) x 3
__global__ void k_err_nodal(ErrDev E, const double* __restrict__ Q, int kind, BeckerPar par, double t,
                            double* __restrict__ partial) {
  __shared__ double red[12][ERR_TPB];
  double s[6] = {0, 0, 0, 0, 0, 0}, m[6] = {0, 0, 0, 0, 0, 0};
  const int64_t KN = E.K * E.Np;
  for (int64_t i = (int64_t)blockIdx.x * ERR_TPB + threadIdx.x; i < KN; i += (int64_t)gridDim.x * ERR_TPB) {
    double X[4];
    exact_state(kind, E.x[i], E.y[i], t, par.v, X);
    const int fl[3] = {0, 1, 3};
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const double q = Q[fl[c] * KN + i], d = fabs(X[fl[c]] - q), a = fabs(q);
      s[2 * c] += d; s[2 * c + 1] += a;
      m[2 * c] = fmax(m[2 * c], d); m[2 * c + 1] = fmax(m[2 * c + 1], a);
    }
  }
#pragma unroll
  for (int c = 0; c < 6; ++c) { red[c][threadIdx.x] = s[c]; red[6 + c][threadIdx.x] = m[c]; }
  __syncthreads();
  for (int w = ERR_TPB / 2; w > 0; w >>= 1) {
    if ((int)threadIdx.x < w) {
#pragma unroll
      for (int c = 0; c < 6; ++c) {
        red[c][threadIdx.x] += red[c][threadIdx.x + w];
        red[6 + c][threadIdx.x] = fmax(red[6 + c][threadIdx.x], red[6 + c][threadIdx.x + w]);
      }
    }
    __syncthreads();
  }
  if (threadIdx.x == 0)
#pragma unroll
    for (int c = 0; c < 12; ++c) partial[(size_t)blockIdx.x * 12 + c] = red[c][0];
}

// the three sums of dg2D_CNS_convergence_test.jl:1075-1077 over wall/lid face nodes, u = Vf*(Q[2]./Q[1], Q[3]./Q[1]):
// Jf*wf*u_2^2 (all boundary nodes), Jf*wf*u_1^2 (walls), Jf*wf*(u_1 - vlid)^2 (lid); bc: 1 wall, 2 lid
__global__ void k_err_boundary(ErrDev E, const uint8_t* __restrict__ bc, const double* __restrict__ vlid,
                               const double* __restrict__ Q, double Jf, double* __restrict__ partial) {
  double acc[3] = {0, 0, 0};
  const int64_t n = E.K * E.Nfq, KN = E.K * E.Np;
  for (int64_t i = (int64_t)blockIdx.x * ERR_TPB + threadIdx.x; i < n; i += (int64_t)gridDim.x * ERR_TPB) {
    const int b = bc[i];
    if (b != 1 && b != 2) continue;
    const int64_t e = i / E.Nfq;
    const int fq = (int)(i - e * E.Nfq);
    const double* row = E.Vf + (size_t)fq * E.Np;
    double u1 = 0, u2 = 0;
    for (int j = 0; j < E.Np; ++j) {
      const double rho = Q[e * E.Np + j];
      u1 += row[j] * (Q[KN + e * E.Np + j] / rho);
      u2 += row[j] * (Q[2 * KN + e * E.Np + j] / rho);
    }
    const double w = Jf * E.wf[fq];
    acc[0] += w * u2 * u2;
    if (b == 2) {
      const double ul = vlid ? vlid[i] : 1.0;
      acc[2] += w * (u1 - ul) * (u1 - ul);
    } else {
      acc[1] += w * u1 * u1;
    }
  }
  block_sum<3>(acc, partial);
}

BeckerPar pack_par(const double* par) {
  BeckerPar p{};
  if (par)
    for (int i = 0; i < 6; ++i) p.v[i] = par[i];
  return p;
}

}  // namespace

int launch_err_l2(const ErrDev& E, const double* Q, int kind, const double* par, double t, double* partial, int nblocks,
                  hipStream_t s) {
  hipLaunchKernelGGL(k_err_l2, dim3(nblocks), dim3(ERR_TPB), 0, s, E, Q, kind, pack_par(par), t, partial);
  return (int)hipGetLastError();
}

int launch_err_nodal(const ErrDev& E, const double* Q, int kind, const double* par, double t, double* partial, int nblocks,
                     hipStream_t s) {
  hipLaunchKernelGGL(k_err_nodal, dim3(nblocks), dim3(ERR_TPB), 0, s, E, Q, kind, pack_par(par), t, partial);
  return (int)hipGetLastError();
}

int launch_err_boundary(const ErrDev& E, const uint8_t* bc, const double* vlid, const double* Q, double Jf, double* partial,
                        int nblocks, hipStream_t s) {
  hipLaunchKernelGGL(k_err_boundary, dim3(nblocks), dim3(ERR_TPB), 0, s, E, bc, vlid, Q, Jf, partial);
  return (int)hipGetLastError();
}

// the kernels' logarithm on an array (esdg_debug_log; tools/logtest.py compares it with numpy.log)
__global__ void k_log_test(const double* __restrict__ x, double* __restrict__ y, int64_t n) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) y[i] = devmath::log_pos(x[i]);
}

int launch_log_test(const double* x, double* y, int64_t n, hipStream_t s) {
  hipLaunchKernelGGL(k_log_test, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, x, y, n);
  return (int)hipGetLastError();
}

}  // namespace esdg
