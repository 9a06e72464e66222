// esdg_kernels_fast.hip -- tensor-line kernels for gfx950 (MI355X), used when the driver's SBP
// operators have the tensor-product Gauss structure (always true for init_reference_quad with a
// Gauss rule; checked in esdg_api.hip, otherwise the generic kernels of esdg_kernels.hip run).
//
// Same algorithm and face-trace protocol as esdg_kernels.hip (see the reference citations there);
// what changes is the mapping to the machine:
//   * lane <-> (element, volume node): a block of T threads owns E = T/Nq elements; the state load and
//     the rhs store are one coalesced 8-byte access per lane and field, no staging pass;
//   * Vq / Pq are applied by sum factorisation (2 x N1 FMAs per node instead of N1^2) through LDS;
//   * flux differencing walks the 2*N1 tensor lines: per direction a lane evaluates N1/2 forward
//     volume pairs (circulant schedule, every unordered pair once) + its 2 face pairs with its own
//     node in registers, contributions to the partner nodes go through one LDS exchange record per
//     lane; 200 EC fluxes per element at N=4, none duplicated;
//   * the EC flux uses ONE refined v_rcp_f64 for its three quotients and no data-dependent branch;
//   * everything pointwise works on (rho,u,v,beta,log rho,log beta): entropy variables follow as
//     v = (g - s - (g-1) beta |u|^2, 2(g-1) beta u, 2(g-1) beta v, -2(g-1) beta), s = -(g-1) log rho - log beta - log 2,
//     so a volume node costs 2 logs; face traces carry the same 6 numbers + lam + E (64-byte records),
//     so the consumer kernels do no transcendental work on traces.
#include "esdg_dev.hpp"

namespace esdg {

// threads per block / elements per block, per N1 = N+1:  E*max(Nq, Nfq) <= T
template <int N1> struct FCfg { static constexpr int T = 64, E = 1; };
template <> struct FCfg<2> { static constexpr int T = 64, E = 8; };
template <> struct FCfg<3> { static constexpr int T = 64, E = 5; };
template <> struct FCfg<4> { static constexpr int T = 64, E = 4; };
template <> struct FCfg<5> { static constexpr int T = 64, E = 2; };
template <> struct FCfg<6> { static constexpr int T = 128, E = 3; };
template <> struct FCfg<7> { static constexpr int T = 128, E = 2; };
template <> struct FCfg<8> { static constexpr int T = 64, E = 1; };

namespace fastdev {

template <bool MODAL> struct Gas {
  static constexpr double GM1 = MODAL ? 0.4 : (1.4 - 1);   // literal 0.4 in the CNS drivers, gamma-1 in the Euler one
};

__device__ __forceinline__ double rcp_refined(double x) {
  double r = __builtin_amdgcn_rcp(x);
  double e = __builtin_fma(-x, r, 1.0);
  r = __builtin_fma(r, e, r);
  e = __builtin_fma(-x, r, 1.0);
  return __builtin_fma(r, e, r);
}

// Entropy-conservative flux (euler_fluxes.jl:23-48 with logmean.jl:14-28), q = (rho,u,v,beta,lrho,lbeta).
// The three quotients (rho log-mean, 1/beta log-mean, pa) share one reciprocal; the reference's
// |f|<1e-4 series branch is kept (selected, not branched).  1/P(v) of the series is expanded to
// 1 + .2v + .0912v^2 (v < 1e-8: truncation < 1e-24).
template <bool MODAL>
__device__ __forceinline__ void ec_flux_fast(const double* qL, const double* qR, double* Fx, double* Fy) {
  constexpr double GM1 = Gas<MODAL>::GM1;
  const double dr = qR[0] - qL[0], ravg = .5 * (qR[0] + qL[0]);
  const double db = qR[3] - qL[3], bavg = .5 * (qR[3] + qL[3]);
  const double A = qL[4] - qR[4], B = qL[5] - qR[5];
  const bool ser_r = fabs(dr) < 1e-4 * ravg, ser_b = fabs(db) < 1e-4 * bavg;
  const double yr = ser_r ? ravg : A;
  const double yb = ser_b ? bavg : db;
  const double yp = qL[3] + qR[3];
  const double ybp = yb * yp;
  const double R = rcp_refined(yr * ybp);
  const double ir = R * ybp;          // 1/yr
  const double ryr = R * yr;
  const double ib = ryr * yp;         // 1/yb
  const double ip = ryr * yb;         // 1/(betaL+betaR)
  const double fr = dr * ir, vr = fr * fr;
  const double rholog = ser_r ? ravg * (1 + vr * (-.2 - vr * (.0512 - vr * 0.026038857142857))) : -fr;
  const double fb = db * ib, vb = fb * fb;
  const double ibetalog = ser_b ? ib * (1 + vb * (.2 + vb * .0912)) : -(B * ib);
  const double uavg = .5 * (qL[1] + qR[1]), vavg = .5 * (qL[2] + qR[2]);
  const double unorm = qL[1] * qR[1] + qL[2] * qR[2];
  const double pa = ravg * ip;
  const double f4aux = rholog * ibetalog * (1.0 / (2 * GM1)) + pa + .5 * rholog * unorm;
  Fx[0] = rholog * uavg;
  Fx[1] = Fx[0] * uavg + pa;
  Fx[2] = Fx[0] * vavg;
  Fx[3] = f4aux * uavg;
  Fy[0] = rholog * vavg;
  Fy[1] = Fx[2];
  Fy[2] = Fy[0] * vavg + pa;
  Fy[3] = f4aux * vavg;
}

// conservative -> (rho,u,v,beta,log rho,log beta)   (betafun euler_variables.jl:30-48 / cavity :484)
template <bool MODAL>
__device__ __forceinline__ void prim_logs(const double* U, double* q) {
  constexpr double GM1 = Gas<MODAL>::GM1;
  const double ir = 1.0 / U[0];
  q[0] = U[0];
  q[1] = U[1] * ir;
  q[2] = U[2] * ir;
  const double rhoe = U[3] - .5 * (U[1] * U[1] + U[2] * U[2]) * ir;
  q[3] = U[0] / (2 * GM1 * rhoe);
  q[4] = log(U[0]);
  q[5] = log(q[3]);
}

// entropy variables from primitives + logs (identities of euler_variables.jl:79-92)
template <bool MODAL>
__device__ __forceinline__ void v_of_prim(const double* q, double* V) {
  constexpr double GM1 = Gas<MODAL>::GM1;
  const double s = -GM1 * q[4] - q[5] - 0.6931471805599453;
  const double b2 = 2 * GM1 * q[3];
  V[0] = 1.4 - s - .5 * b2 * (q[1] * q[1] + q[2] * q[2]);
  V[1] = b2 * q[1];
  V[2] = b2 * q[2];
  V[3] = -b2;
}

// conservative variables of entropy variables (euler_variables.jl:95-120 / cavity :473-478), no pow
template <bool MODAL>
__device__ __forceinline__ void u_of_v(const double* V, double* U) {
  constexpr double GM1 = Gas<MODAL>::GM1;
  const double vUnorm = V[1] * V[1] + V[2] * V[2];
  const double h = vUnorm / (2 * V[3]);
  const double s = 1.4 - V[0] + h;
  const double rhoeV = exp((log(GM1) - 1.4 * log(-V[3]) - s) / GM1);
  U[0] = rhoeV * (-V[3]);
  U[1] = rhoeV * V[1];
  U[2] = rhoeV * V[2];
  U[3] = rhoeV * (1 - h);
}

// wavespeed (euler_variables.jl:7-10, sqrt(|u_n|) quirk Q1 / cavity :507)
template <bool MODAL>
__device__ __forceinline__ double lf_lambda(const double* U, double nxJ, double nyJ, double sJ) {
  const double rhoUn = (U[1] * nxJ + U[2] * nyJ) / sJ;
  if (MODAL) return fabs(sqrt(fabs(rhoUn / U[0])) + sqrt(1.4 * 0.4 * (U[3] - .5 * rhoUn * rhoUn / U[0]) / U[0]));
  const double p = (1.4 - 1) * (U[3] - .5 * (rhoUn * rhoUn) / U[0]);
  return fabs(sqrt(fabs(rhoUn / U[0])) + sqrt(1.4 * p / U[0]));
}

// viscous_matrices! + sigma rows 2..4 (cavity :613-645, 786-801); lam already sign-flipped (quirk Q4)
__device__ __forceinline__ void viscous_stress(const double* v, const double* tx, const double* ty, double lam,
                                               double mu, double Pr, double* sx, double* sy) {
  const double v2 = v[0], v3 = v[1], v4 = v[2];
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
  sx[0] = Kxx22 * tx[0] + Kxx24 * tx[2] + Kxy23 * ty[1] + Kxy24 * ty[2];
  sx[1] = Kxx33 * tx[1] + Kxx34 * tx[2] + Kxy32 * ty[0] + Kxy34 * ty[2];
  sx[2] = Kxx24 * tx[0] + Kxx34 * tx[1] + Kxx44 * tx[2] + Kxy42 * ty[0] + Kxy43 * ty[1] + Kxy44 * ty[2];
  sy[0] = Kxy32 * tx[1] + Kxy42 * tx[2] + Kyy22 * ty[0] + Kyy24 * ty[2];
  sy[1] = Kxy23 * tx[0] + Kxy43 * tx[2] + Kyy33 * ty[1] + Kyy34 * ty[2];
  sy[2] = Kxy24 * tx[0] + Kxy34 * tx[1] + Kxy44 * tx[2] + Kyy24 * ty[0] + Kyy34 * ty[1] + Kyy44 * ty[2];
}

// ---- state load (+ Vq by sum factorisation) -------------------------------------------------
// Returns the conservative state at this lane's Gauss node.  sA, sB: LDS scratch [E][4][Nq] each.
template <int N1>
__device__ __forceinline__ void issue_state_loads(const double* __restrict__ Q, int64_t K, int64_t e0, bool active,
                                                  double* x) {
  constexpr int Nq = N1 * N1;
  x[0] = 1.0; x[1] = 0.0; x[2] = 0.0; x[3] = 1.0;
  if (active) {
#pragma unroll
    for (int f = 0; f < 4; ++f) x[f] = Q[(int64_t)f * K * Nq + e0 * Nq + threadIdx.x];
  }
}

template <int N1, bool MODAL>
__device__ __forceinline__ void state_at_quad(const FastTables& F, bool active, int ev, int q, double* sA, double* sB,
                                              const double* x, double* U);

template <int N1, bool MODAL>
__device__ __forceinline__ void load_state_at_quad(const FastTables& F, const double* __restrict__ Q, int64_t K,
                                                   int64_t e0, bool active, int ev, int q, double* sA, double* sB,
                                                   double* U) {
  double x[4];
  issue_state_loads<N1>(Q, K, e0, active, x);
  state_at_quad<N1, MODAL>(F, active, ev, q, sA, sB, x, U);
}

template <int N1, bool MODAL>
__device__ __forceinline__ void state_at_quad(const FastTables& F, bool active, int ev, int q, double* sA, double* sB,
                                              const double* x, double* U) {
  constexpr int Nq = N1 * N1;
  const int tid = threadIdx.x;
  const bool inrange = tid < FCfg<N1>::E * Nq;   // lanes beyond E*Nq own no LDS slot
  if (!MODAL) {
#pragma unroll
    for (int f = 0; f < 4; ++f) U[f] = x[f];
    return;
  }
  const int lo = q % N1, hi = q / N1;
  double c[N1];
#pragma unroll
  for (int i = 0; i < N1; ++i) c[i] = F.Iq[lo * N1 + i];
  if (inrange) {
#pragma unroll
    for (int f = 0; f < 4; ++f) sA[(ev * 4 + f) * Nq + q] = x[f];
  }
  __syncthreads();
  // stage 1: W[b + N1 j] = sum_i Iq[b,i] Qn[i + N1 j]   (this lane: b = lo, j = hi)
#pragma unroll
  for (int f = 0; f < 4; ++f) {
    const double* src = sA + (ev * 4 + f) * Nq + N1 * hi;
    double s = 0.0;
#pragma unroll
    for (int i = 0; i < N1; ++i) s += c[i] * src[i];
    if (inrange) sB[(ev * 4 + f) * Nq + q] = s;
  }
  __syncthreads();
  // stage 2: Uq[a + N1 b] = sum_j Iq[a,j] W[b + N1 j]   (this lane: a = lo, b = hi)
#pragma unroll
  for (int f = 0; f < 4; ++f) {
    const double* src = sB + (ev * 4 + f) * Nq + hi;
    double s = 0.0;
#pragma unroll
    for (int j = 0; j < N1; ++j) s += c[j] * src[N1 * j];
    U[f] = s;
  }
  if (!active) { U[0] = 1.0; U[1] = 0.0; U[2] = 0.0; U[3] = 1.0; }
}

// ---- rhs store (+ Pq by sum factorisation) --------------------------------------------------
template <int N1, bool MODAL>
__device__ __forceinline__ void store_rhs_from_quad(const FastTables& F, double* __restrict__ rhs, int64_t K,
                                                    int64_t e0, bool active, int ev, int q, double* sA, double* sB,
                                                    const double* R) {
  constexpr int Nq = N1 * N1;
  const int tid = threadIdx.x;
  const bool inrange = tid < FCfg<N1>::E * Nq;
  double out[4];
  if (!MODAL) {
#pragma unroll
    for (int f = 0; f < 4; ++f) out[f] = R[f];
  } else {
    const int lo = q % N1, hi = q / N1;
    if (inrange) {
#pragma unroll
      for (int f = 0; f < 4; ++f) sA[(ev * 4 + f) * Nq + q] = R[f];
    }
    __syncthreads();
    // stage 1: W[i + N1 a] = sum_b Ip[i,b] R[a + N1 b]   (this lane: i = lo, a = hi)
#pragma unroll
    for (int f = 0; f < 4; ++f) {
      const double* src = sA + (ev * 4 + f) * Nq + hi;
      double s = 0.0;
#pragma unroll
      for (int b = 0; b < N1; ++b) s += F.Ip[lo * N1 + b] * src[N1 * b];
      if (inrange) sB[(ev * 4 + f) * Nq + q] = s;
    }
    __syncthreads();
    // stage 2: out[i + N1 j] = sum_a Ip[j,a] W[i + N1 a]   (this lane: i = lo, j = hi)
#pragma unroll
    for (int f = 0; f < 4; ++f) {
      const double* src = sB + (ev * 4 + f) * Nq + lo;
      double s = 0.0;
#pragma unroll
      for (int a = 0; a < N1; ++a) s += F.Ip[hi * N1 + a] * src[N1 * a];
      out[f] = s;
    }
  }
  if (active) {
#pragma unroll
    for (int f = 0; f < 4; ++f) rhs[(int64_t)f * K * Nq + e0 * Nq + tid] = out[f];
  }
}

// ---- viscous stages ----------------------------------------------------------------------------
// face lanes: half jumps of the projected entropy variables (+ penalty), dg_grad! :548-569, :817-822
template <int N1, bool WITH_PEN, bool PRE = false>
__device__ __forceinline__ void visc_face_jumps(const Tables& T, const MeshDev& M, const Phys& ph, int64_t e0,
                                                bool factive, int ef, int fn, const double* __restrict__ sV,
                                                const double* __restrict__ A_v, const double* vPre, double* sDv,
                                                double* sPen) {
  constexpr int Nq = N1 * N1, Nfq = 4 * N1;
  if (!factive) return;
  double vf[3] = {0, 0, 0};
#pragma unroll
  for (int t = 0; t < N1; ++t) {
    const double a = T.Ef_val[fn * T.wEf + t];
    const int col = T.Ef_idx[fn * T.wEf + t];
#pragma unroll
    for (int c = 0; c < 3; ++c) vf[c] += a * sV[(ef * 3 + c) * Nq + col];
  }
  double vP[3];
  if (PRE) {
    vP[0] = vPre[0]; vP[1] = vPre[1]; vP[2] = vPre[2];
  } else {
    const int64_t n = (e0 + ef) * Nfq + fn;
    const double* vp = A_v + (int64_t)M.mapP[n] * AV_NC;
    vP[0] = vp[0]; vP[1] = vp[1]; vP[2] = vp[2];
  }
  const double tau = -1 / ph.Re / vf[2];
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    const double dv = vP[c] - vf[c];
    sDv[(ef * 3 + c) * Nfq + fn] = .5 * dv;
    if (WITH_PEN) sPen[(ef * 3 + c) * Nfq + fn] = tau * dv;
  }
}

// volume lanes: BR1 gradient at this Gauss node and sigma = K(v) grad v
template <int N1>
__device__ __forceinline__ void visc_sigma(const Tables& T, const FastTables& F, const Phys& ph, const double* g,
                                           int ev, int q, const double* __restrict__ sV,
                                           const double* __restrict__ sDv, double* sgx, double* sgy) {
  constexpr int Nq = N1 * N1, Nfq = 4 * N1;
  double vr[3] = {0, 0, 0}, vs[3] = {0, 0, 0};
#pragma unroll
  for (int t = 0; t < N1; ++t) {
    const double ar = T.Dr_val[q * T.wD + t], as = T.Ds_val[q * T.wD + t];
    const int cr = T.Dr_idx[q * T.wD + t], cs = T.Ds_idx[q * T.wD + t];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      vr[c] += ar * sV[(ev * 3 + c) * Nq + cr];
      vs[c] += as * sV[(ev * 3 + c) * Nq + cs];
    }
  }
  double lx[3] = {0, 0, 0}, ly[3] = {0, 0, 0};
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    const double a = F.pl_lf[q * 4 + t];
    const int fn = F.pl_fn[q * 4 + t];
    const double* gn = g + 5 + 3 * (fn / N1);
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const double d = sDv[(ev * 3 + c) * Nfq + fn];
      lx[c] += a * (d * gn[0]);
      ly[c] += a * (d * gn[1]);
    }
  }
  const double iJ = 1.0 / g[4];
  double tx[3], ty[3];
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    tx[c] = ((g[0] * vr[c] + g[1] * vs[c]) + lx[c]) * iJ;
    ty[c] = ((g[2] * vr[c] + g[3] * vs[c]) + ly[c]) * iJ;
  }
  const double v[3] = {sV[(ev * 3 + 0) * Nq + q], sV[(ev * 3 + 1) * Nq + q], sV[(ev * 3 + 2) * Nq + q]};
  viscous_stress(v, tx, ty, -ph.lambda, ph.mu, ph.Pr, sgx, sgy);
}

template <int N1>
__device__ __forceinline__ void face_normal_stress(const Tables& T, const double* __restrict__ sS, int ef, int fn,
                                                   double nxJ, double nyJ, double* sn) {
  constexpr int Nq = N1 * N1;
  double fx[3] = {0, 0, 0}, fy[3] = {0, 0, 0};
#pragma unroll
  for (int t = 0; t < N1; ++t) {
    const double a = T.Ef_val[fn * T.wEf + t];
    const int col = T.Ef_idx[fn * T.wEf + t];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      fx[c] += a * sS[(ef * 6 + c) * Nq + col];
      fy[c] += a * sS[(ef * 6 + 3 + c) * Nq + col];
    }
  }
#pragma unroll
  for (int c = 0; c < 3; ++c) sn[c] = fx[c] * nxJ + fy[c] * nyJ;
}

}  // namespace fastdev

using namespace fastdev;

// ---------------------------------------------------------------------------------------------
// phase 0
// ---------------------------------------------------------------------------------------------
template <int N1, bool MODAL, bool VISC>
__global__ __launch_bounds__(FCfg<N1>::T) void kf_project(Tables T, FastTables F, MeshDev M, Phys ph,
                                                          const double* __restrict__ Q, double* __restrict__ A_U,
                                                          double* __restrict__ A_v) {
  constexpr int Nq = N1 * N1, Nfq = 4 * N1, E = FCfg<N1>::E;
  __shared__ double sA[E * 4 * Nq];
  __shared__ double sB[E * 4 * Nq];
  const int tid = threadIdx.x;
  const int64_t e0 = (int64_t)blockIdx.x * E;
  const int nE = (int)min((int64_t)E, M.K - e0);
  const int ev = tid / Nq, q = tid - ev * Nq;
  const bool vactive = tid < nE * Nq;
  const int ef = tid / Nfq, fn = tid - ef * Nfq;
  const bool factive = tid < nE * Nfq;

  double U[4];
  load_state_at_quad<N1, MODAL>(F, Q, M.K, e0, vactive, ev < E ? ev : 0, q, sA, sB, U);
  double qh[6], V[4];
  prim_logs<MODAL>(U, qh);
  v_of_prim<MODAL>(qh, V);
  __syncthreads();
  if (vactive) {
#pragma unroll
    for (int c = 0; c < 4; ++c) sA[(ev * 4 + c) * Nq + q] = V[c];
  }
  __syncthreads();
  if (factive) {
    double Vf[4] = {0, 0, 0, 0};
#pragma unroll
    for (int t = 0; t < N1; ++t) {
      const double a = T.Ef_val[fn * T.wEf + t];
      const int col = T.Ef_idx[fn * T.wEf + t];
#pragma unroll
      for (int c = 0; c < 4; ++c) Vf[c] += a * sA[(ef * 4 + c) * Nq + col];
    }
    double Uf[4], qf[6];
    u_of_v<MODAL>(Vf, Uf);
    prim_logs<MODAL>(Uf, qf);
    const double* g = M.geo + (e0 + ef) * GEO_STRIDE + 5 + 3 * (fn / N1);
    const double lam = lf_lambda<MODAL>(Uf, g[0], g[1], g[2]);
    const int64_t n = (e0 + ef) * Nfq + fn;
    double2* a = reinterpret_cast<double2*>(A_U + n * FAU_NC);
    a[0] = make_double2(qf[0], qf[1]);
    a[1] = make_double2(qf[2], qf[3]);
    a[2] = make_double2(qf[4], qf[5]);
    a[3] = make_double2(lam, Uf[3]);
    if (VISC) {
      double* b = A_v + n * AV_NC;
      b[0] = Vf[1]; b[1] = Vf[2]; b[2] = Vf[3];
    }
  }
}

// ---------------------------------------------------------------------------------------------
// phase 1 (CNS): sigma and its normal traces
// ---------------------------------------------------------------------------------------------
template <int N1>
__global__ __launch_bounds__(FCfg<N1>::T) void kf_sigma(Tables T, FastTables F, MeshDev M, Phys ph,
                                                        const double* __restrict__ Q, const double* __restrict__ A_v,
                                                        double* __restrict__ B) {
  constexpr int Nq = N1 * N1, Nfq = 4 * N1, E = FCfg<N1>::E;
  __shared__ double sA[E * 4 * Nq];
  __shared__ double sB[E * 4 * Nq];
  __shared__ double sV[E * 3 * Nq];
  __shared__ double sDv[E * 3 * Nfq];
  __shared__ double sS[E * 6 * Nq];
  const int tid = threadIdx.x;
  const int64_t e0 = (int64_t)blockIdx.x * E;
  const int nE = (int)min((int64_t)E, M.K - e0);
  const int ev = tid / Nq, q = tid - ev * Nq;
  const bool vactive = tid < nE * Nq;
  const int ef = tid / Nfq, fn = tid - ef * Nfq;
  const bool factive = tid < nE * Nfq;

  double U[4];
  load_state_at_quad<N1, true>(F, Q, M.K, e0, vactive, ev < E ? ev : 0, q, sA, sB, U);
  double qh[6], V[4];
  prim_logs<true>(U, qh);
  v_of_prim<true>(qh, V);
  if (vactive) {
#pragma unroll
    for (int c = 0; c < 3; ++c) sV[(ev * 3 + c) * Nq + q] = V[c + 1];
  }
  __syncthreads();
  visc_face_jumps<N1, false>(T, M, ph, e0, factive, ef, fn, sV, A_v, nullptr, sDv, nullptr);
  __syncthreads();
  if (vactive) {
    double sgx[3], sgy[3];
    visc_sigma<N1>(T, F, ph, M.geo + (e0 + ev) * GEO_STRIDE, ev, q, sV, sDv, sgx, sgy);
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      sS[(ev * 6 + c) * Nq + q] = sgx[c];
      sS[(ev * 6 + 3 + c) * Nq + q] = sgy[c];
    }
  }
  __syncthreads();
  if (factive) {
    const double* gn = M.geo + (e0 + ef) * GEO_STRIDE + 5 + 3 * (fn / N1);
    double sn[3];
    face_normal_stress<N1>(T, sS, ef, fn, gn[0], gn[1], sn);
    double* b = B + ((e0 + ef) * Nfq + fn) * B_NC;
    b[0] = sn[0]; b[1] = sn[1]; b[2] = sn[2];
  }
}

// ---------------------------------------------------------------------------------------------
// last phase
// ---------------------------------------------------------------------------------------------
template <int N1, bool VISC>
struct RhsLds {
  static constexpr int Nq = N1 * N1, Nfq = 4 * N1, Nh = Nq + Nfq, E = FCfg<N1>::E, NF = N1 / 2, NS = NF + 2;
  static constexpr int nQh = E * Nh * 6;
  static constexpr int nV = VISC ? E * 3 * Nq : 0;
  static constexpr int nQFf = E * Nfq * 4;
  static constexpr int nFl = E * Nfq * 4;
  static constexpr int nX_flux = E * Nq * NS * 4;
  static constexpr int nX_interp = 2 * E * 4 * Nq;
  static constexpr int nX_visc = VISC ? (E * 9 * Nfq + E * 6 * Nq) : 0;
  static constexpr int nX = nX_flux > nX_interp ? (nX_flux > nX_visc ? nX_flux : nX_visc)
                                                : (nX_interp > nX_visc ? nX_interp : nX_visc);
  static constexpr int total = nQh + nV + nQFf + nFl + nX;
};

template <int N1, bool MODAL, bool VISC>
__global__ __launch_bounds__(FCfg<N1>::T) void kf_rhs(Tables T, FastTables F, MeshDev M, Phys ph,
                                                      const double* __restrict__ Q, const double* __restrict__ A_U,
                                                      const double* __restrict__ A_v, const double* __restrict__ B,
                                                      double* __restrict__ rhs) {
  using L = RhsLds<N1, VISC>;
  constexpr int Nq = L::Nq, Nfq = L::Nfq, Nh = L::Nh, E = L::E, NF = L::NF, NS = L::NS;
  __shared__ __align__(16) double lds[L::total];
  double* sQh = lds;                 // [E][Nh][6]
  double* sV = sQh + L::nQh;         // [E][3][Nq]   (VISC)
  double* sQFf = sV + L::nV;         // [E][Nfq][4]
  double* sFl = sQFf + L::nQFf;      // [E][Nfq][4]
  double* sX = sFl + L::nFl;         // exchange [E][Nq][NS][4]; also interp scratch and viscous scratch
  const int tid = threadIdx.x;
  const int64_t e0 = (int64_t)blockIdx.x * E;
  const int nE = (int)min((int64_t)E, M.K - e0);
  const int ev_raw = tid / Nq, q = tid - ev_raw * Nq;
  const bool vactive = tid < nE * Nq;
  const int ev = ev_raw < E ? ev_raw : 0;
  const int ef = tid / Nfq, fn = tid - ef * Nfq;
  const bool factive = tid < nE * Nfq;
  const double* g = M.geo + (e0 + (vactive ? ev : 0)) * GEO_STRIDE;

  // ---- issue every global load this block needs up front: the state first (needed first; vmcnt
  //      retires in order), then the face traces, which depend on nothing computed here ----------
  double xq[4];
  issue_state_loads<N1>(Q, M.K, e0, vactive, xq);
  double qM[8], qP[8], vPn[3] = {0, 0, 0}, bPn[3] = {0, 0, 0};
  if (factive && !(ph.dbg & 4)) {
    const int64_t n = (e0 + ef) * Nfq + fn;
    const int64_t mp = M.mapP[n];
    const double2* aM = reinterpret_cast<const double2*>(A_U + n * FAU_NC);
    const double2* aP = reinterpret_cast<const double2*>(A_U + mp * FAU_NC);
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const double2 m = aM[c], p = aP[c];
      qM[2 * c] = m.x; qM[2 * c + 1] = m.y;
      qP[2 * c] = p.x; qP[2 * c + 1] = p.y;
    }
    if (VISC) {
      const double* vp = A_v + mp * AV_NC;
      const double* bp = B + mp * B_NC;
#pragma unroll
      for (int c = 0; c < 3; ++c) { vPn[c] = vp[c]; bPn[c] = bp[c]; }
    }
  }
  // ---- state at the Gauss node, primitives + logs ------------------------------------------
  double U[4];
  state_at_quad<N1, MODAL>(F, vactive, ev, q, sX, sX + E * 4 * Nq, xq, U);
  double qh[6];
  prim_logs<MODAL>(U, qh);
  if (vactive) {
    double2* d = reinterpret_cast<double2*>(sQh + (ev * Nh + q) * 6);
    d[0] = make_double2(qh[0], qh[1]);
    d[1] = make_double2(qh[2], qh[3]);
    d[2] = make_double2(qh[4], qh[5]);
    if (VISC) {
      double V[4];
      v_of_prim<MODAL>(qh, V);
#pragma unroll
      for (int c = 0; c < 3; ++c) sV[(ev * 3 + c) * Nq + q] = V[c + 1];
    }
  }
  // ---- face lanes: interface flux from the prefetched own + neighbour traces -----------------
  if (factive && !(ph.dbg & 4)) {
    double2* d = reinterpret_cast<double2*>(sQh + (ef * Nh + Nq + fn) * 6);
    d[0] = make_double2(qM[0], qM[1]);
    d[1] = make_double2(qM[2], qM[3]);
    d[2] = make_double2(qM[4], qM[5]);
    double Fx[4], Fy[4];
    ec_flux_fast<MODAL>(qM, qP, Fx, Fy);
    const double* gn = M.geo + (e0 + ef) * GEO_STRIDE + 5 + 3 * (fn / N1);
    const double LFc = ph.inviscid_dissp ? ph.lf_scale * fmax(qM[6], qP[6]) * gn[2] : 0.0;
    const double dU[4] = {qP[0] - qM[0], qP[0] * qP[1] - qM[0] * qM[1], qP[0] * qP[2] - qM[0] * qM[2], qP[7] - qM[7]};
    double2* o = reinterpret_cast<double2*>(sFl + (ef * Nfq + fn) * 4);
    o[0] = make_double2(Fx[0] * gn[0] + Fy[0] * gn[1] - LFc * dU[0], Fx[1] * gn[0] + Fy[1] * gn[1] - LFc * dU[1]);
    o[1] = make_double2(Fx[2] * gn[0] + Fy[2] * gn[1] - LFc * dU[2], Fx[3] * gn[0] + Fy[3] * gn[1] - LFc * dU[3]);
  }
  __syncthreads();
  // ---- flux differencing along the tensor lines --------------------------------------------
  double acc[4] = {0, 0, 0, 0};
#pragma unroll
  for (int d = 0; d < 2; ++d) {
    if (ph.dbg & 1) break;   // ablation (diagnostic): no flux differencing
    if (vactive) {
      const int base = (d * Nq + q);
#pragma unroll
      for (int k = 0; k < NS; ++k) {
        int pid;
        double cr, cs;
        if (k < NF) {
          pid = F.fwd_id[base * NF + k];
          cr = F.fwd_c[(base * NF + k) * 2];
          cs = F.fwd_c[(base * NF + k) * 2 + 1];
        } else {
          pid = Nq + F.face_id[base * 2 + (k - NF)];
          cr = F.face_c[(base * 2 + (k - NF)) * 2];
          cs = F.face_c[(base * 2 + (k - NF)) * 2 + 1];
        }
        double v4[4] = {0, 0, 0, 0};
        if (pid != 0xFF) {
          const double2* pp = reinterpret_cast<const double2*>(sQh + (ev * Nh + pid) * 6);
          const double2 p0 = pp[0], p1 = pp[1], p2 = pp[2];
          const double qj[6] = {p0.x, p0.y, p1.x, p1.y, p2.x, p2.y};
          double Fx[4], Fy[4];
          ec_flux_fast<MODAL>(qh, qj, Fx, Fy);
          const double cx = 2 * (g[0] * cr + g[1] * cs), cy = 2 * (g[2] * cr + g[3] * cs);
#pragma unroll
          for (int c = 0; c < 4; ++c) {
            v4[c] = cx * Fx[c] + cy * Fy[c];
            acc[c] += v4[c];
          }
        }
        double2* o = reinterpret_cast<double2*>(sX + ((ev * Nq + q) * NS + k) * 4);
        o[0] = make_double2(v4[0], v4[1]);
        o[1] = make_double2(v4[2], v4[3]);
      }
    }
    __syncthreads();
    if (vactive) {
      const int base = (d * Nq + q);
#pragma unroll
      for (int k = 0; k < NF; ++k) {
        const int p = F.bwd_src[base * NF + k];
        if (p != 0xFF) {
          const double2* o = reinterpret_cast<const double2*>(sX + ((ev * Nq + p) * NS + k) * 4);
          const double2 a = o[0], b = o[1];
          acc[0] -= a.x; acc[1] -= a.y; acc[2] -= b.x; acc[3] -= b.y;
        }
      }
    }
    if (factive && F.fr_dir[fn] == d) {
      const int slot = NF + F.fr_slot[fn];
      double s[4] = {0, 0, 0, 0};
#pragma unroll
      for (int j = 0; j < N1; ++j) {
        const int p = F.fr_src[fn * N1 + j];
        const double2* o = reinterpret_cast<const double2*>(sX + ((ef * Nq + p) * NS + slot) * 4);
        const double2 a = o[0], b = o[1];
        s[0] -= a.x; s[1] -= a.y; s[2] -= b.x; s[3] -= b.y;
      }
      double2* o = reinterpret_cast<double2*>(sQFf + (ef * Nfq + fn) * 4);
      o[0] = make_double2(s[0], s[1]);
      o[1] = make_double2(s[2], s[3]);
    }
    __syncthreads();
  }
  // ---- collocated rhs: -(Ph*QF + Lf*flux)/J ---------------------------------------------------
  double R[4];
  {
    const double pd = F.ph_diag[q];
    double a[4] = {pd * acc[0], pd * acc[1], pd * acc[2], pd * acc[3]};
    if (vactive) {
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const int f = F.pl_fn[q * 4 + t];
        const double wp = F.pl_ph[q * 4 + t], wl = F.pl_lf[q * 4 + t];
        const double2* x = reinterpret_cast<const double2*>(sQFf + (ev * Nfq + f) * 4);
        const double2* y = reinterpret_cast<const double2*>(sFl + (ev * Nfq + f) * 4);
        const double2 x0 = x[0], x1 = x[1], y0 = y[0], y1 = y[1];
        a[0] += wp * x0.x + wl * y0.x;
        a[1] += wp * x0.y + wl * y0.y;
        a[2] += wp * x1.x + wl * y1.x;
        a[3] += wp * x1.y + wl * y1.y;
      }
    }
    const double J = g[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) R[c] = -a[c] / J;
  }
  // ---- viscous terms ---------------------------------------------------------------------------
  if (VISC && !(ph.dbg & 2)) {
    double* sDv = sX;                    // [E][3][Nfq]
    double* sPen = sDv + E * 3 * Nfq;    // [E][3][Nfq]
    double* sSj = sPen + E * 3 * Nfq;    // [E][3][Nfq]
    double* sS = sSj + E * 3 * Nfq;      // [E][6][Nq]
    visc_face_jumps<N1, true, true>(T, M, ph, e0, factive, ef, fn, sV, A_v, vPn, sDv, sPen);
    __syncthreads();
    if (vactive) {
      double sgx[3], sgy[3];
      visc_sigma<N1>(T, F, ph, g, ev, q, sV, sDv, sgx, sgy);
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        sS[(ev * 6 + c) * Nq + q] = sgx[c];
        sS[(ev * 6 + 3 + c) * Nq + q] = sgy[c];
      }
    }
    __syncthreads();
    if (factive) {
      const double* gn = M.geo + (e0 + ef) * GEO_STRIDE + 5 + 3 * (fn / N1);
      double sn[3];
      face_normal_stress<N1>(T, sS, ef, fn, gn[0], gn[1], sn);
#pragma unroll
      for (int c = 0; c < 3; ++c) sSj[(ef * 3 + c) * Nfq + fn] = .5 * (-bPn[c] - sn[c]);
    }
    __syncthreads();
    if (vactive) {
      double dxr[3] = {0, 0, 0}, dxs[3] = {0, 0, 0}, dyr[3] = {0, 0, 0}, dys[3] = {0, 0, 0};
#pragma unroll
      for (int t = 0; t < N1; ++t) {
        const double ar = T.Dr_val[q * T.wD + t], as = T.Ds_val[q * T.wD + t];
        const int cr = T.Dr_idx[q * T.wD + t], cs = T.Ds_idx[q * T.wD + t];
#pragma unroll
        for (int c = 0; c < 3; ++c) {
          dxr[c] += ar * sS[(ev * 6 + c) * Nq + cr];
          dxs[c] += as * sS[(ev * 6 + c) * Nq + cs];
          dyr[c] += ar * sS[(ev * 6 + 3 + c) * Nq + cr];
          dys[c] += as * sS[(ev * 6 + 3 + c) * Nq + cs];
        }
      }
      double sf[3] = {0, 0, 0}, pn[3] = {0, 0, 0};
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const double w = F.pl_lf[q * 4 + t];
        const int col = F.pl_fn[q * 4 + t];
#pragma unroll
        for (int c = 0; c < 3; ++c) {
          sf[c] += w * sSj[(ev * 3 + c) * Nfq + col];
          pn[c] += w * sPen[(ev * 3 + c) * Nfq + col];
        }
      }
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        double r = ((g[0] * dxr[c] + g[1] * dxs[c] + g[2] * dyr[c] + g[3] * dys[c]) + sf[c]) / g[4];
        if (ph.viscous_dissp) r += pn[c];
        R[c + 1] += r;
      }
    }
    __syncthreads();
  }
  store_rhs_from_quad<N1, MODAL>(F, rhs, M.K, e0, vactive, ev, q, sX, sX + E * 4 * Nq, R);
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

int launch_project_fast(const Tables& T, const FastTables& F, const MeshDev& M, const Phys& ph, const double* Q,
                        double* A_U, double* A_v, hipStream_t s) {
  if (M.K == 0) return 0;
  const bool modal = ph.formulation != 0, visc = ph.formulation == 1;
  ESDG_DISPATCH_N1(T.N1, {
    constexpr int E = FCfg<N1>::E;
    constexpr int TT = FCfg<N1>::T;
    const int nb = (int)((M.K + E - 1) / E);
    if (!modal)
      hipLaunchKernelGGL((kf_project<N1, false, false>), dim3(nb), dim3(TT), 0, s, T, F, M, ph, Q, A_U, A_v);
    else if (visc)
      hipLaunchKernelGGL((kf_project<N1, true, true>), dim3(nb), dim3(TT), 0, s, T, F, M, ph, Q, A_U, A_v);
    else
      hipLaunchKernelGGL((kf_project<N1, true, false>), dim3(nb), dim3(TT), 0, s, T, F, M, ph, Q, A_U, A_v);
  });
  return (int)hipGetLastError();
}

int launch_sigma_fast(const Tables& T, const FastTables& F, const MeshDev& M, const Phys& ph, const double* Q,
                      const double* A_v, double* B, hipStream_t s) {
  if (M.K == 0) return 0;
  ESDG_DISPATCH_N1(T.N1, {
    constexpr int E = FCfg<N1>::E;
    constexpr int TT = FCfg<N1>::T;
    const int nb = (int)((M.K + E - 1) / E);
    hipLaunchKernelGGL((kf_sigma<N1>), dim3(nb), dim3(TT), 0, s, T, F, M, ph, Q, A_v, B);
  });
  return (int)hipGetLastError();
}

int launch_rhs_fast(const Tables& T, const FastTables& F, const MeshDev& M, const Phys& ph, const double* Q,
                    const double* A_U, const double* A_v, const double* B, double* rhs, hipStream_t s) {
  if (M.K == 0) return 0;
  const bool modal = ph.formulation != 0, visc = ph.formulation == 1;
  ESDG_DISPATCH_N1(T.N1, {
    constexpr int E = FCfg<N1>::E;
    constexpr int TT = FCfg<N1>::T;
    const int nb = (int)((M.K + E - 1) / E);
    if (!modal)
      hipLaunchKernelGGL((kf_rhs<N1, false, false>), dim3(nb), dim3(TT), 0, s, T, F, M, ph, Q, A_U, A_v, B, rhs);
    else if (visc)
      hipLaunchKernelGGL((kf_rhs<N1, true, true>), dim3(nb), dim3(TT), 0, s, T, F, M, ph, Q, A_U, A_v, B, rhs);
    else
      hipLaunchKernelGGL((kf_rhs<N1, true, false>), dim3(nb), dim3(TT), 0, s, T, F, M, ph, Q, A_U, A_v, B, rhs);
  });
  return (int)hipGetLastError();
}

}  // namespace esdg
