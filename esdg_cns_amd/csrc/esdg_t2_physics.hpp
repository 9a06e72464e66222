// esdg_t2_physics.hpp -- pointwise physics of the 2D tensor kernels (esdg_kernels_tensor2.hip, esdg_kernels_tensor3.hip):
// entropy-conservative directional flux with the wave-uniform log-mean variants, primitive / entropy-variable maps, viscous
// stress.  Reference citations at each function.
#pragma once
#include "esdg_dev.hpp"
#include "esdg_devmath.hpp"
#include "esdg_wall_closures.hpp"

namespace esdg {
namespace t2 {

using namespace devmath;

typedef double2 d2;

constexpr double GM1 = 0.4;   // the CNS drivers' literal (cavity_optimized.jl:463)

// viscous_matrices! + sigma rows 2..4 (cavity :613-645, 786-801); lam already sign-flipped (quirk Q4); gk = gamma*mu/Pr
__device__ __forceinline__ void viscous_stress(const double* v, const double* tx, const double* ty, double lam, double mu,
                                               double gk, double* sx, double* sy) {
  const double v2 = v[0], v3 = v[1], v4 = v[2];
  const double i1 = rcp_refined(v4);
  const double i2 = i1 * i1, i3 = i2 * i1;
  const double l2m = lam + 2.0 * mu;
  const double a24 = v2 * i2, a34 = v3 * i2;
  const double Kxx22 = -l2m * i1, Kxx24 = l2m * a24, Kxx33 = -mu * i1, Kxx34 = mu * a34,
               Kxx44 = -i3 * (l2m * (v2 * v2) + mu * (v3 * v3) - gk * v4);
  const double Kxy23 = -lam * i1, Kxy24 = lam * a34, Kxy32 = Kxx33, Kxy34 = mu * a24, Kxy42 = Kxx34, Kxy43 = lam * a24,
               Kxy44 = i3 * (lam + mu) * (-v2 * v3);
  const double Kyy22 = Kxx33, Kyy24 = Kxy34, Kyy33 = Kxx22, Kyy34 = l2m * a34,
               Kyy44 = -i3 * (l2m * (v3 * v3) + mu * (v2 * v2) - gk * v4);
  sx[0] = Kxx22 * tx[0] + Kxx24 * tx[2] + Kxy23 * ty[1] + Kxy24 * ty[2];
  sx[1] = Kxx33 * tx[1] + Kxx34 * tx[2] + Kxy32 * ty[0] + Kxy34 * ty[2];
  sx[2] = Kxx24 * tx[0] + Kxx34 * tx[1] + Kxx44 * tx[2] + Kxy42 * ty[0] + Kxy43 * ty[1] + Kxy44 * ty[2];
  sy[0] = Kxy32 * tx[1] + Kxy42 * tx[2] + Kyy22 * ty[0] + Kyy24 * ty[2];
  sy[1] = Kxy23 * tx[0] + Kxy43 * tx[2] + Kyy33 * ty[1] + Kyy34 * ty[2];
  sy[2] = Kxy24 * tx[0] + Kxy34 * tx[1] + Kxy44 * tx[2] + Kyy24 * ty[0] + Kyy34 * ty[1] + Kyy44 * ty[2];
}

template <bool MODAL> struct Gas2 { static constexpr double GM1 = MODAL ? 0.4 : (1.4 - 1); };   // literal 0.4 in the CNS drivers

// (gx,gy) . (Fx,Fy) of the entropy-conservative flux (euler_fluxes.jl:23-48 with logmean.jl:14-28) between the states
// (rho,u,v,beta,lrho,lbeta); one refined reciprocal serves the three quotients.  The reference's |f| < 1e-4 series branch
// (logmean.jl:23-27) is taken per lane by selection (MODE 0); when EVERY lane of the wave takes the series for both means
// (MODE 1: smooth regions, e.g. the far field of the vortex) or NO lane takes it for either (MODE 2) the unselected
// half is not computed at all -- 13-15 of ~64 VALU instructions per flux.  The three variants evaluate the same
// expressions with explicit FMAs, so a lane's result does not depend on which variant its wave ran (the ranged-launch
// and shard tests compare bit for bit across different wave compositions).
//
// Round 5 -- the same values from SUMS instead of averages.  Every .5 of the formulas (the four averages, the half of the kinetic
// term) is a power of two and commutes with rounding, so it can ride in a constant instead of costing an instruction: the cores
// take srho = rho_L + rho_R and sbeta = beta_L + beta_R (= the reference's yp, which the pressure average needs anyway), run the
// series polynomials on (f/2)^2 with coefficients scaled by exact powers of two, keep 2 pa and 2 f4aux, and absorb the halves in
// the metric vector (.5 gx, .5 gy: loop invariants of the callers) -- rholog, 1/betalog and the four flux components are BIT FOR BIT
// those of the formulation with averages (every intermediate is the old one times an exact power of two), at 3 instructions
// less per flux.
// The innermost step of either Horner chain has two non-inline constants, and gfx950 VOP3 reads at most one operand from the
// constant bus (an SGPR pair; there are no 64-bit literals): the other must sit in VGPRs.  Left to itself hipcc rematerialises it in
// front of every flux that sits behind a branch (two v_mov_b32 + a v_mov_b64 copy for the destructive v_fmac: ~8 of a flux's ~50
// instructions in kt3_rhs).  SeriesK carries the two constants; a kernel that evaluates many fluxes passes a pinned copy
// (series_k_pinned: opaque to the compiler, so they stay in six VGPRs for the stage), the others the literals.
struct SeriesK { double r2, r1, b1; };
__device__ __forceinline__ SeriesK series_k() { return {8 * -.0512, 2 * -.2, 8 * .2}; }
__device__ __forceinline__ SeriesK series_k_pinned() {
  SeriesK k = series_k();
  asm volatile("" : "+v"(k.r2), "+v"(k.r1), "+v"(k.b1));
  return k;
}
// half of the rho series in w = (f/2)^2:  .5 (1 + 4w (-.2 + 4w (-.0512 + 4w c3)))
__device__ __forceinline__ double logmean_series_rho_h(double w, const SeriesK& k) {
  return __builtin_fma(w, __builtin_fma(w, __builtin_fma(w, 32 * 0.026038857142857, k.r2), k.r1), .5);
}
// twice the 1/beta series in w = (f/2)^2:  2 (1 + 4w (.2 + 4w .0912))
__device__ __forceinline__ double logmean_series_ibeta_2(double w, const SeriesK& k) {
  return __builtin_fma(w, __builtin_fma(w, 32 * .0912, k.b1), 2.0);
}
// dr = rho_R - rho_L, sr = rho_R + rho_L, db, sb likewise; ser_r / ser_b: this lane takes the series for rho / beta (MODE 0)
template <bool MODAL, int MODE>
__device__ __forceinline__ void ec_flux_core(const double* qL, const double* qR, double gx, double gy, double* F, double dr, double sr,
                                             double db, double sb, bool ser_r, bool ser_b, const SeriesK& sk = series_k()) {
  constexpr double GM1 = Gas2<MODAL>::GM1;
  // yr: 2 rho_avg (series) or log rho_L - log rho_R; yb: 2 beta_avg (series) or beta_R - beta_L; yp = beta_L + beta_R
  double yr, yb;
  if (MODE == 1) { yr = sr; yb = sb; }
  else {
    const double A = qL[4] - qR[4];
    yr = MODE == 2 ? A : (ser_r ? sr : A);
    yb = MODE == 2 ? db : (ser_b ? sb : db);
  }
  const double ybp = yb * sb;
  const double R = rcp_refined(yr * ybp);
  const double ir = R * ybp;     // 1 / yr
  const double ryr = R * yr;
  const double ib = ryr * sb;    // 1 / yb
  const double ip = ryr * yb;    // 1 / (beta_L + beta_R)
  const double fr = dr * ir;     // series lanes: f / 2
  const double fb = db * ib;
  double rholog, ibetalog;
  if (MODE == 2) {
    rholog = -fr;
    ibetalog = -((qL[5] - qR[5]) * ib);
  } else {
    const double srs = sr * logmean_series_rho_h(fr * fr, sk), sbs = ib * logmean_series_ibeta_2(fb * fb, sk);
    if (MODE == 1) { rholog = srs; ibetalog = sbs; }
    else { rholog = ser_r ? srs : -fr; ibetalog = ser_b ? sbs : -((qL[5] - qR[5]) * ib); }
  }
  const double su = qL[1] + qR[1], sv = qL[2] + qR[2];
  const double unorm = __builtin_fma(qL[2], qR[2], qL[1] * qR[1]);
  const double pa2 = sr * ip;                                                                   // 2 pa
  const double f4aux2 = __builtin_fma(rholog, __builtin_fma(ibetalog, 1.0 / GM1, unorm), pa2);   // 2 f4aux = rholog (1/(g-1) / betalog + uL.uR) + 2 pa
  const double hgx = .5 * gx, hgy = .5 * gy;
  const double un = __builtin_fma(hgy, sv, hgx * su);
  F[0] = rholog * un;
  const double hF0 = .5 * F[0];
  F[1] = __builtin_fma(hF0, su, pa2 * hgx);
  F[2] = __builtin_fma(hF0, sv, pa2 * hgy);
  F[3] = f4aux2 * (.5 * un);
}
template <bool MODAL>
__device__ __forceinline__ void ec_flux_dir(const double* qL, const double* qR, double gx, double gy, double* F, const SeriesK& sk = series_k()) {
  const double dr = qR[0] - qL[0], sr = qR[0] + qL[0];
  const double db = qR[3] - qL[3], sb = qR[3] + qL[3];
  const bool ser_r = fabs(dr) < (.5 * 1e-4) * sr, ser_b = fabs(db) < (.5 * 1e-4) * sb;
#if defined(ESDG_T2_FORCE_MODE)   // ISA-attribution builds only (tools/isa_buckets.py): one variant, no ballots -- straight-line code
  ec_flux_core<MODAL, ESDG_T2_FORCE_MODE>(qL, qR, gx, gy, F, dr, sr, db, sb, ser_r, ser_b, sk);
#elif defined(ESDG_T2_NO_UNIFORM_LOGMEAN)
  ec_flux_core<MODAL, 0>(qL, qR, gx, gy, F, dr, sr, db, sb, ser_r, ser_b, sk);
#else
  // (one ballot per comparison -- the v_cmp's own lane mask -- and scalar logic on the masks: the ballot of `ser_r && ser_b` made hipcc
  // turn the combined mask into a VGPR and compare it again, twice per flux)
  const unsigned long long active = __builtin_amdgcn_ballot_w64(true), br = __builtin_amdgcn_ballot_w64(ser_r), bb = __builtin_amdgcn_ballot_w64(ser_b);
  if ((br & bb) == active) ec_flux_core<MODAL, 1>(qL, qR, gx, gy, F, dr, sr, db, sb, ser_r, ser_b, sk);
  else if ((br | bb) == 0) ec_flux_core<MODAL, 2>(qL, qR, gx, gy, F, dr, sr, db, sb, ser_r, ser_b, sk);
  else ec_flux_core<MODAL, 0>(qL, qR, gx, gy, F, dr, sr, db, sb, ser_r, ser_b, sk);
#endif
}

// conservative -> (rho,u,v,beta)
template <bool MODAL>
__device__ __forceinline__ void prims(const double* U, double* q) {
  constexpr double GM1 = Gas2<MODAL>::GM1;
  const double m2 = U[1] * U[1] + U[2] * U[2];
  const double rre = U[0] * U[3] - .5 * m2;
  const double R = rcp_refined(U[0] * rre);
  const double ir = R * rre;
  q[0] = U[0];
  q[1] = U[1] * ir;
  q[2] = U[2] * ir;
  q[3] = (U[0] * U[0]) * (U[0] * R) * (1.0 / (2 * GM1));
}
// conservative -> (rho,u,v,beta,log rho,log beta)
template <bool MODAL>
__device__ __forceinline__ void prim_logs(const double* U, double* q) {
  prims<MODAL>(U, q);
  q[4] = log_pos(U[0]);
  q[5] = log_pos(q[3]);
}

// entropy variables from primitives + logs (identities of euler_variables.jl:79-92)
template <bool MODAL>
__device__ __forceinline__ void v_of_prim2(const double* q, double* V) {
  constexpr double GM1 = Gas2<MODAL>::GM1;
  const double s = -GM1 * q[4] - q[5] - 0.6931471805599453;
  const double b2 = 2 * GM1 * q[3];
  V[0] = 1.4 - s - .5 * b2 * (q[1] * q[1] + q[2] * q[2]);
  V[1] = b2 * q[1];
  V[2] = b2 * q[2];
  V[3] = -b2;
}
// (rho, u, v, beta) of entropy variables: u_vfun (euler_variables.jl:95-120 / cavity :473-478, no pow) followed by the
// primitive conversion of the conservative state it returns
template <bool MODAL>
__device__ __forceinline__ void prim_of_v2(const double* V, double* q) {
  constexpr double GM1 = Gas2<MODAL>::GM1;
  const double vUnorm = V[1] * V[1] + V[2] * V[2];
  const double h = vUnorm * .5 * rcp_refined(V[3]);
  const double s = 1.4 - V[0] + h;
  const double rhoeV = exp((log(GM1) - 1.4 * log_pos(-V[3]) - s) * (1.0 / GM1));
  const double U[4] = {rhoeV * (-V[3]), rhoeV * V[1], rhoeV * V[2], rhoeV * (1 - h)};
  const double m2 = U[1] * U[1] + U[2] * U[2];
  const double rre = U[0] * U[3] - .5 * m2;
  const double R = rcp_refined(U[0] * rre);
  const double ir = R * rre;
  q[0] = U[0]; q[1] = U[1] * ir; q[2] = U[2] * ir;
  q[3] = (U[0] * U[0]) * (U[0] * R) * (1.0 / (2 * GM1));
}

// Phase 0 only (kt2_project), round 5: the same two maps with gamma = 1.4's exact exponents instead of three logarithms.
//   s = log(p / rho^gamma) = -0.4 log rho - log beta - log 2 = -log(32 beta^5 rho^2) / 5       ONE logarithm per volume node
//   rhoe(v) = ((gamma-1) / (-v4)^gamma)^(1/(gamma-1)) exp(-s/(gamma-1)) = 0.4^2.5 (-v4)^(-3.5) exp(-2.5 s)   (-v4)^(-3.5) = r^7, r = rsqrt(-v4)
// -- the reference's own form, a power times an exponential (euler_variables.jl:107-120 / cavity :473-478), with the power taken
// by a refined v_rsq_f64 and three multiplications instead of a logarithm inside the exponent; u = -v_i / v4 and beta = -v4 / (2 (gamma-1))
// in closed form from r^2 = 1 / (-v4).  345 -> 292 VALU per wave of kt2_project; last bits of the trace records change (every factor is
// accurate to an ulp or two, as before).
__device__ __forceinline__ double rsqrt_refined(double x) {
  const double y = __builtin_amdgcn_rsq(x);
  // two coupled Newton steps (as sqrt_fast): g -> sqrt(x), h -> 1 / (2 sqrt(x))
  double g = x * y, h = .5 * y;
  double r = __builtin_fma(-h, g, .5);
  g = __builtin_fma(g, r, g);
  h = __builtin_fma(h, r, h);
  r = __builtin_fma(-h, g, .5);
  h = __builtin_fma(h, r, h);
  return h + h;
}
template <bool MODAL>
__device__ __forceinline__ void v_of_state_onelog(const double* U, double* V) {
  constexpr double GM1 = Gas2<MODAL>::GM1;
  double q[4];
  prims<MODAL>(U, q);
  const double b2 = q[3] * q[3], r2 = q[0] * q[0];
  const double s = -0.2 * log_pos(32.0 * (b2 * b2) * (q[3] * r2));
  const double bb = 2 * GM1 * q[3];
  V[0] = 1.4 - s - .5 * bb * (q[1] * q[1] + q[2] * q[2]);
  V[1] = bb * q[1];
  V[2] = bb * q[2];
  V[3] = -bb;
}
template <bool MODAL>
__device__ __forceinline__ void prim_of_v2_fast(const double* V, double* q) {
  constexpr double GM1 = Gas2<MODAL>::GM1;
  const double mv = -V[3];
  const double r = rsqrt_refined(mv), r2 = r * r;          // r2 = 1 / (-v4)
  const double h = -.5 * (V[1] * V[1] + V[2] * V[2]) * r2;  // |vU|^2 / (2 v4)
  const double s = 1.4 - V[0] + h;
  const double r4 = r2 * r2;
  const double rhoeV = 0.10119288512538815 * ((r4 * r2) * r) * exp(-2.5 * s);   // 0.4^2.5 (-v4)^(-3.5) exp(-2.5 s)
  q[0] = rhoeV * mv;
  q[1] = V[1] * r2;
  q[2] = V[2] * r2;
  q[3] = mv * (1.0 / (2 * GM1));
}

}  // namespace t2
}  // namespace esdg
