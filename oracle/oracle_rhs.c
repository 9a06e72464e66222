/* ORACLE (test infrastructure, NOT product code).
 *
 * Plain-C, loop-structured CPU restatement of the reference's explicit-RK right-hand-side
 * evaluations (yiminllin/ESDG-CNS, pure Julia; Julia is not available in this pipeline, so
 * this file -- cross-checked against the independent numpy restatement in
 * oracle/ref_rhs_numpy.py and pinned by the reference's own property tests
 * examples/EntropyStableEuler.jl/test/runtests.jl -- is the parity oracle and, timed, the
 * "port" CPU baseline of bench.py).  Only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg may load it.  The product (esdg_cns_amd/) never does.
 *
 * Parity status: pointwise physics pinned by runtests.jl inputs with closed-form expected
 * values (tests/test_oracle_physics.py); full-RHS parity is NOT pinned by stored reference
 * numbers (the reference stores none) -- it is pinned by two independent
 * restatements agreeing to round-off plus the invariants of SURVEY.md section 8(c).
 *
 * Layout: a Julia (n x K) column-major matrix is the C array x[e*n + i]; one array per field,
 * stacked field-major: X[f*K*n + e*n + i].  Operators are dense row-major A[i*ncols + j].
 * mapP is int64, 1-based, linear into (Nfq x K) exactly as the Julia driver holds it.
 *
 * Build: gcc -O2 -ffp-contract=off -fno-fast-math (see oracle/Makefile); -fopenmp optional
 * (element loops only; thread count set by oracle_set_threads, default 1 = the reference).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define GAMMA 1.4 /* examples/EntropyStableEuler/EntropyStableEuler.jl:9 */

/* Working precision.  Default: double = the reference's Float64 arithmetic, statement by statement.
 * -DORACLE_QUAD (oracle/Makefile target liboracle_quad.so): the SAME statements evaluated in IEEE binary128
 * (__float128 + libquadmath) on the same double inputs, operators and double-valued literals, rounded to double
 * once at the end -- the "truth" evaluator the parity tests use to separate the rounding error of a faithful
 * Float64 implementation (e_orc = |oracle_f64 - truth|) from the device's (e_gpu = |gpu - truth|).
 * Inputs (const double*) are never converted in place: C promotes double * real to real. */
#ifdef ORACLE_QUAD
#include <quadmath.h>
typedef __float128 real;
#define R_(f) f##q
#else
typedef double real;
#define R_(f) f
#endif

static real* to_real(const double* x, size_t n) {
  real* r = (real*)malloc(n * sizeof(real));
  for (size_t i = 0; i < n; ++i) r[i] = x[i];
  return r;
}
static void to_double(const real* x, size_t n, double* d) {
  for (size_t i = 0; i < n; ++i) d[i] = (double)x[i];
}
int oracle_real_bits(void) { return (int)(8 * sizeof(real)); }

static int g_threads = 1;
void oracle_set_threads(int n) { g_threads = n < 1 ? 1 : n; }
int oracle_get_max_threads(void) {
#ifdef _OPENMP
  return omp_get_max_threads();
#else
  return 1;
#endif
}

/* ------------------------------------------------------------------------------------------
 * Pointwise physics: examples/EntropyStableEuler/{logmean,euler_fluxes,euler_variables}.jl
 * ---------------------------------------------------------------------------------------- */
/* logmean.jl:14-28 */
static real logmean_r(real aL, real aR, real logL, real logR) {
  real da = aR - aL;
  real aavg = .5 * (aR + aL);
  real f = da / aavg;
  real v = f * f;
  if (R_(fabs)(f) < 1e-4)
    return aavg * (1 + v * (-.2 - v * (.0512 - v * 0.026038857142857)));
  return -da / (logL - logR);
}
double oracle_logmean(double aL, double aR, double logL, double logR) { return (double)logmean_r(aL, aR, logL, logR); }

/* euler_fluxes.jl:23-48; UL/UR = (rho,u,v,beta), logs = (log rho, log beta) */
static void euler_fluxes_2d_r(const real* UL, const real* UR, const real* logL, const real* logR, real* Fx,
                             real* Fy) {
  real rhoL = UL[0], uL = UL[1], vL = UL[2], betaL = UL[3];
  real rhoR = UR[0], uR = UR[1], vR = UR[2], betaR = UR[3];
  real rholog = logmean_r(rhoL, rhoR, logL[0], logR[0]);
  real betalog = logmean_r(betaL, betaR, logL[1], logR[1]);
  real rhoavg = .5 * (rhoL + rhoR);
  real uavg = .5 * (uL + uR);
  real vavg = .5 * (vL + vR);
  real unorm = uL * uR + vL * vR;
  real pa = rhoavg / (betaL + betaR);
  real f4aux = rholog / (2 * (GAMMA - 1) * betalog) + pa + .5 * rholog * unorm;
  Fx[0] = rholog * uavg;
  Fx[1] = Fx[0] * uavg + pa;
  Fx[2] = Fx[0] * vavg;
  Fx[3] = f4aux * uavg;
  Fy[0] = rholog * vavg;
  Fy[1] = Fx[2];
  Fy[2] = Fy[0] * vavg + pa;
  Fy[3] = f4aux * vavg;
}
void oracle_euler_fluxes_2d(const double* UL, const double* UR, const double* logL, const double* logR, double* Fx,
                            double* Fy) {
  real a[4], b[4], la[2], lb[2], fx[4], fy[4];
  for (int i = 0; i < 4; ++i) { a[i] = UL[i]; b[i] = UR[i]; }
  for (int i = 0; i < 2; ++i) { la[i] = logL[i]; lb[i] = logR[i]; }
  euler_fluxes_2d_r(a, b, la, lb, fx, fy);
  to_double(fx, 4, Fx);
  to_double(fy, 4, Fy);
}

/* euler_variables.jl:79-92 (v_ufun via rhoefun :59-62, sfun :65-68) */
static void v_ufun_r(const real* U, real* V) {
  real rho = U[0], rhou = U[1], rhov = U[2], E = U[3];
  real rhoe = E - .5 * (rhou * rhou + rhov * rhov) / rho;
  real sU = R_(log)((GAMMA - 1) * rhoe / R_(pow)(rho, GAMMA));
  V[0] = (-E + rhoe * (GAMMA + 1 - sU)) / rhoe;
  V[1] = rhou / rhoe;
  V[2] = rhov / rhoe;
  V[3] = (-rho) / rhoe;
}
void oracle_v_ufun(const double* U, double* V) {
  real u[4], v[4];
  for (int i = 0; i < 4; ++i) u[i] = U[i];
  v_ufun_r(u, v);
  to_double(v, 4, V);
}

/* euler_variables.jl:95-120 (u_vfun via s_vfun, rhoe_vfun) */
static void u_vfun_r(const real* V, real* U) {
  real v1 = V[0], v2 = V[1], v3 = V[2], v4 = V[3];
  real vUnorm = v2 * v2 + v3 * v3;
  real s = GAMMA - v1 + vUnorm / (2 * v4);
  real rhoeV = R_(pow)((GAMMA - 1) / R_(pow)(-v4, GAMMA), 1 / (GAMMA - 1)) * R_(exp)(-s / (GAMMA - 1));
  U[0] = rhoeV * (-v4);
  U[1] = rhoeV * v2;
  U[2] = rhoeV * v3;
  U[3] = rhoeV * (1 - vUnorm / (2 * v4));
}
void oracle_u_vfun(const double* V, double* U) {
  real v[4], u[4];
  for (int i = 0; i < 4; ++i) v[i] = V[i];
  u_vfun_r(v, u);
  to_double(u, 4, U);
}

/* euler_variables.jl:30-48 */
static real betafun_r(const real* U) {
  real rhounorm = (U[1] * U[1] + U[2] * U[2]) / U[0];
  real p = (GAMMA - 1) * (U[3] - .5 * rhounorm);
  return U[0] / (2 * p);
}
double oracle_betafun(const double* U) {
  real u[4];
  for (int i = 0; i < 4; ++i) u[i] = U[i];
  return (double)betafun_r(u);
}

/* euler_variables.jl:7-10 -- note sqrt(abs(u_n)) (quirk Q1) */
static real wavespeed_r(real rho, real rhou, real E) {
  real p = (GAMMA - 1) * (E - .5 * (rhou * rhou) / rho);
  real cvel = R_(sqrt)(GAMMA * p / rho);
  return R_(sqrt)(R_(fabs)(rhou / rho)) + cvel;
}
double oracle_wavespeed(double rho, double rhou, double E) { return (double)wavespeed_r(rho, rhou, E); }

/* ------------------------------------------------------------------------------------------
 * small dense helpers:  Y(n x K) = A(n x m) * X(m x K), per element
 * ---------------------------------------------------------------------------------------- */
static void matmul_elems(const double* A, int n, int m, const real* X, real* Y, int K) {
#pragma omp parallel for num_threads(g_threads) schedule(static)
  for (int e = 0; e < K; ++e)
    for (int i = 0; i < n; ++i) {
      real s = 0.0;
      for (int j = 0; j < m; ++j) s += A[i * m + j] * X[(size_t)e * m + j];
      Y[(size_t)e * n + i] = s;
    }
}

/* ------------------------------------------------------------------------------------------
 * Euler, collocated quad:  examples/dg2D_euler_quad.jl:102-194
 * ---------------------------------------------------------------------------------------- */
/* sparse_hadamard_sum, dg2D_euler_quad.jl:102-138: row-wise, nonzero column ids per row */
static void sparse_hadamard_sum(const real* Qhe /*[4][Nh]*/, int Nh, const double* Qr, const double* Qs,
                                const int* rowptr, const int* colidx, real rxJ, real sxJ, real ryJ,
                                real syJ, real* out /*[4][Nh]*/) {
  real lg[2 * 64 * 4];
  real* lrho = lg;
  real* lbeta = lg + Nh;
  for (int i = 0; i < Nh; ++i) {
    lrho[i] = R_(log)(Qhe[0 * Nh + i]);
    lbeta[i] = R_(log)(Qhe[3 * Nh + i]);
  }
  for (int i = 0; i < Nh; ++i) {
    real Qi[4] = {Qhe[i], Qhe[Nh + i], Qhe[2 * Nh + i], Qhe[3 * Nh + i]};
    real li[2] = {lrho[i], lbeta[i]};
    real rhsi[4] = {0, 0, 0, 0};
    for (int t = rowptr[i]; t < rowptr[i + 1]; ++t) {
      int j = colidx[t];
      real Qj[4] = {Qhe[j], Qhe[Nh + j], Qhe[2 * Nh + j], Qhe[3 * Nh + j]};
      real lj[2] = {lrho[j], lbeta[j]};
      real Fx[4], Fy[4];
      euler_fluxes_2d_r(Qi, Qj, li, lj, Fx, Fy);
      for (int f = 0; f < 4; ++f) {
        real Fr = rxJ * Fx[f] + ryJ * Fy[f];
        real Fs = sxJ * Fx[f] + syJ * Fy[f];
        rhsi[f] += Qr[i * Nh + j] * Fr + Qs[i * Nh + j] * Fs;
      }
    }
    for (int f = 0; f < 4; ++f) out[f * Nh + i] = rhsi[f];
  }
}

/* rhs, dg2D_euler_quad.jl:141-194.
 * Q, rhs: [4][K][Nq]; Ef: Nfq x Nq; Qr,Qs: Nh x Nh (droptol'd); rowptr/colidx: Qrsids (0-based);
 * Ph: Nq x Nh; Lf: Nq x Nfq; rxJ..syJ: [K][Nh] (Vh-interpolated); J: [K][Nq]; wJq: [K][Nq];
 * nxJ,nyJ,sJ: [K][Nfq]; mapP: [K][Nfq] 1-based.  lf_scale = .5 in the reference (:165).
 * Returns rhstest (0 unless compute_rhstest). */
static real euler_rhs_r(int K, int Nq, int Nfq, const real* Q, const double* Ef, const double* Qr,
                        const double* Qs, const int* rowptr, const int* colidx, const double* Ph,
                        const double* Lf, const double* rxJ, const double* sxJ, const double* ryJ,
                        const double* syJ, const double* J, const double* wJq, const double* nxJ,
                        const double* nyJ, const double* sJ, const int64_t* mapP, double lf_scale,
                        int compute_rhstest, real* rhs) {
  const int Nh = Nq + Nfq;
  const size_t KNq = (size_t)K * Nq, KNf = (size_t)K * Nfq, KNh = (size_t)K * Nh;
  real* VU = (real*)malloc(4 * KNq * sizeof(real));
  real* VUf = (real*)malloc(4 * KNf * sizeof(real));
  real* Uf = (real*)malloc(4 * KNf * sizeof(real));
  real* Qh = (real*)malloc(4 * KNh * sizeof(real));
  real* lam = (real*)malloc(KNf * sizeof(real));
  real* flux = (real*)malloc(4 * KNf * sizeof(real));

  /* :149 VU = v_ufun(Q...) */
#pragma omp parallel for num_threads(g_threads) schedule(static)
  for (size_t n = 0; n < KNq; ++n) {
    real U[4] = {Q[n], Q[KNq + n], Q[2 * KNq + n], Q[3 * KNq + n]}, V[4];
    v_ufun_r(U, V);
    for (int f = 0; f < 4; ++f) VU[f * KNq + n] = V[f];
  }
  /* :150 Uf = u_vfun(Ef*VU) */
  for (int f = 0; f < 4; ++f) matmul_elems(Ef, Nfq, Nq, VU + f * KNq, VUf + f * KNf, K);
#pragma omp parallel for num_threads(g_threads) schedule(static)
  for (size_t n = 0; n < KNf; ++n) {
    real V[4] = {VUf[n], VUf[KNf + n], VUf[2 * KNf + n], VUf[3 * KNf + n]}, U[4];
    u_vfun_r(V, U);
    for (int f = 0; f < 4; ++f) Uf[f * KNf + n] = U[f];
  }
  /* :151-155 (rho,rhou,rhov,E) = vcat(Q,Uf); beta; Qh = (rho,u,v,beta) */
#pragma omp parallel for num_threads(g_threads) schedule(static)
  for (int e = 0; e < K; ++e)
    for (int i = 0; i < Nh; ++i) {
      real U[4];
      for (int f = 0; f < 4; ++f)
        U[f] = i < Nq ? Q[f * KNq + (size_t)e * Nq + i] : Uf[f * KNf + (size_t)e * Nfq + (i - Nq)];
      real beta = betafun_r(U);
      size_t o = (size_t)e * Nh + i;
      Qh[o] = U[0];
      Qh[KNh + o] = U[1] / U[0];
      Qh[2 * KNh + o] = U[2] / U[0];
      Qh[3 * KNh + o] = beta;
    }
  /* :162-164 lam */
#pragma omp parallel for num_threads(g_threads) schedule(static)
  for (size_t n = 0; n < KNf; ++n) {
    real rhoUM_n = (Uf[KNf + n] * nxJ[n] + Uf[2 * KNf + n] * nyJ[n]) / sJ[n];
    lam[n] = R_(fabs)(wavespeed_r(Uf[n], rhoUM_n, Uf[3 * KNf + n]));
  }
  /* :158-170 QM/QP, LFc, surface flux */
#pragma omp parallel for num_threads(g_threads) schedule(static)
  for (int e = 0; e < K; ++e)
    for (int i = 0; i < Nfq; ++i) {
      size_t n = (size_t)e * Nfq + i;
      size_t p = (size_t)(mapP[n] - 1);
      size_t ep = p / Nfq, ip = p % Nfq;
      size_t om = (size_t)e * Nh + Nq + i, op = ep * Nh + Nq + ip;
      real QM[4] = {Qh[om], Qh[KNh + om], Qh[2 * KNh + om], Qh[3 * KNh + om]};
      real QP[4] = {Qh[op], Qh[KNh + op], Qh[2 * KNh + op], Qh[3 * KNh + op]};
      real lM[2] = {R_(log)(QM[0]), R_(log)(QM[3])}, lP[2] = {R_(log)(QP[0]), R_(log)(QP[3])};
      real Fx[4], Fy[4];
      euler_fluxes_2d_r(QM, QP, lM, lP, Fx, Fy);
      real LFc = lf_scale * R_(fmax)(lam[n], lam[p]) * sJ[n];
      for (int f = 0; f < 4; ++f)
        flux[f * KNf + n] = Fx[f] * nxJ[n] + Fy[f] * nyJ[n] - LFc * (Uf[f * KNf + p] - Uf[f * KNf + n]);
    }
  /* :170 rhsQ = Lf*flux */
  for (int f = 0; f < 4; ++f) matmul_elems(Lf, Nq, Nfq, flux + f * KNf, rhs + f * KNq, K);
  /* :173-182 volume loop */
#pragma omp parallel for num_threads(g_threads) schedule(static)
  for (int e = 0; e < K; ++e) {
    real Qhe[4 * 64 * 4], QFe[4 * 64 * 4];
    for (int f = 0; f < 4; ++f)
      for (int i = 0; i < Nh; ++i) Qhe[f * Nh + i] = Qh[f * KNh + (size_t)e * Nh + i];
    size_t g = (size_t)e * Nh; /* vgeo_local = (rxJ,sxJ,ryJ,syJ)[1,e] */
    sparse_hadamard_sum(Qhe, Nh, Qr, Qs, rowptr, colidx, rxJ[g], sxJ[g], ryJ[g], syJ[g], QFe);
    for (int f = 0; f < 4; ++f)
      for (int i = 0; i < Nq; ++i) {
        real s = 0.0;
        for (int j = 0; j < Nh; ++j) s += Ph[i * Nh + j] * QFe[f * Nh + j];
        rhs[f * KNq + (size_t)e * Nq + i] += 2 * s;
      }
  }
  /* :184 */
  for (int f = 0; f < 4; ++f)
    for (size_t n = 0; n < KNq; ++n) rhs[f * KNq + n] = -rhs[f * KNq + n] / J[n];
  /* :186-191 */
  real rhstest = 0.0;
  if (compute_rhstest)
    for (int f = 0; f < 4; ++f)
      for (size_t n = 0; n < KNq; ++n) rhstest += wJq[n] * VU[f * KNq + n] * rhs[f * KNq + n];
  free(VU); free(VUf); free(Uf); free(Qh); free(lam); free(flux);
  return rhstest;
}
double oracle_euler_rhs(int K, int Nq, int Nfq, const double* Q, const double* Ef, const double* Qr,
                        const double* Qs, const int* rowptr, const int* colidx, const double* Ph,
                        const double* Lf, const double* rxJ, const double* sxJ, const double* ryJ,
                        const double* syJ, const double* J, const double* wJq, const double* nxJ,
                        const double* nyJ, const double* sJ, const int64_t* mapP, double lf_scale,
                        int compute_rhstest, double* rhs) {
  const size_t n = 4 * (size_t)K * Nq;
  real* Qr_ = to_real(Q, n);
  real* out = (real*)malloc(n * sizeof(real));
  real rt = euler_rhs_r(K, Nq, Nfq, Qr_, Ef, Qr, Qs, rowptr, colidx, Ph, Lf, rxJ, sxJ, ryJ, syJ, J, wJq, nxJ, nyJ, sJ,
                        mapP, lf_scale, compute_rhstest, out);
  to_double(out, n, rhs);
  free(Qr_); free(out);
  return (double)rt;
}

/* ------------------------------------------------------------------------------------------
 * CNS, modal ESDG: examples/CompressibleNS/dg2D_CNS_cavity_optimized.jl
 * ---------------------------------------------------------------------------------------- */
typedef struct {
  int K, Np, Nq, Nfq;
  /* operators (dense row-major) */
  const double *Vq /*Nq x Np*/, *Pq /*Np x Nq*/, *Vf /*Nfq x Np*/, *LIFT /*Np x Nfq*/;
  const double *Dr, *Ds /*Np x Np*/, *VhP /*Nh x Nq*/, *Ph /*Np x Nh*/;
  const double *Qrh, *Qsh /*Nh x Nh skew*/;
  /* mesh */
  const double *rxJ, *sxJ, *ryJ, *syJ /*[K][Nh]*/, *J /*[K][Np]*/, *wJq /*[K][Nq]*/;
  const double *nxJ, *nyJ, *sJ /*[K][Nfq]*/;
  const int64_t* mapP /*[K][Nfq] 1-based*/;
  /* boundary: nodes of md.mapB (1-based linear), kind 0 = wall, 1 = lid (init_BC_funs :135-155) */
  int Nb;
  const int64_t* mapB;
  const int32_t* bkind;
  int BCTYPE;
  /* physics */
  double Re, lambda, mu, Pr;
  int inviscid_dissp, viscous_dissp;
  /* BCTYPE 4 = the boundary closures of the shock-tube driver (dg2D_CNS_modalESDG.jl:161-217): bkind 1 = Dirichlet
   * inflow with the state (rho,u,v,p) below, bkind 0 = copy of the interior value; lam = lamP = 0 on both; no penalty */
  double inflow[4];
  /* lid velocity per mapB entry (read where bkind = 1): ones in cavity_optimized.jl:147 (NULL here),
   * (1+cos(pi*xlid))/2 in dg2D_CNS_convergence_test.jl:76 */
  const double* vlid;
} oracle_cns_t;

/* dg2D_CNS_cavity_optimized.jl:461-467: hard-coded gamma literals (quirk Q5) */
static void v_hardcoded(const real* U, real* V) {
  real n = U[1] * U[1] + U[2] * U[2];
  real rhoe = U[3] - .5 * n / U[0];
  real sU = R_(log)(0.4 * rhoe / R_(pow)(U[0], 1.4));
  V[0] = (-U[3] + rhoe * (2.4 - sU)) / rhoe;
  V[1] = U[1] / rhoe;
  V[2] = U[2] / rhoe;
  V[3] = -U[0] / rhoe;
}

/* rhs_inviscid!, :447-528 with update_flux! :308-324 and flux_differencing! :326-348.
 * Q, rhs: [4][K][Np]. */
static void cns_rhs_inviscid_r(const oracle_cns_t* c, const real* Q, real* rhs) {
  const int K = c->K, Np = c->Np, Nq = c->Nq, Nfq = c->Nfq, Nh = Nq + Nfq;
  const size_t KNp = (size_t)K * Np, KNq = (size_t)K * Nq, KNf = (size_t)K * Nfq, KNh = (size_t)K * Nh;
  real* Qq = (real*)malloc(4 * KNq * sizeof(real));
  real* VU = (real*)malloc(4 * KNq * sizeof(real));
  real* Uh = (real*)malloc(4 * KNh * sizeof(real));
  real* Qh = (real*)malloc(4 * KNh * sizeof(real));
  real* QP = (real*)malloc(4 * KNf * sizeof(real));
  real* lam = (real*)malloc(KNf * sizeof(real));
  real* flux = (real*)malloc(4 * KNf * sizeof(real));
  real* QF = (real*)calloc(4 * KNh, sizeof(real));
  real* tmpN = (real*)malloc(4 * KNp * sizeof(real));

  for (int f = 0; f < 4; ++f) matmul_elems(c->Vq, Nq, Np, Q + f * KNp, Qq + f * KNq, K); /* :459 */
#pragma omp parallel for num_threads(g_threads) schedule(static)
  for (size_t n = 0; n < KNq; ++n) { /* :461-467 */
    real U[4] = {Qq[n], Qq[KNq + n], Qq[2 * KNq + n], Qq[3 * KNq + n]}, V[4];
    v_hardcoded(U, V);
    for (int f = 0; f < 4; ++f) VU[f * KNq + n] = V[f];
  }
  for (int f = 0; f < 4; ++f) matmul_elems(c->VhP, Nh, Nq, VU + f * KNq, Uh + f * KNh, K); /* :470 */
#pragma omp parallel for num_threads(g_threads) schedule(static)
  for (size_t n = 0; n < KNh; ++n) { /* :473-488 */
    real v1 = Uh[n], v2 = Uh[KNh + n], v3 = Uh[2 * KNh + n], v4 = Uh[3 * KNh + n];
    real tmp = v2 * v2 + v3 * v3;
    real tmp2 = R_(pow)(0.4 / R_(pow)(-v4, 1.4), 1 / 0.4) * R_(exp)(-(1.4 - v1 + tmp / (2 * v4)) / 0.4);
    real u1 = tmp2 * (-v4), u2 = tmp2 * v2, u3 = tmp2 * v3, u4 = tmp2 * (1 - tmp / (2 * v4));
    Uh[n] = u1; Uh[KNh + n] = u2; Uh[2 * KNh + n] = u3; Uh[3 * KNh + n] = u4;
    real beta = u1 / (2 * 0.4 * (u4 - .5 * (u2 * u2 + u3 * u3) / u1));
    Qh[n] = u1; Qh[KNh + n] = u2 / u1; Qh[2 * KNh + n] = u3 / u1; Qh[3 * KNh + n] = beta;
  }
  /* :495-498 QM = face rows, QP = QM[mapP], impose_BCs_inviscid! (:157-176) */
  for (int e = 0; e < K; ++e)
    for (int i = 0; i < Nfq; ++i) {
      size_t n = (size_t)e * Nfq + i, p = (size_t)(c->mapP[n] - 1);
      size_t op = (p / Nfq) * Nh + Nq + (p % Nfq);
      for (int f = 0; f < 4; ++f) QP[f * KNf + n] = Qh[f * KNh + op];
    }
  unsigned char* nolf = (unsigned char*)calloc(KNf, 1); /* impose_BCs_lam! (modalESDG :180-185): lam = lamP = 0 */
  for (int b = 0; b < c->Nb; ++b) {
    size_t n = (size_t)(c->mapB[b] - 1);
    size_t om = (n / Nfq) * Nh + Nq + (n % Nfq);
    if (c->BCTYPE == 4) { /* impose_BCs_inviscid!, dg2D_CNS_modalESDG.jl:168-178 */
      nolf[n] = 1;
      if (c->bkind[b]) {
        QP[n] = c->inflow[0];
        QP[KNf + n] = c->inflow[1];
        QP[2 * KNf + n] = c->inflow[2];
        QP[3 * KNf + n] = c->inflow[0] / (2 * c->inflow[3]);
      } else {
        for (int f = 0; f < 4; ++f) QP[f * KNf + n] = Qh[f * KNh + om];
      }
      continue;
    }
    real nx = c->nxJ[n] / c->sJ[n], ny = c->nyJ[n] / c->sJ[n];
    real u1 = Qh[KNh + om], u2 = Qh[2 * KNh + om];
    real Un = u1 * nx + u2 * ny;
    QP[n] = Qh[om];
    QP[3 * KNf + n] = Qh[3 * KNh + om];
    QP[KNf + n] = u1 - 2 * Un * nx;
    QP[2 * KNf + n] = u2 - 2 * Un * ny;
  }
  /* :501-508 lam, LFc */
#pragma omp parallel for num_threads(g_threads) schedule(static)
  for (int e = 0; e < K; ++e)
    for (int i = 0; i < Nfq; ++i) {
      size_t n = (size_t)e * Nfq + i, om = (size_t)e * Nh + Nq + i;
      real rhoM = Uh[om], rhouM = Uh[KNh + om], rhovM = Uh[2 * KNh + om], EM = Uh[3 * KNh + om];
      real rhoUM_n = (rhouM * c->nxJ[n] + rhovM * c->nyJ[n]) / c->sJ[n];
      lam[n] = R_(fabs)(R_(sqrt)(R_(fabs)(rhoUM_n / rhoM)) + R_(sqrt)(1.4 * 0.4 * (EM - .5 * rhoUM_n * rhoUM_n / rhoM) / rhoM));
    }
  /* :510-514 update_flux! then LIFT */
#pragma omp parallel for num_threads(g_threads) schedule(static)
  for (int e = 0; e < K; ++e)
    for (int i = 0; i < Nfq; ++i) {
      size_t n = (size_t)e * Nfq + i, p = (size_t)(c->mapP[n] - 1);
      size_t om = (size_t)e * Nh + Nq + i, op = (p / Nfq) * Nh + Nq + (p % Nfq);
      real QMl[4] = {Qh[om], Qh[KNh + om], Qh[2 * KNh + om], Qh[3 * KNh + om]};
      real QPl[4] = {QP[n], QP[KNf + n], QP[2 * KNf + n], QP[3 * KNf + n]};
      real lM[2] = {R_(log)(QMl[0]), R_(log)(QMl[3])}, lP[2] = {R_(log)(QPl[0]), R_(log)(QPl[3])};
      real Fx[4], Fy[4];
      euler_fluxes_2d_r(QPl, QMl, lP, lM, Fx, Fy); /* (QP,QM) order, quirk Q8 */
      real LFc = nolf[n] ? 0.0 : .25 * R_(fmax)(lam[n], lam[p]) * c->sJ[n];
      for (int f = 0; f < 4; ++f) {
        real v = Fx[f] * c->nxJ[n] + Fy[f] * c->nyJ[n];
        if (c->inviscid_dissp) v -= LFc * (Uh[f * KNh + op] - Uh[f * KNh + om]);
        flux[f * KNf + n] = v;
      }
    }
  for (int f = 0; f < 4; ++f) matmul_elems(c->LIFT, Np, Nfq, flux + f * KNf, rhs + f * KNp, K);
  /* :516 flux_differencing! (symmetric; dense Qrh/Qsh; skip face x face; diagonal pairs evaluated) */
#pragma omp parallel for num_threads(g_threads) schedule(static)
  for (int k = 0; k < K; ++k) {
    size_t g = (size_t)k * Nh;
    real rx = c->rxJ[g], ry = c->ryJ[g], sx = c->sxJ[g], sy = c->syJ[g];
    for (int j = 0; j < Nh; ++j)
      for (int i = j; i < Nh; ++i)
        if (i < Nq || j < Nq) {
          real Qi[4] = {Qh[g + i], Qh[KNh + g + i], Qh[2 * KNh + g + i], Qh[3 * KNh + g + i]};
          real Qj[4] = {Qh[g + j], Qh[KNh + g + j], Qh[2 * KNh + g + j], Qh[3 * KNh + g + j]};
          real li[2] = {R_(log)(Qi[0]), R_(log)(Qi[3])}, lj[2] = {R_(log)(Qj[0]), R_(log)(Qj[3])};
          real Fx[4], Fy[4];
          euler_fluxes_2d_r(Qi, Qj, li, lj, Fx, Fy);
          real Qr = c->Qrh[i * Nh + j], Qs = c->Qsh[i * Nh + j];
          for (int d = 0; d < 4; ++d) {
            real val = 2 * ((rx * Qr + sx * Qs) * Fx[d] + (ry * Qr + sy * Qs) * Fy[d]);
            QF[d * KNh + g + i] += val;
            QF[d * KNh + g + j] -= val;
          }
        }
  }
  /* :517-518 rhsQ = -(Ph*QF + rhsQ)./J */
  for (int f = 0; f < 4; ++f) {
    matmul_elems(c->Ph, Np, Nh, QF + f * KNh, tmpN + f * KNp, K);
    for (size_t n = 0; n < KNp; ++n) rhs[f * KNp + n] = -(tmpN[f * KNp + n] + rhs[f * KNp + n]) / c->J[n];
  }
  free(Qq); free(VU); free(Uh); free(Qh); free(QP); free(lam); free(flux); free(QF); free(tmpN); free(nolf);
}
void oracle_cns_rhs_inviscid(const oracle_cns_t* c, const double* Q, double* rhs) {
  const size_t n = 4 * (size_t)c->K * c->Np;
  real* Qr_ = to_real(Q, n);
  real* out = (real*)malloc(n * sizeof(real));
  cns_rhs_inviscid_r(c, Qr_, out);
  to_double(out, n, rhs);
  free(Qr_); free(out);
}

/* viscous_matrices!, :613-645 (let lambda = -lambda, quirk Q4); entries not listed stay 0 */
static void viscous_matrices(real Kxx[4][4], real Kxy[4][4], real Kyy[4][4], const real* v,
                             double lambda_in, double mu, double Pr) {
  real lambda = -lambda_in;
  real v2 = v[1], v3 = v[2], v4 = v[3];
  real inv_v4_cubed = 1 / (v4 * v4 * v4);
  real l2m = lambda + 2.0 * mu;
  Kxx[1][1] = inv_v4_cubed * -l2m * (v4 * v4);
  Kxx[1][3] = inv_v4_cubed * l2m * v2 * v4;
  Kxx[2][2] = inv_v4_cubed * -mu * (v4 * v4);
  Kxx[2][3] = inv_v4_cubed * mu * v3 * v4;
  Kxx[3][1] = inv_v4_cubed * l2m * v2 * v4;
  Kxx[3][2] = inv_v4_cubed * mu * v3 * v4;
  Kxx[3][3] = inv_v4_cubed * -(l2m * (v2 * v2) + mu * (v3 * v3) - GAMMA * mu * v4 / Pr);
  Kxy[1][2] = inv_v4_cubed * -lambda * (v4 * v4);
  Kxy[1][3] = inv_v4_cubed * lambda * v3 * v4;
  Kxy[2][1] = inv_v4_cubed * -mu * (v4 * v4);
  Kxy[2][3] = inv_v4_cubed * mu * v2 * v4;
  Kxy[3][1] = inv_v4_cubed * mu * v3 * v4;
  Kxy[3][2] = inv_v4_cubed * lambda * v2 * v4;
  Kxy[3][3] = inv_v4_cubed * (lambda + mu) * (-v2 * v3);
  Kyy[1][1] = inv_v4_cubed * -mu * (v4 * v4);
  Kyy[1][3] = inv_v4_cubed * mu * v2 * v4;
  Kyy[2][2] = inv_v4_cubed * -l2m * (v4 * v4);
  Kyy[2][3] = inv_v4_cubed * l2m * v3 * v4;
  Kyy[3][1] = inv_v4_cubed * mu * v2 * v4;
  Kyy[3][2] = inv_v4_cubed * l2m * v3 * v4;
  Kyy[3][3] = inv_v4_cubed * -(l2m * (v3 * v3) + mu * (v2 * v2) - GAMMA * mu * v4 / Pr);
}

/* rhs_viscous!, :749-849 with dg_grad! :548-569 and dg_div! :590-611.  Returns rhstest (visc_test). */
static real cns_rhs_viscous_r(const oracle_cns_t* c, const real* Q, real* rhs) {
  const int K = c->K, Np = c->Np, Nq = c->Nq, Nfq = c->Nfq, Nh = Nq + Nfq;
  const size_t KNp = (size_t)K * Np, KNq = (size_t)K * Nq, KNf = (size_t)K * Nfq;
#define ALLOC(n) (real*)calloc((n), sizeof(real))
  real *Qq = ALLOC(4 * KNq), *VUq0 = ALLOC(4 * KNq), *VU = ALLOC(4 * KNp), *VUf = ALLOC(4 * KNf),
         *VUP = ALLOC(4 * KNf), *VUx = ALLOC(4 * KNp), *VUy = ALLOC(4 * KNp), *VUxq = ALLOC(4 * KNq),
         *VUyq = ALLOC(4 * KNq), *VUq = ALLOC(4 * KNq), *sxq = ALLOC(4 * KNq), *syq = ALLOC(4 * KNq),
         *sx = ALLOC(4 * KNp), *sy = ALLOC(4 * KNp), *sxf = ALLOC(4 * KNf), *syf = ALLOC(4 * KNf),
         *sxP = ALLOC(4 * KNf), *syP = ALLOC(4 * KNf), *pen = ALLOC(4 * KNf), *penL = ALLOC(4 * KNp),
         *t1 = ALLOC(KNp), *t2 = ALLOC(KNp), *t3 = ALLOC(KNp), *t4 = ALLOC(KNp), *tf = ALLOC(KNf),
         *tl = ALLOC(KNp);
  /* :763-772 */
  for (int f = 0; f < 4; ++f) matmul_elems(c->Vq, Nq, Np, Q + f * KNp, Qq + f * KNq, K);
#pragma omp parallel for num_threads(g_threads) schedule(static)
  for (size_t n = 0; n < KNq; ++n) {
    real U[4] = {Qq[n], Qq[KNq + n], Qq[2 * KNq + n], Qq[3 * KNq + n]}, V[4];
    v_hardcoded(U, V);
    for (int f = 0; f < 4; ++f) VUq0[f * KNq + n] = V[f];
  }
  for (int f = 0; f < 4; ++f) matmul_elems(c->Pq, Np, Nq, VUq0 + f * KNq, VU + f * KNp, K);
  /* :775-777 */
  for (int f = 0; f < 4; ++f) matmul_elems(c->Vf, Nfq, Np, VU + f * KNp, VUf + f * KNf, K);
  for (int f = 0; f < 4; ++f)
    for (size_t n = 0; n < KNf; ++n) VUP[f * KNf + n] = VUf[f * KNf + (size_t)(c->mapP[n] - 1)];
  /* impose_BCs_entropyvars! :178-216 */
  for (int b = 0; b < c->Nb; ++b) {
    size_t n = (size_t)(c->mapB[b] - 1);
    int lid = c->bkind[b];
    real vf2 = VUf[KNf + n], vf3 = VUf[2 * KNf + n], vf4 = VUf[3 * KNf + n];
    if (c->BCTYPE == 1) {
      VUP[KNf + n] = lid ? -vf2 - 2 * (c->vlid ? c->vlid[b] : 1.0) * vf4 : -vf2;
      VUP[2 * KNf + n] = -vf3;
      VUP[3 * KNf + n] = vf4;
    } else if (c->BCTYPE == 2) {
      real theta = 1.0 / (0.3 * 0.3) / 1.4 / 0.4;
      VUP[KNf + n] = lid ? 2.0 / theta - vf2 : -vf2;
      VUP[2 * KNf + n] = -vf3;
      VUP[3 * KNf + n] = -2.0 / theta - vf4;
    } else if (c->BCTYPE == 3) {
      real nx = c->nxJ[n] / c->sJ[n], ny = c->nyJ[n] / c->sJ[n];
      real VUn = vf2 * nx + vf3 * ny;
      VUP[3 * KNf + n] = vf4;
      VUP[KNf + n] = vf2 - 2 * VUn * nx;
      VUP[2 * KNf + n] = vf3 - 2 * VUn * ny;
    } else if (c->BCTYPE == 4) { /* dg2D_CNS_modalESDG.jl:187-203: VL = v_ufun(rhoL, rhoL*uL, rhoL*vL, EL) / VUf */
      if (lid) {
        real rho = c->inflow[0], u = c->inflow[1], v = c->inflow[2], p = c->inflow[3];
        real U[4] = {rho, rho * u, rho * v, p / (GAMMA - 1) + .5 * rho * (u * u + v * v)}, VL[4];
        v_ufun_r(U, VL);
        for (int f = 0; f < 4; ++f) VUP[f * KNf + n] = VL[f];
      } else {
        for (int f = 0; f < 4; ++f) VUP[f * KNf + n] = VUf[f * KNf + n];
      }
    }
  }
  /* dg_grad! :548-569 */
  for (int f = 0; f < 4; ++f) {
    matmul_elems(c->Dr, Np, Np, VU + f * KNp, t1, K);
    matmul_elems(c->Ds, Np, Np, VU + f * KNp, t2, K);
    for (size_t n = 0; n < KNf; ++n) tf[n] = .5 * (VUP[f * KNf + n] - VUf[f * KNf + n]) * c->nxJ[n];
    matmul_elems(c->LIFT, Np, Nfq, tf, t3, K);
    for (size_t n = 0; n < KNf; ++n) tf[n] = .5 * (VUP[f * KNf + n] - VUf[f * KNf + n]) * c->nyJ[n];
    matmul_elems(c->LIFT, Np, Nfq, tf, t4, K);
    for (int e = 0; e < K; ++e)
      for (int i = 0; i < Np; ++i) {
        size_t n = (size_t)e * Np + i, g = (size_t)e * Nh + i; /* rxj = rxJ[1:Np,:] */
        VUx[f * KNp + n] = ((c->rxJ[g] * t1[n] + c->sxJ[g] * t2[n]) + t3[n]) / c->J[n];
        VUy[f * KNp + n] = ((c->ryJ[g] * t1[n] + c->syJ[g] * t2[n]) + t4[n]) / c->J[n];
      }
  }
  /* :780-782 */
  for (int f = 0; f < 4; ++f) {
    matmul_elems(c->Vq, Nq, Np, VUx + f * KNp, VUxq + f * KNq, K);
    matmul_elems(c->Vq, Nq, Np, VUy + f * KNp, VUyq + f * KNq, K);
    matmul_elems(c->Vq, Nq, Np, VU + f * KNp, VUq + f * KNq, K);
  }
  /* :785-801 sigma loop */
#pragma omp parallel for num_threads(g_threads) schedule(static)
  for (int e = 0; e < K; ++e) {
    real Kxx[4][4] = {{0}}, Kxy[4][4] = {{0}}, Kyy[4][4] = {{0}};
    for (int i = 0; i < Nq; ++i) {
      size_t n = (size_t)e * Nq + i;
      real vqi[4] = {VUq[n], VUq[KNq + n], VUq[2 * KNq + n], VUq[3 * KNq + n]};
      viscous_matrices(Kxx, Kxy, Kyy, vqi, c->lambda, c->mu, c->Pr);
      for (int col = 1; col < 4; ++col) {
        real vxi = VUxq[col * KNq + n], vyi = VUyq[col * KNq + n];
        for (int row = 1; row < 4; ++row) {
          sxq[row * KNq + n] += Kxx[row][col] * vxi + Kxy[row][col] * vyi;
          syq[row * KNq + n] += Kxy[col][row] * vxi + Kyy[row][col] * vyi;
        }
      }
    }
  }
  /* :803-807 */
  real rhstest = 0.0;
  for (int f = 0; f < 4; ++f) {
    real a = 0.0, b = 0.0;
    for (size_t n = 0; n < KNq; ++n) a += c->wJq[n] * VUxq[f * KNq + n] * sxq[f * KNq + n];
    for (size_t n = 0; n < KNq; ++n) b += c->wJq[n] * VUyq[f * KNq + n] * syq[f * KNq + n];
    rhstest += a;
    rhstest += b;
  }
  /* :808-815 */
  for (int f = 0; f < 4; ++f) {
    matmul_elems(c->Pq, Np, Nq, sxq + f * KNq, sx + f * KNp, K);
    matmul_elems(c->Pq, Np, Nq, syq + f * KNq, sy + f * KNp, K);
    matmul_elems(c->Vf, Nfq, Np, sx + f * KNp, sxf + f * KNf, K);
    matmul_elems(c->Vf, Nfq, Np, sy + f * KNp, syf + f * KNf, K);
    for (size_t n = 0; n < KNf; ++n) {
      size_t p = (size_t)(c->mapP[n] - 1);
      sxP[f * KNf + n] = sxf[f * KNf + p];
      syP[f * KNf + n] = syf[f * KNf + p];
    }
  }
  /* impose_BCs_stress! :218-262 */
  for (int b = 0; b < c->Nb; ++b) {
    size_t n = (size_t)(c->mapB[b] - 1);
    int lid = c->bkind[b];
    if (c->BCTYPE == 1) {
      for (int f = 1; f <= 2; ++f) { sxP[f * KNf + n] = sxf[f * KNf + n]; syP[f * KNf + n] = syf[f * KNf + n]; }
      if (lid) {
        const real vl = c->vlid ? c->vlid[b] : 1.0;
        sxP[3 * KNf + n] = -sxf[3 * KNf + n] + 2 * vl * sxf[KNf + n];
        syP[3 * KNf + n] = -syf[3 * KNf + n] + 2 * vl * syf[KNf + n];
      } else {
        sxP[3 * KNf + n] = -sxf[3 * KNf + n];
        syP[3 * KNf + n] = -syf[3 * KNf + n];
      }
    } else if (c->BCTYPE == 2) {
      for (int f = 1; f <= 3; ++f) { sxP[f * KNf + n] = sxf[f * KNf + n]; syP[f * KNf + n] = syf[f * KNf + n]; }
    } else if (c->BCTYPE == 3) {
      real n1 = c->nxJ[n] / c->sJ[n], n2 = c->nyJ[n] / c->sJ[n];
      real sx1 = sxf[KNf + n], sx2 = sxf[2 * KNf + n], sy1 = syf[KNf + n], sy2 = syf[2 * KNf + n];
      real snx = sx1 * n1 + sx2 * n2, sny = sy1 * n1 + sy2 * n2;
      sxP[KNf + n] = -sx1 + 2 * n1 * snx;
      syP[KNf + n] = -sy1 + 2 * n1 * sny;
      sxP[2 * KNf + n] = -sx2 + 2 * n2 * snx;
      syP[2 * KNf + n] = -sy2 + 2 * n2 * sny;
      sxP[3 * KNf + n] = -sxf[3 * KNf + n];
      syP[3 * KNf + n] = -syf[3 * KNf + n];
    } else if (c->BCTYPE == 4) { /* dg2D_CNS_modalESDG.jl:205-216: sigma+ = sigma- on both sides */
      for (int f = 0; f < 4; ++f) { sxP[f * KNf + n] = sxf[f * KNf + n]; syP[f * KNf + n] = syf[f * KNf + n]; }
    }
  }
  /* :817-840 penalty (the shock-tube driver has this block commented out, dg2D_CNS_modalESDG.jl:494-518) */
  if (c->viscous_dissp && c->BCTYPE != 4) {
    for (size_t n = 0; n < KNf; ++n) {
      real tau = -1 / c->Re / VUf[3 * KNf + n];
      for (int f = 1; f < 4; ++f) pen[f * KNf + n] = tau * (VUP[f * KNf + n] - VUf[f * KNf + n]);
    }
    for (int b = 0; b < c->Nb; ++b) {
      size_t n = (size_t)(c->mapB[b] - 1);
      real tau = -1 / c->Re / VUf[3 * KNf + n];
      real dV2 = VUP[KNf + n] - VUf[KNf + n], dV3 = VUP[2 * KNf + n] - VUf[2 * KNf + n],
             dV4 = VUP[3 * KNf + n] - VUf[3 * KNf + n];
      real a2 = 1.0 / 2 * (VUP[KNf + n] + VUf[KNf + n]), a3 = 1.0 / 2 * (VUP[2 * KNf + n] + VUf[2 * KNf + n]);
      pen[KNf + n] = tau * dV2;
      pen[2 * KNf + n] = tau * dV3;
      if (c->BCTYPE == 1)
        pen[3 * KNf + n] = -tau * (a2 * dV2 + a3 * dV3) / VUf[3 * KNf + n];
      else
        pen[3 * KNf + n] = -tau * (a2 * dV2 + a3 * dV3 + dV4 * dV4 / 2) / VUf[3 * KNf + n];
    }
    for (int f = 0; f < 4; ++f) matmul_elems(c->LIFT, Np, Nfq, pen + f * KNf, penL + f * KNp, K);
  }
  /* dg_div! :590-611 */
  for (int f = 0; f < 4; ++f) {
    matmul_elems(c->Dr, Np, Np, sx + f * KNp, t1, K);
    matmul_elems(c->Ds, Np, Np, sx + f * KNp, t2, K);
    matmul_elems(c->Dr, Np, Np, sy + f * KNp, t3, K);
    matmul_elems(c->Ds, Np, Np, sy + f * KNp, t4, K);
    for (size_t n = 0; n < KNf; ++n)
      tf[n] = .5 * ((sxP[f * KNf + n] - sxf[f * KNf + n]) * c->nxJ[n] + (syP[f * KNf + n] - syf[f * KNf + n]) * c->nyJ[n]);
    matmul_elems(c->LIFT, Np, Nfq, tf, tl, K);
    for (int e = 0; e < K; ++e)
      for (int i = 0; i < Np; ++i) {
        size_t n = (size_t)e * Np + i, g = (size_t)e * Nh + i;
        real vol = c->rxJ[g] * t1[n] + c->sxJ[g] * t2[n] + c->ryJ[g] * t3[n] + c->syJ[g] * t4[n];
        real r = (vol + tl[n]) / c->J[n];
        if (c->viscous_dissp && c->BCTYPE != 4) r = r + penL[f * KNp + n];
        rhs[f * KNp + n] = r;
      }
  }
  free(Qq); free(VUq0); free(VU); free(VUf); free(VUP); free(VUx); free(VUy); free(VUxq); free(VUyq);
  free(VUq); free(sxq); free(syq); free(sx); free(sy); free(sxf); free(syf); free(sxP); free(syP);
  free(pen); free(penL); free(t1); free(t2); free(t3); free(t4); free(tf); free(tl);
  return rhstest;
}
double oracle_cns_rhs_viscous(const oracle_cns_t* c, const double* Q, double* rhs) {
  const size_t n = 4 * (size_t)c->K * c->Np;
  real* Qr_ = to_real(Q, n);
  real* out = (real*)malloc(n * sizeof(real));
  real rt = cns_rhs_viscous_r(c, Qr_, out);
  to_double(out, n, rhs);
  free(Qr_); free(out);
  return (double)rt;
}

/* rhsRK!, :955-972.  diag[0] = rhstest, diag[1] = rhstest_visc (computed when compute_diag). */
static void cns_rhsRK_r(const oracle_cns_t* c, const real* Q, real* rhs, int compute_diag, real* diag) {
  const int K = c->K, Np = c->Np, Nq = c->Nq;
  const size_t KNp = (size_t)K * Np, KNq = (size_t)K * Nq;
  real* visc = (real*)malloc(4 * KNp * sizeof(real));
  cns_rhs_inviscid_r(c, Q, rhs);
  real visc_test = cns_rhs_viscous_r(c, Q, visc);
  for (size_t n = 0; n < 4 * KNp; ++n) rhs[n] = rhs[n] + visc[n];
  if (compute_diag) {
    real *Qq = ALLOC(4 * KNq), *VU = ALLOC(4 * KNq), *VUn = ALLOC(4 * KNp), *VUq = ALLOC(4 * KNq),
           *rq = ALLOC(KNq), *vq = ALLOC(KNq);
    for (int f = 0; f < 4; ++f) matmul_elems(c->Vq, Nq, Np, Q + f * KNp, Qq + f * KNq, K);
    for (size_t n = 0; n < KNq; ++n) {
      real U[4] = {Qq[n], Qq[KNq + n], Qq[2 * KNq + n], Qq[3 * KNq + n]}, V[4];
      v_ufun_r(U, V);
      for (int f = 0; f < 4; ++f) VU[f * KNq + n] = V[f];
    }
    real rhstest = 0.0, rhstest_visc = 0.0;
    for (int f = 0; f < 4; ++f) {
      matmul_elems(c->Pq, Np, Nq, VU + f * KNq, VUn + f * KNp, K); /* VUq = Vq*Pq*VU */
      matmul_elems(c->Vq, Nq, Np, VUn + f * KNp, VUq + f * KNq, K);
      matmul_elems(c->Vq, Nq, Np, rhs + f * KNp, rq, K);
      matmul_elems(c->Vq, Nq, Np, visc + f * KNp, vq, K);
      real a = 0.0, b = 0.0;
      for (size_t n = 0; n < KNq; ++n) a += c->wJq[n] * VUq[f * KNq + n] * rq[n];
      for (size_t n = 0; n < KNq; ++n) b += c->wJq[n] * VUq[f * KNq + n] * vq[n];
      rhstest += a;
      rhstest_visc += b;
    }
    diag[0] = rhstest;
    diag[1] = rhstest_visc + visc_test;
    free(Qq); free(VU); free(VUn); free(VUq); free(rq); free(vq);
  }
  free(visc);
}
void oracle_cns_rhsRK(const oracle_cns_t* c, const double* Q, double* rhs, int compute_diag, double* diag) {
  const size_t n = 4 * (size_t)c->K * c->Np;
  real* Qr_ = to_real(Q, n);
  real* out = (real*)malloc(n * sizeof(real));
  real dg[2] = {0, 0};
  cns_rhsRK_r(c, Qr_, out, compute_diag, dg);
  to_double(out, n, rhs);
  if (compute_diag) to_double(dg, 2, diag);
  free(Qr_); free(out);
}

/* ------------------------------------------------------------------------------------------
 * Euler, collocated hex:  examples/dg3D_euler_hex.jl:122-222 (3D physics: euler_fluxes.jl:51-89,
 * euler_variables.jl with 3-component momentum).  The reference flags the driver broken (:1): the
 * breakage is in hex_face_vertices (src/UniformHexMesh.jl:83-93), i.e. in the set-up, not in `rhs`;
 * see oracle/ref_setup.py.  lf_scale replaces the literal 0*.25 of :193.
 * ---------------------------------------------------------------------------------------- */
/* euler_fluxes.jl:51-89; UL/UR = (rho,u,v,w,beta) */
static void euler_fluxes_3d_r(const real* UL, const real* UR, const real* logL, const real* logR, real* Fx,
                             real* Fy, real* Fz) {
  real rhoL = UL[0], uL = UL[1], vL = UL[2], wL = UL[3], betaL = UL[4];
  real rhoR = UR[0], uR = UR[1], vR = UR[2], wR = UR[3], betaR = UR[4];
  real rholog = logmean_r(rhoL, rhoR, logL[0], logR[0]);
  real betalog = logmean_r(betaL, betaR, logL[1], logR[1]);
  real rhoavg = .5 * (rhoL + rhoR);
  real uavg = .5 * (uL + uR);
  real vavg = .5 * (vL + vR);
  real wavg = .5 * (wL + wR);
  real unorm = uL * uR + vL * vR + wL * wR;
  real pa = rhoavg / (betaL + betaR);
  real E_plus_p = rholog / (2 * (GAMMA - 1) * betalog) + pa + .5 * rholog * unorm;
  Fx[0] = rholog * uavg;
  Fx[1] = Fx[0] * uavg + pa;
  Fx[2] = Fx[0] * vavg;
  Fx[3] = Fx[0] * wavg;
  Fx[4] = E_plus_p * uavg;
  Fy[0] = rholog * vavg;
  Fy[1] = Fx[2];
  Fy[2] = Fy[0] * vavg + pa;
  Fy[3] = Fy[0] * wavg;
  Fy[4] = E_plus_p * vavg;
  Fz[0] = rholog * wavg;
  Fz[1] = Fx[3];
  Fz[2] = Fy[3];
  Fz[3] = Fz[0] * wavg + pa;
  Fz[4] = E_plus_p * wavg;
}
void oracle_euler_fluxes_3d(const double* UL, const double* UR, const double* logL, const double* logR, double* Fx,
                            double* Fy, double* Fz) {
  real a[5], b[5], la[2], lb[2], fx[5], fy[5], fz[5];
  for (int i = 0; i < 5; ++i) { a[i] = UL[i]; b[i] = UR[i]; }
  for (int i = 0; i < 2; ++i) { la[i] = logL[i]; lb[i] = logR[i]; }
  euler_fluxes_3d_r(a, b, la, lb, fx, fy, fz);
  to_double(fx, 5, Fx);
  to_double(fy, 5, Fy);
  to_double(fz, 5, Fz);
}

/* euler_variables.jl:79-92, 5 fields */
static void v_ufun_3d_r(const real* U, real* V) {
  real rho = U[0], E = U[4];
  real rhoe = E - .5 * (U[1] * U[1] + U[2] * U[2] + U[3] * U[3]) / rho;
  real sU = R_(log)((GAMMA - 1) * rhoe / R_(pow)(rho, GAMMA));
  V[0] = (-E + rhoe * (GAMMA + 1 - sU)) / rhoe;
  V[1] = U[1] / rhoe;
  V[2] = U[2] / rhoe;
  V[3] = U[3] / rhoe;
  V[4] = (-rho) / rhoe;
}
void oracle_v_ufun_3d(const double* U, double* V) {
  real u[5], v[5];
  for (int i = 0; i < 5; ++i) u[i] = U[i];
  v_ufun_3d_r(u, v);
  to_double(v, 5, V);
}

/* euler_variables.jl:95-120, 5 fields */
static void u_vfun_3d_r(const real* V, real* U) {
  real v5 = V[4];
  real vUnorm = V[1] * V[1] + V[2] * V[2] + V[3] * V[3];
  real s = GAMMA - V[0] + vUnorm / (2 * v5);
  real rhoeV = R_(pow)((GAMMA - 1) / R_(pow)(-v5, GAMMA), 1 / (GAMMA - 1)) * R_(exp)(-s / (GAMMA - 1));
  U[0] = rhoeV * (-v5);
  U[1] = rhoeV * V[1];
  U[2] = rhoeV * V[2];
  U[3] = rhoeV * V[3];
  U[4] = rhoeV * (1 - vUnorm / (2 * v5));
}
void oracle_u_vfun_3d(const double* V, double* U) {
  real v[5], u[5];
  for (int i = 0; i < 5; ++i) v[i] = V[i];
  u_vfun_3d_r(v, u);
  to_double(u, 5, U);
}

/* euler_variables.jl:30-48, 5 fields */
static real betafun_3d_r(const real* U) {
  real rhounorm = (U[1] * U[1] + U[2] * U[2] + U[3] * U[3]) / U[0];
  real p = (GAMMA - 1) * (U[4] - .5 * rhounorm);
  return U[0] / (2 * p);
}
double oracle_betafun_3d(const double* U) {
  real u[5];
  for (int i = 0; i < 5; ++i) u[i] = U[i];
  return (double)betafun_3d_r(u);
}

typedef struct {
  int K, Nq, Nfq;
  const double *Ef /*Nfq x Nq*/, *Qr, *Qs, *Qt /*Nh x Nh droptol'd*/, *Ph /*Nq x Nh, includes the 2*/, *Lf /*Nq x Nfq*/;
  const int *rowptr, *colidx; /* Qnzids, 0-based */
  const double* vgeo[9];      /* rxJ,sxJ,txJ,ryJ,syJ,tyJ,rzJ,szJ,tzJ: [K][Nh] */
  const double *J, *wJq;      /* [K][Nq] */
  const double *nxJ, *nyJ, *nzJ, *sJ; /* [K][Nfq] */
  const int64_t* mapP;        /* [K][Nfq] 1-based */
  double lf_scale;
} oracle_hex_t;

/* sparse_hadamard_sum, dg3D_euler_hex.jl:122-164 (metric of a pair = average of the two nodes, :145-146) */
static void sparse_hadamard_sum_hex(const oracle_hex_t* c, const real* Qhe /*[5][Nh]*/, const real* ge /*[9][Nh]*/,
                                    real* out /*[5][Nh] + 2 Nh of scratch*/) {
  const int Nh = c->Nq + c->Nfq;
  real *lrho = out + 5 * Nh, *lbeta = out + 6 * Nh;   /* scratch behind the result: the caller's buffer is [5 + 2][Nh] (any degree) */
  for (int i = 0; i < Nh; ++i) {
    lrho[i] = R_(log)(Qhe[i]);
    lbeta[i] = R_(log)(Qhe[4 * Nh + i]);
  }
  for (int i = 0; i < Nh; ++i) {
    real Qi[5], li[2] = {lrho[i], lbeta[i]}, rhsi[5] = {0, 0, 0, 0, 0};
    for (int f = 0; f < 5; ++f) Qi[f] = Qhe[f * Nh + i];
    for (int t = c->rowptr[i]; t < c->rowptr[i + 1]; ++t) {
      const int j = c->colidx[t];
      real Qj[5], lj[2] = {lrho[j], lbeta[j]}, g[9], Fx[5], Fy[5], Fz[5];
      for (int f = 0; f < 5; ++f) Qj[f] = Qhe[f * Nh + j];
      for (int m = 0; m < 9; ++m) g[m] = .5 * (ge[m * Nh + i] + ge[m * Nh + j]);
      euler_fluxes_3d_r(Qi, Qj, li, lj, Fx, Fy, Fz);
      for (int f = 0; f < 5; ++f) {
        real Fr = g[0] * Fx[f] + g[3] * Fy[f] + g[6] * Fz[f];
        real Fs = g[1] * Fx[f] + g[4] * Fy[f] + g[7] * Fz[f];
        real Ft = g[2] * Fx[f] + g[5] * Fy[f] + g[8] * Fz[f];
        rhsi[f] += c->Qr[i * Nh + j] * Fr + c->Qs[i * Nh + j] * Fs + c->Qt[i * Nh + j] * Ft;
      }
    }
    for (int f = 0; f < 5; ++f) out[f * Nh + i] = rhsi[f];
  }
}

/* rhs, dg3D_euler_hex.jl:167-222.  Q, rhs: [5][K][Nq].  Returns rhstest (0 unless compute_rhstest). */
static real hex_rhs_r(const oracle_hex_t* c, const real* Q, int compute_rhstest, real* rhs) {
  const int K = c->K, Nq = c->Nq, Nfq = c->Nfq, Nh = Nq + Nfq;
  const size_t KNq = (size_t)K * Nq, KNf = (size_t)K * Nfq, KNh = (size_t)K * Nh;
  real* VU = (real*)malloc(5 * KNq * sizeof(real));
  real* VUf = (real*)malloc(5 * KNf * sizeof(real));
  real* Uf = (real*)malloc(5 * KNf * sizeof(real));
  real* Qh = (real*)malloc(5 * KNh * sizeof(real));
  real* lam = (real*)malloc(KNf * sizeof(real));
  real* flux = (real*)malloc(5 * KNf * sizeof(real));
  /* :174-176 */
#pragma omp parallel for num_threads(g_threads) schedule(static)
  for (size_t n = 0; n < KNq; ++n) {
    real U[5], V[5];
    for (int f = 0; f < 5; ++f) U[f] = Q[f * KNq + n];
    v_ufun_3d_r(U, V);
    for (int f = 0; f < 5; ++f) VU[f * KNq + n] = V[f];
  }
  for (int f = 0; f < 5; ++f) matmul_elems(c->Ef, Nfq, Nq, VU + f * KNq, VUf + f * KNf, K);
#pragma omp parallel for num_threads(g_threads) schedule(static)
  for (size_t n = 0; n < KNf; ++n) {
    real V[5], U[5];
    for (int f = 0; f < 5; ++f) V[f] = VUf[f * KNf + n];
    u_vfun_3d_r(V, U);
    for (int f = 0; f < 5; ++f) Uf[f * KNf + n] = U[f];
  }
  /* :177-182 Uh = vcat(Q,Uf); beta; Qh = (rho,u,v,w,beta) */
#pragma omp parallel for num_threads(g_threads) schedule(static)
  for (int e = 0; e < K; ++e)
    for (int i = 0; i < Nh; ++i) {
      real U[5];
      for (int f = 0; f < 5; ++f)
        U[f] = i < Nq ? Q[f * KNq + (size_t)e * Nq + i] : Uf[f * KNf + (size_t)e * Nfq + (i - Nq)];
      size_t o = (size_t)e * Nh + i;
      Qh[o] = U[0];
      Qh[KNh + o] = U[1] / U[0];
      Qh[2 * KNh + o] = U[2] / U[0];
      Qh[3 * KNh + o] = U[3] / U[0];
      Qh[4 * KNh + o] = betafun_3d_r(U);
    }
  /* :189-192 lam */
#pragma omp parallel for num_threads(g_threads) schedule(static)
  for (size_t n = 0; n < KNf; ++n) {
    real rhoU_n = (Uf[KNf + n] * c->nxJ[n] + Uf[2 * KNf + n] * c->nyJ[n] + Uf[3 * KNf + n] * c->nzJ[n]) / c->sJ[n];
    lam[n] = R_(fabs)(wavespeed_r(Uf[n], rhoU_n, Uf[4 * KNf + n]));
  }
  /* :185-198 QM/QP, LFc, surface flux */
#pragma omp parallel for num_threads(g_threads) schedule(static)
  for (int e = 0; e < K; ++e)
    for (int i = 0; i < Nfq; ++i) {
      size_t n = (size_t)e * Nfq + i;
      size_t p = (size_t)(c->mapP[n] - 1);
      size_t ep = p / Nfq, ip = p % Nfq;
      size_t om = (size_t)e * Nh + Nq + i, op = ep * Nh + Nq + ip;
      real QM[5], QP[5], Fx[5], Fy[5], Fz[5];
      for (int f = 0; f < 5; ++f) {
        QM[f] = Qh[f * KNh + om];
        QP[f] = Qh[f * KNh + op];
      }
      real lM[2] = {R_(log)(QM[0]), R_(log)(QM[4])}, lP[2] = {R_(log)(QP[0]), R_(log)(QP[4])};
      euler_fluxes_3d_r(QM, QP, lM, lP, Fx, Fy, Fz);
      real LFc = c->lf_scale * R_(fmax)(lam[n], lam[p]) * c->sJ[n];
      for (int f = 0; f < 5; ++f)
        flux[f * KNf + n] = Fx[f] * c->nxJ[n] + Fy[f] * c->nyJ[n] + Fz[f] * c->nzJ[n] - LFc * (Uf[f * KNf + p] - Uf[f * KNf + n]);
    }
  for (int f = 0; f < 5; ++f) matmul_elems(c->Lf, Nq, Nfq, flux + f * KNf, rhs + f * KNq, K);
  /* :200-210 volume loop */
#pragma omp parallel for num_threads(g_threads) schedule(static)
  for (int e = 0; e < K; ++e) {
    real* Qhe = (real*)malloc((size_t)(5 + 9 + 5 + 2) * Nh * sizeof(real));   /* + 2 Nh: logs of the element's hybrid nodes */
    real* ge = Qhe + 5 * Nh;
    real* QFe = ge + 9 * Nh;
    for (int f = 0; f < 5; ++f)
      for (int i = 0; i < Nh; ++i) Qhe[f * Nh + i] = Qh[f * KNh + (size_t)e * Nh + i];
    for (int m = 0; m < 9; ++m)
      for (int i = 0; i < Nh; ++i) ge[m * Nh + i] = c->vgeo[m][(size_t)e * Nh + i];
    sparse_hadamard_sum_hex(c, Qhe, ge, QFe);
    for (int f = 0; f < 5; ++f)
      for (int i = 0; i < Nq; ++i) {
        real s = 0.0;
        for (int j = 0; j < Nh; ++j) s += c->Ph[i * Nh + j] * QFe[f * Nh + j];
        rhs[f * KNq + (size_t)e * Nq + i] += s;
      }
    free(Qhe);
  }
  /* :212 */
  for (int f = 0; f < 5; ++f)
    for (size_t n = 0; n < KNq; ++n) rhs[f * KNq + n] = -rhs[f * KNq + n] / c->J[n];
  real rhstest = 0.0;
  if (compute_rhstest)
    for (int f = 0; f < 5; ++f)
      for (size_t n = 0; n < KNq; ++n) rhstest += c->wJq[n] * VU[f * KNq + n] * rhs[f * KNq + n];
  free(VU); free(VUf); free(Uf); free(Qh); free(lam); free(flux);
  return rhstest;
}
double oracle_hex_rhs(const oracle_hex_t* c, const double* Q, int compute_rhstest, double* rhs) {
  const size_t n = 5 * (size_t)c->K * c->Nq;
  real* Qr_ = to_real(Q, n);
  real* out = (real*)malloc(n * sizeof(real));
  real rt = hex_rhs_r(c, Qr_, compute_rhstest, out);
  to_double(out, n, rhs);
  free(Qr_); free(out);
  return (double)rt;
}
