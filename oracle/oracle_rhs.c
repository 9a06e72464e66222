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
double oracle_logmean(double aL, double aR, double logL, double logR) {
  double da = aR - aL;
  double aavg = .5 * (aR + aL);
  double f = da / aavg;
  double v = f * f;
  if (fabs(f) < 1e-4)
    return aavg * (1 + v * (-.2 - v * (.0512 - v * 0.026038857142857)));
  return -da / (logL - logR);
}

/* euler_fluxes.jl:23-48; UL/UR = (rho,u,v,beta), logs = (log rho, log beta) */
void oracle_euler_fluxes_2d(const double* UL, const double* UR, const double* logL, const double* logR,
                            double* Fx, double* Fy) {
  double rhoL = UL[0], uL = UL[1], vL = UL[2], betaL = UL[3];
  double rhoR = UR[0], uR = UR[1], vR = UR[2], betaR = UR[3];
  double rholog = oracle_logmean(rhoL, rhoR, logL[0], logR[0]);
  double betalog = oracle_logmean(betaL, betaR, logL[1], logR[1]);
  double rhoavg = .5 * (rhoL + rhoR);
  double uavg = .5 * (uL + uR);
  double vavg = .5 * (vL + vR);
  double unorm = uL * uR + vL * vR;
  double pa = rhoavg / (betaL + betaR);
  double f4aux = rholog / (2 * (GAMMA - 1) * betalog) + pa + .5 * rholog * unorm;
  Fx[0] = rholog * uavg;
  Fx[1] = Fx[0] * uavg + pa;
  Fx[2] = Fx[0] * vavg;
  Fx[3] = f4aux * uavg;
  Fy[0] = rholog * vavg;
  Fy[1] = Fx[2];
  Fy[2] = Fy[0] * vavg + pa;
  Fy[3] = f4aux * vavg;
}

/* euler_variables.jl:79-92 (v_ufun via rhoefun :59-62, sfun :65-68) */
void oracle_v_ufun(const double* U, double* V) {
  double rho = U[0], rhou = U[1], rhov = U[2], E = U[3];
  double rhoe = E - .5 * (rhou * rhou + rhov * rhov) / rho;
  double sU = log((GAMMA - 1) * rhoe / pow(rho, GAMMA));
  V[0] = (-E + rhoe * (GAMMA + 1 - sU)) / rhoe;
  V[1] = rhou / rhoe;
  V[2] = rhov / rhoe;
  V[3] = (-rho) / rhoe;
}

/* euler_variables.jl:95-120 (u_vfun via s_vfun, rhoe_vfun) */
void oracle_u_vfun(const double* V, double* U) {
  double v1 = V[0], v2 = V[1], v3 = V[2], v4 = V[3];
  double vUnorm = v2 * v2 + v3 * v3;
  double s = GAMMA - v1 + vUnorm / (2 * v4);
  double rhoeV = pow((GAMMA - 1) / pow(-v4, GAMMA), 1 / (GAMMA - 1)) * exp(-s / (GAMMA - 1));
  U[0] = rhoeV * (-v4);
  U[1] = rhoeV * v2;
  U[2] = rhoeV * v3;
  U[3] = rhoeV * (1 - vUnorm / (2 * v4));
}

/* euler_variables.jl:30-48 */
double oracle_betafun(const double* U) {
  double rhounorm = (U[1] * U[1] + U[2] * U[2]) / U[0];
  double p = (GAMMA - 1) * (U[3] - .5 * rhounorm);
  return U[0] / (2 * p);
}

/* euler_variables.jl:7-10 -- note sqrt(abs(u_n)) (quirk Q1) */
double oracle_wavespeed(double rho, double rhou, double E) {
  double p = (GAMMA - 1) * (E - .5 * (rhou * rhou) / rho);
  double cvel = sqrt(GAMMA * p / rho);
  return sqrt(fabs(rhou / rho)) + cvel;
}

/* ------------------------------------------------------------------------------------------
 * small dense helpers:  Y(n x K) = A(n x m) * X(m x K), per element
 * ---------------------------------------------------------------------------------------- */
static void matmul_elems(const double* A, int n, int m, const double* X, double* Y, int K) {
#pragma omp parallel for num_threads(g_threads) schedule(static)
  for (int e = 0; e < K; ++e)
    for (int i = 0; i < n; ++i) {
      double s = 0.0;
      for (int j = 0; j < m; ++j) s += A[i * m + j] * X[(size_t)e * m + j];
      Y[(size_t)e * n + i] = s;
    }
}

/* ------------------------------------------------------------------------------------------
 * Euler, collocated quad:  examples/dg2D_euler_quad.jl:102-194
 * ---------------------------------------------------------------------------------------- */
/* sparse_hadamard_sum, dg2D_euler_quad.jl:102-138: row-wise, nonzero column ids per row */
static void sparse_hadamard_sum(const double* Qhe /*[4][Nh]*/, int Nh, const double* Qr, const double* Qs,
                                const int* rowptr, const int* colidx, double rxJ, double sxJ, double ryJ,
                                double syJ, double* out /*[4][Nh]*/) {
  double lg[2 * 64 * 4];
  double* lrho = lg;
  double* lbeta = lg + Nh;
  for (int i = 0; i < Nh; ++i) {
    lrho[i] = log(Qhe[0 * Nh + i]);
    lbeta[i] = log(Qhe[3 * Nh + i]);
  }
  for (int i = 0; i < Nh; ++i) {
    double Qi[4] = {Qhe[i], Qhe[Nh + i], Qhe[2 * Nh + i], Qhe[3 * Nh + i]};
    double li[2] = {lrho[i], lbeta[i]};
    double rhsi[4] = {0, 0, 0, 0};
    for (int t = rowptr[i]; t < rowptr[i + 1]; ++t) {
      int j = colidx[t];
      double Qj[4] = {Qhe[j], Qhe[Nh + j], Qhe[2 * Nh + j], Qhe[3 * Nh + j]};
      double lj[2] = {lrho[j], lbeta[j]};
      double Fx[4], Fy[4];
      oracle_euler_fluxes_2d(Qi, Qj, li, lj, Fx, Fy);
      for (int f = 0; f < 4; ++f) {
        double Fr = rxJ * Fx[f] + ryJ * Fy[f];
        double Fs = sxJ * Fx[f] + syJ * Fy[f];
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
double oracle_euler_rhs(int K, int Nq, int Nfq, const double* Q, const double* Ef, const double* Qr,
                        const double* Qs, const int* rowptr, const int* colidx, const double* Ph,
                        const double* Lf, const double* rxJ, const double* sxJ, const double* ryJ,
                        const double* syJ, const double* J, const double* wJq, const double* nxJ,
                        const double* nyJ, const double* sJ, const int64_t* mapP, double lf_scale,
                        int compute_rhstest, double* rhs) {
  const int Nh = Nq + Nfq;
  const size_t KNq = (size_t)K * Nq, KNf = (size_t)K * Nfq, KNh = (size_t)K * Nh;
  double* VU = (double*)malloc(4 * KNq * sizeof(double));
  double* VUf = (double*)malloc(4 * KNf * sizeof(double));
  double* Uf = (double*)malloc(4 * KNf * sizeof(double));
  double* Qh = (double*)malloc(4 * KNh * sizeof(double));
  double* lam = (double*)malloc(KNf * sizeof(double));
  double* flux = (double*)malloc(4 * KNf * sizeof(double));

  /* :149 VU = v_ufun(Q...) */
#pragma omp parallel for num_threads(g_threads) schedule(static)
  for (size_t n = 0; n < KNq; ++n) {
    double U[4] = {Q[n], Q[KNq + n], Q[2 * KNq + n], Q[3 * KNq + n]}, V[4];
    oracle_v_ufun(U, V);
    for (int f = 0; f < 4; ++f) VU[f * KNq + n] = V[f];
  }
  /* :150 Uf = u_vfun(Ef*VU) */
  for (int f = 0; f < 4; ++f) matmul_elems(Ef, Nfq, Nq, VU + f * KNq, VUf + f * KNf, K);
#pragma omp parallel for num_threads(g_threads) schedule(static)
  for (size_t n = 0; n < KNf; ++n) {
    double V[4] = {VUf[n], VUf[KNf + n], VUf[2 * KNf + n], VUf[3 * KNf + n]}, U[4];
    oracle_u_vfun(V, U);
    for (int f = 0; f < 4; ++f) Uf[f * KNf + n] = U[f];
  }
  /* :151-155 (rho,rhou,rhov,E) = vcat(Q,Uf); beta; Qh = (rho,u,v,beta) */
#pragma omp parallel for num_threads(g_threads) schedule(static)
  for (int e = 0; e < K; ++e)
    for (int i = 0; i < Nh; ++i) {
      double U[4];
      for (int f = 0; f < 4; ++f)
        U[f] = i < Nq ? Q[f * KNq + (size_t)e * Nq + i] : Uf[f * KNf + (size_t)e * Nfq + (i - Nq)];
      double beta = oracle_betafun(U);
      size_t o = (size_t)e * Nh + i;
      Qh[o] = U[0];
      Qh[KNh + o] = U[1] / U[0];
      Qh[2 * KNh + o] = U[2] / U[0];
      Qh[3 * KNh + o] = beta;
    }
  /* :162-164 lam */
#pragma omp parallel for num_threads(g_threads) schedule(static)
  for (size_t n = 0; n < KNf; ++n) {
    double rhoUM_n = (Uf[KNf + n] * nxJ[n] + Uf[2 * KNf + n] * nyJ[n]) / sJ[n];
    lam[n] = fabs(oracle_wavespeed(Uf[n], rhoUM_n, Uf[3 * KNf + n]));
  }
  /* :158-170 QM/QP, LFc, surface flux */
#pragma omp parallel for num_threads(g_threads) schedule(static)
  for (int e = 0; e < K; ++e)
    for (int i = 0; i < Nfq; ++i) {
      size_t n = (size_t)e * Nfq + i;
      size_t p = (size_t)(mapP[n] - 1);
      size_t ep = p / Nfq, ip = p % Nfq;
      size_t om = (size_t)e * Nh + Nq + i, op = ep * Nh + Nq + ip;
      double QM[4] = {Qh[om], Qh[KNh + om], Qh[2 * KNh + om], Qh[3 * KNh + om]};
      double QP[4] = {Qh[op], Qh[KNh + op], Qh[2 * KNh + op], Qh[3 * KNh + op]};
      double lM[2] = {log(QM[0]), log(QM[3])}, lP[2] = {log(QP[0]), log(QP[3])};
      double Fx[4], Fy[4];
      oracle_euler_fluxes_2d(QM, QP, lM, lP, Fx, Fy);
      double LFc = lf_scale * fmax(lam[n], lam[p]) * sJ[n];
      for (int f = 0; f < 4; ++f)
        flux[f * KNf + n] = Fx[f] * nxJ[n] + Fy[f] * nyJ[n] - LFc * (Uf[f * KNf + p] - Uf[f * KNf + n]);
    }
  /* :170 rhsQ = Lf*flux */
  for (int f = 0; f < 4; ++f) matmul_elems(Lf, Nq, Nfq, flux + f * KNf, rhs + f * KNq, K);
  /* :173-182 volume loop */
#pragma omp parallel for num_threads(g_threads) schedule(static)
  for (int e = 0; e < K; ++e) {
    double Qhe[4 * 64 * 4], QFe[4 * 64 * 4];
    for (int f = 0; f < 4; ++f)
      for (int i = 0; i < Nh; ++i) Qhe[f * Nh + i] = Qh[f * KNh + (size_t)e * Nh + i];
    size_t g = (size_t)e * Nh; /* vgeo_local = (rxJ,sxJ,ryJ,syJ)[1,e] */
    sparse_hadamard_sum(Qhe, Nh, Qr, Qs, rowptr, colidx, rxJ[g], sxJ[g], ryJ[g], syJ[g], QFe);
    for (int f = 0; f < 4; ++f)
      for (int i = 0; i < Nq; ++i) {
        double s = 0.0;
        for (int j = 0; j < Nh; ++j) s += Ph[i * Nh + j] * QFe[f * Nh + j];
        rhs[f * KNq + (size_t)e * Nq + i] += 2 * s;
      }
  }
  /* :184 */
  for (int f = 0; f < 4; ++f)
    for (size_t n = 0; n < KNq; ++n) rhs[f * KNq + n] = -rhs[f * KNq + n] / J[n];
  /* :186-191 */
  double rhstest = 0.0;
  if (compute_rhstest)
    for (int f = 0; f < 4; ++f)
      for (size_t n = 0; n < KNq; ++n) rhstest += wJq[n] * VU[f * KNq + n] * rhs[f * KNq + n];
  free(VU); free(VUf); free(Uf); free(Qh); free(lam); free(flux);
  return rhstest;
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
static void v_hardcoded(const double* U, double* V) {
  double n = U[1] * U[1] + U[2] * U[2];
  double rhoe = U[3] - .5 * n / U[0];
  double sU = log(0.4 * rhoe / pow(U[0], 1.4));
  V[0] = (-U[3] + rhoe * (2.4 - sU)) / rhoe;
  V[1] = U[1] / rhoe;
  V[2] = U[2] / rhoe;
  V[3] = -U[0] / rhoe;
}

/* rhs_inviscid!, :447-528 with update_flux! :308-324 and flux_differencing! :326-348.
 * Q, rhs: [4][K][Np]. */
void oracle_cns_rhs_inviscid(const oracle_cns_t* c, const double* Q, double* rhs) {
  const int K = c->K, Np = c->Np, Nq = c->Nq, Nfq = c->Nfq, Nh = Nq + Nfq;
  const size_t KNp = (size_t)K * Np, KNq = (size_t)K * Nq, KNf = (size_t)K * Nfq, KNh = (size_t)K * Nh;
  double* Qq = (double*)malloc(4 * KNq * sizeof(double));
  double* VU = (double*)malloc(4 * KNq * sizeof(double));
  double* Uh = (double*)malloc(4 * KNh * sizeof(double));
  double* Qh = (double*)malloc(4 * KNh * sizeof(double));
  double* QP = (double*)malloc(4 * KNf * sizeof(double));
  double* lam = (double*)malloc(KNf * sizeof(double));
  double* flux = (double*)malloc(4 * KNf * sizeof(double));
  double* QF = (double*)calloc(4 * KNh, sizeof(double));
  double* tmpN = (double*)malloc(4 * KNp * sizeof(double));

  for (int f = 0; f < 4; ++f) matmul_elems(c->Vq, Nq, Np, Q + f * KNp, Qq + f * KNq, K); /* :459 */
#pragma omp parallel for num_threads(g_threads) schedule(static)
  for (size_t n = 0; n < KNq; ++n) { /* :461-467 */
    double U[4] = {Qq[n], Qq[KNq + n], Qq[2 * KNq + n], Qq[3 * KNq + n]}, V[4];
    v_hardcoded(U, V);
    for (int f = 0; f < 4; ++f) VU[f * KNq + n] = V[f];
  }
  for (int f = 0; f < 4; ++f) matmul_elems(c->VhP, Nh, Nq, VU + f * KNq, Uh + f * KNh, K); /* :470 */
#pragma omp parallel for num_threads(g_threads) schedule(static)
  for (size_t n = 0; n < KNh; ++n) { /* :473-488 */
    double v1 = Uh[n], v2 = Uh[KNh + n], v3 = Uh[2 * KNh + n], v4 = Uh[3 * KNh + n];
    double tmp = v2 * v2 + v3 * v3;
    double tmp2 = pow(0.4 / pow(-v4, 1.4), 1 / 0.4) * exp(-(1.4 - v1 + tmp / (2 * v4)) / 0.4);
    double u1 = tmp2 * (-v4), u2 = tmp2 * v2, u3 = tmp2 * v3, u4 = tmp2 * (1 - tmp / (2 * v4));
    Uh[n] = u1; Uh[KNh + n] = u2; Uh[2 * KNh + n] = u3; Uh[3 * KNh + n] = u4;
    double beta = u1 / (2 * 0.4 * (u4 - .5 * (u2 * u2 + u3 * u3) / u1));
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
    double nx = c->nxJ[n] / c->sJ[n], ny = c->nyJ[n] / c->sJ[n];
    double u1 = Qh[KNh + om], u2 = Qh[2 * KNh + om];
    double Un = u1 * nx + u2 * ny;
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
      double rhoM = Uh[om], rhouM = Uh[KNh + om], rhovM = Uh[2 * KNh + om], EM = Uh[3 * KNh + om];
      double rhoUM_n = (rhouM * c->nxJ[n] + rhovM * c->nyJ[n]) / c->sJ[n];
      lam[n] = fabs(sqrt(fabs(rhoUM_n / rhoM)) + sqrt(1.4 * 0.4 * (EM - .5 * rhoUM_n * rhoUM_n / rhoM) / rhoM));
    }
  /* :510-514 update_flux! then LIFT */
#pragma omp parallel for num_threads(g_threads) schedule(static)
  for (int e = 0; e < K; ++e)
    for (int i = 0; i < Nfq; ++i) {
      size_t n = (size_t)e * Nfq + i, p = (size_t)(c->mapP[n] - 1);
      size_t om = (size_t)e * Nh + Nq + i, op = (p / Nfq) * Nh + Nq + (p % Nfq);
      double QMl[4] = {Qh[om], Qh[KNh + om], Qh[2 * KNh + om], Qh[3 * KNh + om]};
      double QPl[4] = {QP[n], QP[KNf + n], QP[2 * KNf + n], QP[3 * KNf + n]};
      double lM[2] = {log(QMl[0]), log(QMl[3])}, lP[2] = {log(QPl[0]), log(QPl[3])};
      double Fx[4], Fy[4];
      oracle_euler_fluxes_2d(QPl, QMl, lP, lM, Fx, Fy); /* (QP,QM) order, quirk Q8 */
      double LFc = nolf[n] ? 0.0 : .25 * fmax(lam[n], lam[p]) * c->sJ[n];
      for (int f = 0; f < 4; ++f) {
        double v = Fx[f] * c->nxJ[n] + Fy[f] * c->nyJ[n];
        if (c->inviscid_dissp) v -= LFc * (Uh[f * KNh + op] - Uh[f * KNh + om]);
        flux[f * KNf + n] = v;
      }
    }
  for (int f = 0; f < 4; ++f) matmul_elems(c->LIFT, Np, Nfq, flux + f * KNf, rhs + f * KNp, K);
  /* :516 flux_differencing! (symmetric; dense Qrh/Qsh; skip face x face; diagonal pairs evaluated) */
#pragma omp parallel for num_threads(g_threads) schedule(static)
  for (int k = 0; k < K; ++k) {
    size_t g = (size_t)k * Nh;
    double rx = c->rxJ[g], ry = c->ryJ[g], sx = c->sxJ[g], sy = c->syJ[g];
    for (int j = 0; j < Nh; ++j)
      for (int i = j; i < Nh; ++i)
        if (i < Nq || j < Nq) {
          double Qi[4] = {Qh[g + i], Qh[KNh + g + i], Qh[2 * KNh + g + i], Qh[3 * KNh + g + i]};
          double Qj[4] = {Qh[g + j], Qh[KNh + g + j], Qh[2 * KNh + g + j], Qh[3 * KNh + g + j]};
          double li[2] = {log(Qi[0]), log(Qi[3])}, lj[2] = {log(Qj[0]), log(Qj[3])};
          double Fx[4], Fy[4];
          oracle_euler_fluxes_2d(Qi, Qj, li, lj, Fx, Fy);
          double Qr = c->Qrh[i * Nh + j], Qs = c->Qsh[i * Nh + j];
          for (int d = 0; d < 4; ++d) {
            double val = 2 * ((rx * Qr + sx * Qs) * Fx[d] + (ry * Qr + sy * Qs) * Fy[d]);
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

/* viscous_matrices!, :613-645 (let lambda = -lambda, quirk Q4); entries not listed stay 0 */
static void viscous_matrices(double Kxx[4][4], double Kxy[4][4], double Kyy[4][4], const double* v,
                             double lambda_in, double mu, double Pr) {
  double lambda = -lambda_in;
  double v2 = v[1], v3 = v[2], v4 = v[3];
  double inv_v4_cubed = 1 / (v4 * v4 * v4);
  double l2m = lambda + 2.0 * mu;
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
double oracle_cns_rhs_viscous(const oracle_cns_t* c, const double* Q, double* rhs) {
  const int K = c->K, Np = c->Np, Nq = c->Nq, Nfq = c->Nfq, Nh = Nq + Nfq;
  const size_t KNp = (size_t)K * Np, KNq = (size_t)K * Nq, KNf = (size_t)K * Nfq;
#define ALLOC(n) (double*)calloc((n), sizeof(double))
  double *Qq = ALLOC(4 * KNq), *VUq0 = ALLOC(4 * KNq), *VU = ALLOC(4 * KNp), *VUf = ALLOC(4 * KNf),
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
    double U[4] = {Qq[n], Qq[KNq + n], Qq[2 * KNq + n], Qq[3 * KNq + n]}, V[4];
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
    double vf2 = VUf[KNf + n], vf3 = VUf[2 * KNf + n], vf4 = VUf[3 * KNf + n];
    if (c->BCTYPE == 1) {
      VUP[KNf + n] = lid ? -vf2 - 2 * (c->vlid ? c->vlid[b] : 1.0) * vf4 : -vf2;
      VUP[2 * KNf + n] = -vf3;
      VUP[3 * KNf + n] = vf4;
    } else if (c->BCTYPE == 2) {
      double theta = 1.0 / (0.3 * 0.3) / 1.4 / 0.4;
      VUP[KNf + n] = lid ? 2.0 / theta - vf2 : -vf2;
      VUP[2 * KNf + n] = -vf3;
      VUP[3 * KNf + n] = -2.0 / theta - vf4;
    } else if (c->BCTYPE == 3) {
      double nx = c->nxJ[n] / c->sJ[n], ny = c->nyJ[n] / c->sJ[n];
      double VUn = vf2 * nx + vf3 * ny;
      VUP[3 * KNf + n] = vf4;
      VUP[KNf + n] = vf2 - 2 * VUn * nx;
      VUP[2 * KNf + n] = vf3 - 2 * VUn * ny;
    } else if (c->BCTYPE == 4) { /* dg2D_CNS_modalESDG.jl:187-203: VL = v_ufun(rhoL, rhoL*uL, rhoL*vL, EL) / VUf */
      if (lid) {
        double rho = c->inflow[0], u = c->inflow[1], v = c->inflow[2], p = c->inflow[3];
        double U[4] = {rho, rho * u, rho * v, p / (GAMMA - 1) + .5 * rho * (u * u + v * v)}, VL[4];
        oracle_v_ufun(U, VL);
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
    double Kxx[4][4] = {{0}}, Kxy[4][4] = {{0}}, Kyy[4][4] = {{0}};
    for (int i = 0; i < Nq; ++i) {
      size_t n = (size_t)e * Nq + i;
      double vqi[4] = {VUq[n], VUq[KNq + n], VUq[2 * KNq + n], VUq[3 * KNq + n]};
      viscous_matrices(Kxx, Kxy, Kyy, vqi, c->lambda, c->mu, c->Pr);
      for (int col = 1; col < 4; ++col) {
        double vxi = VUxq[col * KNq + n], vyi = VUyq[col * KNq + n];
        for (int row = 1; row < 4; ++row) {
          sxq[row * KNq + n] += Kxx[row][col] * vxi + Kxy[row][col] * vyi;
          syq[row * KNq + n] += Kxy[col][row] * vxi + Kyy[row][col] * vyi;
        }
      }
    }
  }
  /* :803-807 */
  double rhstest = 0.0;
  for (int f = 0; f < 4; ++f) {
    double a = 0.0, b = 0.0;
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
        const double vl = c->vlid ? c->vlid[b] : 1.0;
        sxP[3 * KNf + n] = -sxf[3 * KNf + n] + 2 * vl * sxf[KNf + n];
        syP[3 * KNf + n] = -syf[3 * KNf + n] + 2 * vl * syf[KNf + n];
      } else {
        sxP[3 * KNf + n] = -sxf[3 * KNf + n];
        syP[3 * KNf + n] = -syf[3 * KNf + n];
      }
    } else if (c->BCTYPE == 2) {
      for (int f = 1; f <= 3; ++f) { sxP[f * KNf + n] = sxf[f * KNf + n]; syP[f * KNf + n] = syf[f * KNf + n]; }
    } else if (c->BCTYPE == 3) {
      double n1 = c->nxJ[n] / c->sJ[n], n2 = c->nyJ[n] / c->sJ[n];
      double sx1 = sxf[KNf + n], sx2 = sxf[2 * KNf + n], sy1 = syf[KNf + n], sy2 = syf[2 * KNf + n];
      double snx = sx1 * n1 + sx2 * n2, sny = sy1 * n1 + sy2 * n2;
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
      double tau = -1 / c->Re / VUf[3 * KNf + n];
      for (int f = 1; f < 4; ++f) pen[f * KNf + n] = tau * (VUP[f * KNf + n] - VUf[f * KNf + n]);
    }
    for (int b = 0; b < c->Nb; ++b) {
      size_t n = (size_t)(c->mapB[b] - 1);
      double tau = -1 / c->Re / VUf[3 * KNf + n];
      double dV2 = VUP[KNf + n] - VUf[KNf + n], dV3 = VUP[2 * KNf + n] - VUf[2 * KNf + n],
             dV4 = VUP[3 * KNf + n] - VUf[3 * KNf + n];
      double a2 = 1.0 / 2 * (VUP[KNf + n] + VUf[KNf + n]), a3 = 1.0 / 2 * (VUP[2 * KNf + n] + VUf[2 * KNf + n]);
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
        double vol = c->rxJ[g] * t1[n] + c->sxJ[g] * t2[n] + c->ryJ[g] * t3[n] + c->syJ[g] * t4[n];
        double r = (vol + tl[n]) / c->J[n];
        if (c->viscous_dissp && c->BCTYPE != 4) r = r + penL[f * KNp + n];
        rhs[f * KNp + n] = r;
      }
  }
  free(Qq); free(VUq0); free(VU); free(VUf); free(VUP); free(VUx); free(VUy); free(VUxq); free(VUyq);
  free(VUq); free(sxq); free(syq); free(sx); free(sy); free(sxf); free(syf); free(sxP); free(syP);
  free(pen); free(penL); free(t1); free(t2); free(t3); free(t4); free(tf); free(tl);
  return rhstest;
}

/* rhsRK!, :955-972.  diag[0] = rhstest, diag[1] = rhstest_visc (computed when compute_diag). */
void oracle_cns_rhsRK(const oracle_cns_t* c, const double* Q, double* rhs, int compute_diag, double* diag) {
  const int K = c->K, Np = c->Np, Nq = c->Nq;
  const size_t KNp = (size_t)K * Np, KNq = (size_t)K * Nq;
  double* visc = (double*)malloc(4 * KNp * sizeof(double));
  oracle_cns_rhs_inviscid(c, Q, rhs);
  double visc_test = oracle_cns_rhs_viscous(c, Q, visc);
  for (size_t n = 0; n < 4 * KNp; ++n) rhs[n] = rhs[n] + visc[n];
  if (compute_diag) {
    double *Qq = ALLOC(4 * KNq), *VU = ALLOC(4 * KNq), *VUn = ALLOC(4 * KNp), *VUq = ALLOC(4 * KNq),
           *rq = ALLOC(KNq), *vq = ALLOC(KNq);
    for (int f = 0; f < 4; ++f) matmul_elems(c->Vq, Nq, Np, Q + f * KNp, Qq + f * KNq, K);
    for (size_t n = 0; n < KNq; ++n) {
      double U[4] = {Qq[n], Qq[KNq + n], Qq[2 * KNq + n], Qq[3 * KNq + n]}, V[4];
      oracle_v_ufun(U, V);
      for (int f = 0; f < 4; ++f) VU[f * KNq + n] = V[f];
    }
    double rhstest = 0.0, rhstest_visc = 0.0;
    for (int f = 0; f < 4; ++f) {
      matmul_elems(c->Pq, Np, Nq, VU + f * KNq, VUn + f * KNp, K); /* VUq = Vq*Pq*VU */
      matmul_elems(c->Vq, Nq, Np, VUn + f * KNp, VUq + f * KNq, K);
      matmul_elems(c->Vq, Nq, Np, rhs + f * KNp, rq, K);
      matmul_elems(c->Vq, Nq, Np, visc + f * KNp, vq, K);
      double a = 0.0, b = 0.0;
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

/* ------------------------------------------------------------------------------------------
 * Euler, collocated hex:  examples/dg3D_euler_hex.jl:122-222 (3D physics: euler_fluxes.jl:51-89,
 * euler_variables.jl with 3-component momentum).  The reference flags the driver broken (:1): the
 * breakage is in hex_face_vertices (src/UniformHexMesh.jl:83-93), i.e. in the set-up, not in `rhs`;
 * see oracle/ref_setup.py.  lf_scale replaces the literal 0*.25 of :193.
 * ---------------------------------------------------------------------------------------- */
/* euler_fluxes.jl:51-89; UL/UR = (rho,u,v,w,beta) */
void oracle_euler_fluxes_3d(const double* UL, const double* UR, const double* logL, const double* logR,
                            double* Fx, double* Fy, double* Fz) {
  double rhoL = UL[0], uL = UL[1], vL = UL[2], wL = UL[3], betaL = UL[4];
  double rhoR = UR[0], uR = UR[1], vR = UR[2], wR = UR[3], betaR = UR[4];
  double rholog = oracle_logmean(rhoL, rhoR, logL[0], logR[0]);
  double betalog = oracle_logmean(betaL, betaR, logL[1], logR[1]);
  double rhoavg = .5 * (rhoL + rhoR);
  double uavg = .5 * (uL + uR);
  double vavg = .5 * (vL + vR);
  double wavg = .5 * (wL + wR);
  double unorm = uL * uR + vL * vR + wL * wR;
  double pa = rhoavg / (betaL + betaR);
  double E_plus_p = rholog / (2 * (GAMMA - 1) * betalog) + pa + .5 * rholog * unorm;
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

/* euler_variables.jl:79-92, 5 fields */
void oracle_v_ufun_3d(const double* U, double* V) {
  double rho = U[0], E = U[4];
  double rhoe = E - .5 * (U[1] * U[1] + U[2] * U[2] + U[3] * U[3]) / rho;
  double sU = log((GAMMA - 1) * rhoe / pow(rho, GAMMA));
  V[0] = (-E + rhoe * (GAMMA + 1 - sU)) / rhoe;
  V[1] = U[1] / rhoe;
  V[2] = U[2] / rhoe;
  V[3] = U[3] / rhoe;
  V[4] = (-rho) / rhoe;
}

/* euler_variables.jl:95-120, 5 fields */
void oracle_u_vfun_3d(const double* V, double* U) {
  double v5 = V[4];
  double vUnorm = V[1] * V[1] + V[2] * V[2] + V[3] * V[3];
  double s = GAMMA - V[0] + vUnorm / (2 * v5);
  double rhoeV = pow((GAMMA - 1) / pow(-v5, GAMMA), 1 / (GAMMA - 1)) * exp(-s / (GAMMA - 1));
  U[0] = rhoeV * (-v5);
  U[1] = rhoeV * V[1];
  U[2] = rhoeV * V[2];
  U[3] = rhoeV * V[3];
  U[4] = rhoeV * (1 - vUnorm / (2 * v5));
}

/* euler_variables.jl:30-48, 5 fields */
double oracle_betafun_3d(const double* U) {
  double rhounorm = (U[1] * U[1] + U[2] * U[2] + U[3] * U[3]) / U[0];
  double p = (GAMMA - 1) * (U[4] - .5 * rhounorm);
  return U[0] / (2 * p);
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
static void sparse_hadamard_sum_hex(const oracle_hex_t* c, const double* Qhe /*[5][Nh]*/, const double* ge /*[9][Nh]*/,
                                    double* out /*[5][Nh]*/) {
  const int Nh = c->Nq + c->Nfq;
  double lrho[512], lbeta[512];
  for (int i = 0; i < Nh; ++i) {
    lrho[i] = log(Qhe[i]);
    lbeta[i] = log(Qhe[4 * Nh + i]);
  }
  for (int i = 0; i < Nh; ++i) {
    double Qi[5], li[2] = {lrho[i], lbeta[i]}, rhsi[5] = {0, 0, 0, 0, 0};
    for (int f = 0; f < 5; ++f) Qi[f] = Qhe[f * Nh + i];
    for (int t = c->rowptr[i]; t < c->rowptr[i + 1]; ++t) {
      const int j = c->colidx[t];
      double Qj[5], lj[2] = {lrho[j], lbeta[j]}, g[9], Fx[5], Fy[5], Fz[5];
      for (int f = 0; f < 5; ++f) Qj[f] = Qhe[f * Nh + j];
      for (int m = 0; m < 9; ++m) g[m] = .5 * (ge[m * Nh + i] + ge[m * Nh + j]);
      oracle_euler_fluxes_3d(Qi, Qj, li, lj, Fx, Fy, Fz);
      for (int f = 0; f < 5; ++f) {
        double Fr = g[0] * Fx[f] + g[3] * Fy[f] + g[6] * Fz[f];
        double Fs = g[1] * Fx[f] + g[4] * Fy[f] + g[7] * Fz[f];
        double Ft = g[2] * Fx[f] + g[5] * Fy[f] + g[8] * Fz[f];
        rhsi[f] += c->Qr[i * Nh + j] * Fr + c->Qs[i * Nh + j] * Fs + c->Qt[i * Nh + j] * Ft;
      }
    }
    for (int f = 0; f < 5; ++f) out[f * Nh + i] = rhsi[f];
  }
}

/* rhs, dg3D_euler_hex.jl:167-222.  Q, rhs: [5][K][Nq].  Returns rhstest (0 unless compute_rhstest). */
double oracle_hex_rhs(const oracle_hex_t* c, const double* Q, int compute_rhstest, double* rhs) {
  const int K = c->K, Nq = c->Nq, Nfq = c->Nfq, Nh = Nq + Nfq;
  const size_t KNq = (size_t)K * Nq, KNf = (size_t)K * Nfq, KNh = (size_t)K * Nh;
  if (Nh > 512) return NAN;
  double* VU = (double*)malloc(5 * KNq * sizeof(double));
  double* VUf = (double*)malloc(5 * KNf * sizeof(double));
  double* Uf = (double*)malloc(5 * KNf * sizeof(double));
  double* Qh = (double*)malloc(5 * KNh * sizeof(double));
  double* lam = (double*)malloc(KNf * sizeof(double));
  double* flux = (double*)malloc(5 * KNf * sizeof(double));
  /* :174-176 */
#pragma omp parallel for num_threads(g_threads) schedule(static)
  for (size_t n = 0; n < KNq; ++n) {
    double U[5], V[5];
    for (int f = 0; f < 5; ++f) U[f] = Q[f * KNq + n];
    oracle_v_ufun_3d(U, V);
    for (int f = 0; f < 5; ++f) VU[f * KNq + n] = V[f];
  }
  for (int f = 0; f < 5; ++f) matmul_elems(c->Ef, Nfq, Nq, VU + f * KNq, VUf + f * KNf, K);
#pragma omp parallel for num_threads(g_threads) schedule(static)
  for (size_t n = 0; n < KNf; ++n) {
    double V[5], U[5];
    for (int f = 0; f < 5; ++f) V[f] = VUf[f * KNf + n];
    oracle_u_vfun_3d(V, U);
    for (int f = 0; f < 5; ++f) Uf[f * KNf + n] = U[f];
  }
  /* :177-182 Uh = vcat(Q,Uf); beta; Qh = (rho,u,v,w,beta) */
#pragma omp parallel for num_threads(g_threads) schedule(static)
  for (int e = 0; e < K; ++e)
    for (int i = 0; i < Nh; ++i) {
      double U[5];
      for (int f = 0; f < 5; ++f)
        U[f] = i < Nq ? Q[f * KNq + (size_t)e * Nq + i] : Uf[f * KNf + (size_t)e * Nfq + (i - Nq)];
      size_t o = (size_t)e * Nh + i;
      Qh[o] = U[0];
      Qh[KNh + o] = U[1] / U[0];
      Qh[2 * KNh + o] = U[2] / U[0];
      Qh[3 * KNh + o] = U[3] / U[0];
      Qh[4 * KNh + o] = oracle_betafun_3d(U);
    }
  /* :189-192 lam */
#pragma omp parallel for num_threads(g_threads) schedule(static)
  for (size_t n = 0; n < KNf; ++n) {
    double rhoU_n = (Uf[KNf + n] * c->nxJ[n] + Uf[2 * KNf + n] * c->nyJ[n] + Uf[3 * KNf + n] * c->nzJ[n]) / c->sJ[n];
    lam[n] = fabs(oracle_wavespeed(Uf[n], rhoU_n, Uf[4 * KNf + n]));
  }
  /* :185-198 QM/QP, LFc, surface flux */
#pragma omp parallel for num_threads(g_threads) schedule(static)
  for (int e = 0; e < K; ++e)
    for (int i = 0; i < Nfq; ++i) {
      size_t n = (size_t)e * Nfq + i;
      size_t p = (size_t)(c->mapP[n] - 1);
      size_t ep = p / Nfq, ip = p % Nfq;
      size_t om = (size_t)e * Nh + Nq + i, op = ep * Nh + Nq + ip;
      double QM[5], QP[5], Fx[5], Fy[5], Fz[5];
      for (int f = 0; f < 5; ++f) {
        QM[f] = Qh[f * KNh + om];
        QP[f] = Qh[f * KNh + op];
      }
      double lM[2] = {log(QM[0]), log(QM[4])}, lP[2] = {log(QP[0]), log(QP[4])};
      oracle_euler_fluxes_3d(QM, QP, lM, lP, Fx, Fy, Fz);
      double LFc = c->lf_scale * fmax(lam[n], lam[p]) * c->sJ[n];
      for (int f = 0; f < 5; ++f)
        flux[f * KNf + n] = Fx[f] * c->nxJ[n] + Fy[f] * c->nyJ[n] + Fz[f] * c->nzJ[n] - LFc * (Uf[f * KNf + p] - Uf[f * KNf + n]);
    }
  for (int f = 0; f < 5; ++f) matmul_elems(c->Lf, Nq, Nfq, flux + f * KNf, rhs + f * KNq, K);
  /* :200-210 volume loop */
#pragma omp parallel for num_threads(g_threads) schedule(static)
  for (int e = 0; e < K; ++e) {
    double* Qhe = (double*)malloc((size_t)(5 + 9 + 5) * Nh * sizeof(double));
    double* ge = Qhe + 5 * Nh;
    double* QFe = ge + 9 * Nh;
    for (int f = 0; f < 5; ++f)
      for (int i = 0; i < Nh; ++i) Qhe[f * Nh + i] = Qh[f * KNh + (size_t)e * Nh + i];
    for (int m = 0; m < 9; ++m)
      for (int i = 0; i < Nh; ++i) ge[m * Nh + i] = c->vgeo[m][(size_t)e * Nh + i];
    sparse_hadamard_sum_hex(c, Qhe, ge, QFe);
    for (int f = 0; f < 5; ++f)
      for (int i = 0; i < Nq; ++i) {
        double s = 0.0;
        for (int j = 0; j < Nh; ++j) s += c->Ph[i * Nh + j] * QFe[f * Nh + j];
        rhs[f * KNq + (size_t)e * Nq + i] += s;
      }
    free(Qhe);
  }
  /* :212 */
  for (int f = 0; f < 5; ++f)
    for (size_t n = 0; n < KNq; ++n) rhs[f * KNq + n] = -rhs[f * KNq + n] / c->J[n];
  double rhstest = 0.0;
  if (compute_rhstest)
    for (int f = 0; f < 5; ++f)
      for (size_t n = 0; n < KNq; ++n) rhstest += c->wJq[n] * VU[f * KNq + n] * rhs[f * KNq + n];
  free(VU); free(VUf); free(Uf); free(Qh); free(lam); free(flux);
  return rhstest;
}
