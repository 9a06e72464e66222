// esdg_kernels_tensor.hip -- tensor-product kernels for gfx950 (MI355X / CDNA4), the production path
// whenever the driver's operators factor into 1D tables (esdg_tensor_tables.hpp; always the case for
// init_reference_quad with a Gauss rule -- verified entry by entry in esdg_api.hip, otherwise the generic
// pair-list kernels of esdg_kernels.hip run).
//
// Same algorithm and face-trace protocol as esdg_kernels.hip (reference citations there).  Mapping:
//   * one wave (64 lanes) per workgroup owns E = 64/Nq elements; lane <-> (element, Gauss node), the
//     first E*Nfq lanes double as face-node lanes.  State load / rhs store: one coalesced 8-byte
//     access per lane and field.
//   * the small dense 1D operators (skew SBP matrix, face interpolation, derivative, LGL<->Gauss
//     interpolation; ~2.7 KB at N=4) are staged in LDS once per workgroup; every node / partner /
//     face id is arithmetic on (a, b) -- no per-node index tables are read from memory.
//   * Vq / Pq by sum factorisation through LDS.
//   * flux differencing walks the 2*N1 tensor lines.  Per direction a lane evaluates N1/2 forward
//     volume pairs (circulant schedule: each unordered pair once) and its 2 face pairs with its own node
//     in registers; the partner's share is pushed with ds_add_f64 into per-node LDS accumulators (one
//     wave per workgroup => the accumulation order is fixed by the instruction stream: bitwise
//     reproducible).  200 EC fluxes per element at N=4, none duplicated.
//   * EC flux: ONE refined v_rcp_f64 for its three quotients, the reference's |f|<1e-4 series branch
//     selected without divergence.  Pointwise work stays in (rho,u,v,beta,log rho,log beta); entropy
//     variables follow algebraically; traces carry the same six numbers + lam + E (64-byte records), so
//     consumers do no transcendental work on traces.
//   * all global loads of a workgroup (state, own and neighbour traces, tables) are issued before any
//     arithmetic.
#include "esdg_dev.hpp"
#include "esdg_tensor_tables.hpp"
#include "esdg_devmath.hpp"

namespace esdg {

constexpr int TW = 64;  // lanes per wave

// Work decomposition.  A GROUP of GW consecutive waves shares E elements: lane t of the group owns volume node t % Nq
// of element t / Nq (and face node t % Nfq of element t / Nfq), so an element may straddle two waves of its group.
// N1 = 5 (N = 4): 5 elements on 125 of 128 lanes instead of 2 on 50 of 64; N1 = 6: 7 on 252 of 256 instead of 1 on 36
// of 64.  A workgroup holds NWV waves = NWV/GW groups, each with its own LDS slice; the 1D tables are shared.  All
// cross-lane traffic goes through LDS between __syncthreads(), so the mapping is free; only the ds_add_f64
// accumulations need care (see kt_rhs).
// Per kernel: GW = waves per group, NWV = waves per workgroup (same-box measurements, profiles/experiments/README.md):
//   N1 = 5: kt_project gains 12 % from two-wave groups, kt_sigma and the inviscid kt_rhs lose 4-6 % (LDS-side bound:
//   the LDS pipes do not care how many lanes of an instruction are idle), the viscous kt_rhs gains a few per cent.
template <int N1> struct TCfg {
  static constexpr int P_GW = 1, P_NWV = 4, S_GW = 1, S_NWV = 4;
  static constexpr int r_gw(bool) { return 1; }
  static constexpr int r_nwv(bool) { return 4; }
};
#ifndef ESDG_N5_CFG
#define ESDG_N5_CFG 2, 4, 1, 4, 2, 2, 1, 4
#endif
template <> struct TCfg<5> {
  static constexpr int cfg[8] = {ESDG_N5_CFG};   // P_GW, P_NWV, S_GW, S_NWV, R_GW/R_NWV viscous, R_GW/R_NWV inviscid
  static constexpr int P_GW = cfg[0], P_NWV = cfg[1], S_GW = cfg[2], S_NWV = cfg[3];
  static constexpr int r_gw(bool visc) { return visc ? cfg[4] : cfg[6]; }
  static constexpr int r_nwv(bool visc) { return visc ? cfg[5] : cfg[7]; }
};
template <> struct TCfg<6> {
  static constexpr int P_GW = 4, P_NWV = 4, S_GW = 4, S_NWV = 4;
  static constexpr int r_gw(bool) { return 4; }
  static constexpr int r_nwv(bool) { return 4; }
};
template <> struct TCfg<7> {   // kt_rhs: one element per wave is faster than 5 elements on 4 waves (A/B at N = 6)
  static constexpr int P_GW = 4, P_NWV = 4, S_GW = 4, S_NWV = 4;
  static constexpr int r_gw(bool) { return 1; }
  static constexpr int r_nwv(bool) { return 4; }
};

template <int N1, int GW_> struct Grp {
  static constexpr int GW = GW_;
  static constexpr int GT = TW * GW;      // lanes per group
  static constexpr int E = (GT / (N1 * N1)) < (GT / (4 * N1)) ? (GT / (N1 * N1)) : (GT / (4 * N1));   // elements per group
  static_assert(E >= 1, "group does not hold an element");
  // element e of a group straddles two waves (its lanes [e*Nq, (e+1)*Nq) cross a multiple of 64)
  __host__ __device__ static constexpr bool strad(int e) { return (e * N1 * N1) / TW != ((e + 1) * N1 * N1 - 1) / TW; }
  __host__ __device__ static constexpr int nstrad() {
    int n = 0;
    for (int e = 0; e < E; ++e) n += strad(e) ? 1 : 0;
    return n;
  }
  static constexpr int NS = GW > 1 ? nstrad() : 0;
};
template <int N1, int GW_, int NWV> struct Wg {
  static_assert(NWV % GW_ == 0, "workgroup = whole groups");
  using G = Grp<N1, GW_>;
  static constexpr int GW = GW_;
  static constexpr int NG = NWV / GW;     // groups per workgroup
  static constexpr int TPB = TW * NWV;    // threads per workgroup
  static constexpr int EPB = G::E * NG;   // elements per workgroup
};
template <int N1> struct WgP : Wg<N1, TCfg<N1>::P_GW, TCfg<N1>::P_NWV> {};                              // kt_project
template <int N1> struct WgS : Wg<N1, TCfg<N1>::S_GW, TCfg<N1>::S_NWV> {};                              // kt_sigma
template <int N1, bool VISC> struct WgR : Wg<N1, TCfg<N1>::r_gw(VISC), TCfg<N1>::r_nwv(VISC)> {};        // kt_rhs

namespace tdev {

using namespace devmath;

template <bool MODAL> struct Gas {
  static constexpr double GM1 = MODAL ? 0.4 : (1.4 - 1);  // literal 0.4 in the CNS drivers, gamma-1 in the Euler one
};

// Entropy-conservative flux (euler_fluxes.jl:23-48 with logmean.jl:14-28), q = (rho,u,v,beta,lrho,lbeta).
// One reciprocal serves rho log-mean, 1/(beta log-mean) and pa; the reference's series branch for
// |f| < 1e-4 is selected, not branched; 1/P(v) of that branch is expanded to 1 + .2v + .0912v^2
// (v < 1e-8: truncation < 1e-24).
template <bool MODAL>
__device__ __forceinline__ void ec_flux(const double* qL, const double* qR, double* Fx, double* Fy) {
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
  const double ir = R * ybp;   // 1/yr
  const double ryr = R * yr;
  const double ib = ryr * yp;  // 1/yb
  const double ip = ryr * yb;  // 1/(betaL+betaR)
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

// (gx,gy) . (Fx,Fy) of the same flux: what flux differencing and the interface flux actually need (the weight
// or normal is folded into g), 10 multiply-adds fewer than forming Fx and Fy and contracting them afterwards
template <bool MODAL>
__device__ __forceinline__ void ec_flux_dir(const double* qL, const double* qR, double gx, double gy, double* F) {
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
  const double ir = R * ybp;
  const double ryr = R * yr;
  const double ib = ryr * yp;
  const double ip = ryr * yb;
  const double fr = dr * ir, vr = fr * fr;
  const double rholog = ser_r ? ravg * (1 + vr * (-.2 - vr * (.0512 - vr * 0.026038857142857))) : -fr;
  const double fb = db * ib, vb = fb * fb;
  const double ibetalog = ser_b ? ib * (1 + vb * (.2 + vb * .0912)) : -(B * ib);
  const double uavg = .5 * (qL[1] + qR[1]), vavg = .5 * (qL[2] + qR[2]);
  const double unorm = qL[1] * qR[1] + qL[2] * qR[2];
  const double pa = ravg * ip;
  const double f4aux = rholog * ibetalog * (1.0 / (2 * GM1)) + pa + .5 * rholog * unorm;
  const double un = gx * uavg + gy * vavg;
  F[0] = rholog * un;
  F[1] = F[0] * uavg + pa * gx;
  F[2] = F[0] * vavg + pa * gy;
  F[3] = f4aux * un;
}

// conservative -> (rho,u,v,beta,log rho,log beta)  (betafun euler_variables.jl:30-48 / cavity :484);
// 1/rho and 1/rhoe from one reciprocal
template <bool MODAL>
__device__ __forceinline__ void prim_logs(const double* U, double* q) {
  constexpr double GM1 = Gas<MODAL>::GM1;
  const double m2 = U[1] * U[1] + U[2] * U[2];
  // rhoe*rho = rho*E - m2/2
  const double rre = U[0] * U[3] - .5 * m2;
  const double R = rcp_refined(U[0] * rre);   // 1/(rho^2 rhoe)
  const double ir = R * rre;                  // 1/rho
  q[0] = U[0];
  q[1] = U[1] * ir;
  q[2] = U[2] * ir;
  q[3] = (U[0] * U[0]) * (U[0] * R) * (1.0 / (2 * GM1));   // rho/(2 GM1 rhoe) = rho^3 R /(2 GM1)
  q[4] = log_pos(U[0]);
  q[5] = log_pos(q[3]);
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
  const double h = vUnorm * .5 * rcp_refined(V[3]);
  const double s = 1.4 - V[0] + h;
  const double rhoeV = exp((log(GM1) - 1.4 * log_pos(-V[3]) - s) * (1.0 / GM1));
  U[0] = rhoeV * (-V[3]);
  U[1] = rhoeV * V[1];
  U[2] = rhoeV * V[2];
  U[3] = rhoeV * (1 - h);
}

// wavespeed (euler_variables.jl:7-10, sqrt(|u_n|) quirk Q1 / cavity :507)
template <bool MODAL>
__device__ __forceinline__ double lf_lambda(const double* U, double nxJ, double nyJ, double sJ) {
  constexpr double GM1 = Gas<MODAL>::GM1;
  const double ir = rcp_refined(U[0]);
  const double rhoUn = (U[1] * nxJ + U[2] * nyJ) * rcp_refined(sJ);
  const double p = GM1 * (U[3] - .5 * (rhoUn * rhoUn) * ir);
  return fabs(sqrt(fabs(rhoUn * ir)) + sqrt(1.4 * p * ir));
}

// viscous_matrices! + sigma rows 2..4 (cavity :613-645, 786-801); lam already sign-flipped (quirk Q4);
// gk = gamma*mu/Pr
__device__ __forceinline__ void viscous_stress(const double* v, const double* tx, const double* ty, double lam,
                                               double mu, double gk, double* sx, double* sy) {
  const double v2 = v[0], v3 = v[1], v4 = v[2];
  const double iv = rcp_refined(v4);
  const double i2 = iv * iv, i1 = iv;  // 1/v4^2, 1/v4  (inv*v4^2 = 1/v4, inv*v4 = 1/v4^2)
  const double i3 = i2 * iv;
  const double l2m = lam + 2.0 * mu;
  const double Kxx22 = -l2m * i1, Kxx24 = l2m * v2 * i2, Kxx33 = -mu * i1, Kxx34 = mu * v3 * i2,
               Kxx44 = -i3 * (l2m * (v2 * v2) + mu * (v3 * v3) - gk * v4);
  const double Kxy23 = -lam * i1, Kxy24 = lam * v3 * i2, Kxy32 = -mu * i1, Kxy34 = mu * v2 * i2,
               Kxy42 = mu * v3 * i2, Kxy43 = lam * v2 * i2, Kxy44 = i3 * (lam + mu) * (-v2 * v3);
  const double Kyy22 = -mu * i1, Kyy24 = mu * v2 * i2, Kyy33 = -l2m * i1, Kyy34 = l2m * v3 * i2,
               Kyy44 = -i3 * (l2m * (v3 * v3) + mu * (v2 * v2) - gk * v4);
  sx[0] = Kxx22 * tx[0] + Kxx24 * tx[2] + Kxy23 * ty[1] + Kxy24 * ty[2];
  sx[1] = Kxx33 * tx[1] + Kxx34 * tx[2] + Kxy32 * ty[0] + Kxy34 * ty[2];
  sx[2] = Kxx24 * tx[0] + Kxx34 * tx[1] + Kxx44 * tx[2] + Kxy42 * ty[0] + Kxy43 * ty[1] + Kxy44 * ty[2];
  sy[0] = Kxy32 * tx[1] + Kxy42 * tx[2] + Kyy22 * ty[0] + Kyy24 * ty[2];
  sy[1] = Kxy23 * tx[0] + Kxy43 * tx[2] + Kyy33 * ty[1] + Kyy34 * ty[2];
  sy[2] = Kxy24 * tx[0] + Kxy34 * tx[1] + Kxy44 * tx[2] + Kyy24 * ty[0] + Kyy34 * ty[1] + Kyy44 * ty[2];
}

// lane geometry -------------------------------------------------------------------------------
template <int N1, int GW>
struct Lane {
  static constexpr int Nq = N1 * N1, Nfq = 4 * N1, E = Grp<N1, GW>::E, GT = Grp<N1, GW>::GT;
  int tid, grp, ev, q, a, b, ef, fn;   // tid: lane within the group; grp: group within the workgroup
  bool vin, fin;  // lane owns a volume-node / face-node slot of its group
  __device__ __forceinline__ Lane() {
    tid = threadIdx.x & (GT - 1);
    // a wave lies in one group: keep grp (and with it e0 and the LDS slice bases) in scalar registers
    grp = __builtin_amdgcn_readfirstlane((int)(threadIdx.x / GT));
    const int e = tid / Nq;
    q = tid - e * Nq;
    vin = tid < E * Nq;
    ev = vin ? e : 0;
    a = q % N1;
    b = q / N1;
    const int f = tid / Nfq;
    fn = tid - f * Nfq;
    fin = tid < E * Nfq;
    ef = fin ? f : 0;
  }
  __device__ __forceinline__ int pos(int d) const { return d == 0 ? a : b; }
  __device__ __forceinline__ int oth(int d) const { return d == 0 ? b : a; }
};

template <int N1>
__device__ __forceinline__ int node_of(int d, int i, int o) { return d == 0 ? i + N1 * o : o + N1 * i; }

// copy the 1D tables to LDS
template <int N1>
__device__ __forceinline__ void stage_tables(const TensorTables& TT, double* sTab, int* sInt) {
  constexpr TensorLayout L(N1);
  for (int i = threadIdx.x; i < L.NDBL; i += (int)blockDim.x) sTab[i] = TT.dbl[i];
  for (int i = threadIdx.x; i < L.NINT; i += (int)blockDim.x) sInt[i] = TT.ints[i];
}

// The same copy in two halves for kernels that have other loads in flight: load() issues every table load into registers
// without waiting (all lanes, index clamped: lanes beyond the table re-read its last entry), store() writes them to LDS
// (duplicate lanes write the same value to the same slot).  The strided copy above -- like any LDS store under
// `if (lane < n)` -- makes hipcc put each load next to its store and wait for it with vmcnt(0), i.e. for every load the
// wave has in flight, once per round of the copy (tools/asm_waits.sh shows the waits of a kernel).
template <int N1, int TPB>
struct TableRegs {
  static constexpr TensorLayout L = TensorLayout(N1);
  static constexpr int ND = (L.NDBL + TPB - 1) / TPB, NI = (L.NINT + TPB - 1) / TPB;
  double d[ND];
  int i[NI];
  __device__ __forceinline__ void load(const TensorTables& TT) {
#pragma unroll
    for (int r = 0; r < ND; ++r) d[r] = TT.dbl[min((int)threadIdx.x + r * TPB, L.NDBL - 1)];
#pragma unroll
    for (int r = 0; r < NI; ++r) i[r] = TT.ints[min((int)threadIdx.x + r * TPB, L.NINT - 1)];
  }
  __device__ __forceinline__ void store(double* sTab, int* sInt) const {
#pragma unroll
    for (int r = 0; r < ND; ++r) sTab[min((int)threadIdx.x + r * TPB, L.NDBL - 1)] = d[r];
#pragma unroll
    for (int r = 0; r < NI; ++r) sInt[min((int)threadIdx.x + r * TPB, L.NINT - 1)] = i[r];
  }
};

template <int N1>
__device__ __forceinline__ void issue_state_loads(const double* __restrict__ Q, int64_t K, int64_t e0, bool active,
                                                  int tid, double* x) {
  constexpr int Nq = N1 * N1;
  x[0] = 1.0; x[1] = 0.0; x[2] = 0.0; x[3] = 1.0;
  if (active) {
#pragma unroll
    for (int f = 0; f < 4; ++f) x[f] = Q[(int64_t)f * K * Nq + e0 * Nq + tid];
  }
}

// Uq = (IQ (x) IQ) Qn by sum factorisation; sA/sB: LDS scratch [E][4][Nq] each.  x -> U.
template <int N1, bool MODAL, class LN>
__device__ __forceinline__ void state_at_quad(const LN& ln, const double* sTab, double* sA, double* sB,
                                              const double* x, double* U) {
  constexpr int Nq = N1 * N1;
  constexpr TensorLayout L(N1);
  if (!MODAL) {
#pragma unroll
    for (int f = 0; f < 4; ++f) U[f] = x[f];
    return;
  }
  const int lo = ln.a, hi = ln.b, ev = ln.ev, q = ln.q;
  double c[N1];
#pragma unroll
  for (int i = 0; i < N1; ++i) c[i] = sTab[L.IQ + lo * N1 + i];
  if (ln.vin) {
#pragma unroll
    for (int f = 0; f < 4; ++f) sA[(ev * 4 + f) * Nq + q] = x[f];
  }
  __syncthreads();
  // stage 1: W[b + N1 j] = sum_i IQ[b,i] Qn[i + N1 j]   (this lane: b = lo, j = hi)
#pragma unroll
  for (int f = 0; f < 4; ++f) {
    const double* src = sA + (ev * 4 + f) * Nq + N1 * hi;
    double s = 0.0;
#pragma unroll
    for (int i = 0; i < N1; ++i) s += c[i] * src[i];
    if (ln.vin) sB[(ev * 4 + f) * Nq + q] = s;
  }
  __syncthreads();
  // stage 2: Uq[a + N1 b] = sum_j IQ[a,j] W[b + N1 j]   (this lane: a = lo, b = hi)
#pragma unroll
  for (int f = 0; f < 4; ++f) {
    const double* src = sB + (ev * 4 + f) * Nq + hi;
    double s = 0.0;
#pragma unroll
    for (int j = 0; j < N1; ++j) s += c[j] * src[N1 * j];
    U[f] = s;
  }
}

// The same for kt_project, whose state loads are issued at entry: EVERY lane stages its x (slot sl: its own, or for the few
// lanes beyond the group's volume slots that of lane tid - E*Nq, whose data they loaded too -- a duplicate write).  With the
// store under `if (ln.vin)` hipcc sinks the global loads into that branch, below the table barrier.
template <int N1, bool MODAL, class LN>
__device__ __forceinline__ void state_at_quad_dup(const LN& ln, const double* sTab, double* sA, double* sB, const double* x,
                                                  int sl_e, int sl_q, double* U) {
  constexpr int Nq = N1 * N1;
  constexpr TensorLayout L(N1);
  if (!MODAL) {
#pragma unroll
    for (int f = 0; f < 4; ++f) U[f] = x[f];
    return;
  }
  const int lo = ln.a, hi = ln.b, ev = ln.ev, q = ln.q;
#pragma unroll
  for (int f = 0; f < 4; ++f) sA[(sl_e * 4 + f) * Nq + sl_q] = x[f];
  double c[N1];
#pragma unroll
  for (int i = 0; i < N1; ++i) c[i] = sTab[L.IQ + lo * N1 + i];
  __syncthreads();
#pragma unroll
  for (int f = 0; f < 4; ++f) {
    const double* src = sA + (ev * 4 + f) * Nq + N1 * hi;
    double s = 0.0;
#pragma unroll
    for (int i = 0; i < N1; ++i) s += c[i] * src[i];
    if (ln.vin) sB[(ev * 4 + f) * Nq + q] = s;
  }
  __syncthreads();
#pragma unroll
  for (int f = 0; f < 4; ++f) {
    const double* src = sB + (ev * 4 + f) * Nq + hi;
    double s = 0.0;
#pragma unroll
    for (int j = 0; j < N1; ++j) s += c[j] * src[N1 * j];
    U[f] = s;
  }
}

// out = (IP (x) IP) R by sum factorisation, then the coalesced store
template <int N1, bool MODAL, class LN>
__device__ __forceinline__ void store_rhs_from_quad(const LN& ln, const double* sTab, double* __restrict__ rhs,
                                                    const LsrkFuse& lf, int64_t K, int64_t e0, bool active, double* sA,
                                                    double* sB, const double* R) {
  constexpr int Nq = N1 * N1;
  constexpr TensorLayout L(N1);
  double out[4];
  if (!MODAL) {
#pragma unroll
    for (int f = 0; f < 4; ++f) out[f] = R[f];
  } else {
    const int lo = ln.a, hi = ln.b, ev = ln.ev, q = ln.q;
    if (ln.vin) {
#pragma unroll
      for (int f = 0; f < 4; ++f) sA[(ev * 4 + f) * Nq + q] = R[f];
    }
    __syncthreads();
    // stage 1: W[i + N1 a] = sum_b IP[i,b] R[a + N1 b]   (this lane: i = lo, a = hi)
#pragma unroll
    for (int f = 0; f < 4; ++f) {
      const double* src = sA + (ev * 4 + f) * Nq + hi;
      double s = 0.0;
#pragma unroll
      for (int bb = 0; bb < N1; ++bb) s += sTab[L.IP + lo * N1 + bb] * src[N1 * bb];
      if (ln.vin) sB[(ev * 4 + f) * Nq + q] = s;
    }
    __syncthreads();
    // stage 2: out[i + N1 j] = sum_a IP[j,a] W[i + N1 a]   (this lane: i = lo, j = hi)
#pragma unroll
    for (int f = 0; f < 4; ++f) {
      const double* src = sB + (ev * 4 + f) * Nq + lo;
      double s = 0.0;
#pragma unroll
      for (int aa = 0; aa < N1; ++aa) s += sTab[L.IP + hi * N1 + aa] * src[N1 * aa];
      out[f] = s;
    }
  }
  if (active) {
#pragma unroll
    for (int f = 0; f < 4; ++f) {
      const int64_t idx = (int64_t)f * K * Nq + e0 * Nq + ln.tid;
      if (lf.Qw) {   // fused low-storage RK stage
        const double r = __builtin_fma(lf.a, lf.res[idx], lf.dt * out[f]);
        lf.res[idx] = r;
        lf.Qw[idx] = __builtin_fma(lf.b, r, lf.Qw[idx]);
      } else {
        rhs[idx] = out[f];
      }
    }
  }
}

// (d,t,o) of a face node
__device__ __forceinline__ void face_dto(const int* sInt, int finv_off, int fn, int& d, int& t, int& o) {
  const int w = sInt[finv_off + fn];
  d = w & 1;
  t = (w >> 1) & 1;
  o = w >> 2;
}

}  // namespace tdev

using namespace tdev;

// ---------------------------------------------------------------------------------------------
// phase 0: entropy projection to the faces -> A_U (rho,u,v,beta,lrho,lbeta,lam,E).
// The viscous path needs the neighbour's projected entropy variables Vf*VU = Ef*v(u_q) (rhs_viscous! :771-776):
// they are the entropy variables OF this trace state (u_f = u(Ef v)), so consumers rebuild (v2,v3,v4) =
// (b u, b v, -b), b = 2 (gamma-1) beta, from the A_U record instead of reading a second trace buffer.
// ---------------------------------------------------------------------------------------------
template <int N1, bool MODAL, bool VISC>
__global__ __launch_bounds__(WgP<N1>::TPB) void kt_project(TensorTables TT, MeshDev M, Phys ph, const double* __restrict__ Q,
                                                 double* __restrict__ A_U, double* __restrict__ A_v) {
  using W = WgP<N1>;
  constexpr int Nq = N1 * N1, Nfq = 4 * N1, E = W::G::E;
  constexpr TensorLayout L(N1);
  __shared__ double sTab[L.NDBL];
  __shared__ int sInt[L.NINT];
  __shared__ double sA_[W::NG * E * 4 * Nq];
  __shared__ double sB_[W::NG * E * 4 * Nq];
  const Lane<N1, W::GW> ln;
  double* sA = sA_ + ln.grp * (E * 4 * Nq);
  double* sB = sB_ + ln.grp * (E * 4 * Nq);
  const int64_t e0 = M.e_begin + ((int64_t)blockIdx.x * W::NG + ln.grp) * E;
  const int nE = (int)max((int64_t)0, min((int64_t)E, M.e_begin + M.e_count - e0));
  const bool vactive = ln.tid < nE * Nq, factive = ln.tid < nE * Nfq;
  const int64_t e0s = min(e0, M.e_begin + M.e_count - 1);   // in-range base for the reads of idle lanes / idle groups

  // All global loads of the kernel are issued here, unconditionally and before anything is waited for: tables and state
  // (the strided table copy -> barrier -> state sequence cost the wave two serial memory round trips; until round 3 a third
  // one for the face normals of the wavespeed, which the consumer of the trace now rebuilds).  Idle lanes read node 0 of an
  // in-range element.
  TableRegs<N1, W::TPB> tr;
  tr.load(TT);
  double x[4];
  const int ta = ln.vin ? ln.tid : ln.tid - E * Nq;   // lanes beyond the group's E*Nq volume slots duplicate lane tid - E*Nq
  const int sl_e = ta / Nq, sl_q = ta - sl_e * Nq;
  {
    const bool va = ta < nE * Nq;                      // (slots of elements beyond the mesh: node 0 of an in-range element)
    const int64_t eb = ESDG_EW(va ? e0 : e0s);
    const int tq = va ? ta : 0;
#pragma unroll
    for (int f = 0; f < 4; ++f) x[f] = Q[(int64_t)f * M.K * Nq + eb * Nq + tq];
  }
  __builtin_amdgcn_sched_barrier(0);   // (hipcc otherwise moves the state loads below the barrier, next to their first use)
  tr.store(sTab, sInt);
  __syncthreads();
  double U[4];
  state_at_quad_dup<N1, MODAL>(ln, sTab, sA, sB, x, sl_e, sl_q, U);
  double qh[6], V[4];
  prim_logs<MODAL>(U, qh);
  v_of_prim<MODAL>(qh, V);
  __syncthreads();
  if (ln.vin) {
#pragma unroll
    for (int c = 0; c < 4; ++c) sA[(ln.ev * 4 + c) * Nq + ln.q] = V[c];
  }
  __syncthreads();
  if (factive) {
    int d, t, o;
    face_dto(sInt, L.FINV, ln.fn, d, t, o);
    double Vf[4] = {0, 0, 0, 0};
#pragma unroll
    for (int j = 0; j < N1; ++j) {
      const double w = sTab[L.EE + (d * 2 + t) * N1 + j];
      const int col = node_of<N1>(d, j, o);
#pragma unroll
      for (int c = 0; c < 4; ++c) Vf[c] += w * sA[(ln.ef * 4 + c) * Nq + col];
    }
    double Uf[4], qf[6];
    u_of_v<MODAL>(Vf, Uf);
    {   // (rho, u, v, beta) of the projected state: the whole record (its logs, energy and wavespeed are rebuilt by the consumer)
      constexpr double GM1e = Gas<MODAL>::GM1;
      const double m2 = Uf[1] * Uf[1] + Uf[2] * Uf[2];
      const double rre = Uf[0] * Uf[3] - .5 * m2;
      const double R = rcp_refined(Uf[0] * rre);
      const double ir = R * rre;
      qf[0] = Uf[0]; qf[1] = Uf[1] * ir; qf[2] = Uf[2] * ir;
      qf[3] = (Uf[0] * Uf[0]) * (Uf[0] * R) * (1.0 / (2 * GM1e));
    }
    const int64_t n = (ESDG_EW(e0) + ln.ef) * Nfq + ln.fn;
    double2* a = reinterpret_cast<double2*>(A_U + n * FAU_NC);
    a[0] = make_double2(qf[0], qf[1]);
    a[1] = make_double2(qf[2], qf[3]);
  }
  (void)A_v;
}

// ---------------------------------------------------------------------------------------------
// viscous building blocks (lane-mapped).  sVn: [E][3][Nq] = (v2,v3,v4) of the volume nodes (SoA: conflict-free).
// ---------------------------------------------------------------------------------------------
// face lanes: projected entropy variables at the face node, exterior state (neighbour, or the wall
// boundary condition of impose_BCs_entropyvars!, cavity :178-216), half jump, penalty tau*[[v]] (:817-837)
//   bc: 0 interior/periodic, 1 wall, 2 lid;  gn = (nxJ, nyJ, sJ) of the face;  pn_out (registers) may be null (phase 1)
// exterior entropy variables and penalty at one face node, given its own projected values vf = (v2,v3,v4):
// vP = neighbour's values, or the closure of impose_BCs_entropyvars! (cavity :178-216 / modalESDG :187-203);
// dV = vP - vf; pn_out (may be null) = tau*[[v]] with the boundary overrides of :817-837
__device__ __forceinline__ void face_jump_and_penalty(const double* vf, const double* vPin, int bc, double vlid,
                                                      const double* gn, const Phys& ph, double* dV, double* pn_out) {
  double vP[3] = {vPin[0], vPin[1], vPin[2]};
  if (bc >= 3) {                                          // shock-tube closures, dg2D_CNS_modalESDG.jl:187-203
#pragma unroll
    for (int c = 0; c < 3; ++c) vP[c] = bc == 3 ? ph.inflow_vv[c] : vf[c];
  } else if (bc) {
    // vlid: lid velocity at this node (ones in cavity :147, (1+cos(pi x))/2 in dg2D_CNS_convergence_test.jl:76)
    if (ph.BCTYPE == 1) {                                 // adiabatic no-slip
      vP[0] = bc == 2 ? -vf[0] - 2 * vlid * vf[2] : -vf[0];
      vP[1] = -vf[1];
      vP[2] = vf[2];
    } else if (ph.BCTYPE == 2) {                          // isothermal
      const double theta = 1.0 / (0.3 * 0.3) / 1.4 / 0.4;
      vP[0] = bc == 2 ? 2.0 / theta - vf[0] : -vf[0];
      vP[1] = -vf[1];
      vP[2] = -2.0 / theta - vf[2];
    } else {                                              // slip / reflective
      const double is = rcp_refined(gn[2]);
      const double nx = gn[0] * is, ny = gn[1] * is;
      const double vn = vf[0] * nx + vf[1] * ny;
      vP[0] = vf[0] - 2 * vn * nx;
      vP[1] = vf[1] - 2 * vn * ny;
      vP[2] = vf[2];
    }
  }
#pragma unroll
  for (int c = 0; c < 3; ++c) dV[c] = vP[c] - vf[c];
  if (pn_out) {
    const double iv4 = rcp_refined(vf[2]);
    const double tau = -iv4 * ph.inv_Re;
    pn_out[0] = tau * dV[0]; pn_out[1] = tau * dV[1]; pn_out[2] = tau * dV[2];
    if (bc) {   // :827-837
      const double a2 = .5 * (vP[0] + vf[0]), a3 = .5 * (vP[1] + vf[1]);
      double s = a2 * dV[0] + a3 * dV[1];
      if (ph.BCTYPE != 1) s += dV[2] * dV[2] * .5;
      pn_out[2] = -tau * s * iv4;
    }
  }
}

// face lanes: projected entropy variables at the face node by interpolation of the nodal ones (Vf*VU), half jump to
// sDv (may be null), penalty (may be null).  bc: 0 interior/periodic, 1 wall, 2 lid, 3 inflow, 4 copy
template <int N1, class LN>
__device__ __forceinline__ void visc_face_jumps(const LN& ln, const double* sTab, const int* sInt,
                                                const double* sVn, const double* vPin, int bc, double vlid,
                                                const double* gn, const Phys& ph, double* sDv, double* pn_out) {
  constexpr int Nq = N1 * N1, Nfq = 4 * N1;
  constexpr TensorLayout L(N1);
  int d, t, o;
  face_dto(sInt, L.FINV, ln.fn, d, t, o);
  double vf[3] = {0, 0, 0};
#pragma unroll
  for (int j = 0; j < N1; ++j) {
    const double w = sTab[L.EE + (d * 2 + t) * N1 + j];
    const double* r = sVn + ln.ef * 3 * Nq + node_of<N1>(d, j, o);
    vf[0] += w * r[0];
    vf[1] += w * r[Nq];
    vf[2] += w * r[2 * Nq];
  }
  double dV[3];
  face_jump_and_penalty(vf, vPin, bc, vlid, gn, ph, dV, pn_out);
  if (sDv) {
#pragma unroll
    for (int c = 0; c < 3; ++c) sDv[(ln.ef * 3 + c) * Nfq + ln.fn] = .5 * dV[c];
  }
}

// volume lanes: BR1 gradient of (v2,v3,v4) at the node and sigma = K(v) grad v
template <int N1, class LN>
__device__ __forceinline__ void visc_sigma(const LN& ln, const double* sTab, const int* sInt,
                                           const TensorTables& TT, const Phys& ph, const double* g, const double* fn3e,
                                           const double* sVn, const double* sDv, double* sgx, double* sgy,
                                           double* gradx = nullptr, double* grady = nullptr) {   // fn3e: MeshDev::fnrm of the element
  constexpr int Nq = N1 * N1, Nfq = 4 * N1;
  constexpr TensorLayout L(N1);
  double tx[3] = {0, 0, 0}, ty[3] = {0, 0, 0};
#pragma unroll 1
  for (int d = 0; d < 2; ++d) {
    const int op = d == 0 ? TT.op0 : TT.op1;
    const double gx = g[op], gy = g[2 + op];
    const int pos = ln.pos(d), oth = ln.oth(d);
    double dv[3] = {0, 0, 0};
#pragma unroll
    for (int j = 0; j < N1; ++j) {
      const double w = sTab[L.DG + (d * N1 + pos) * N1 + j];
      const double* r = sVn + ln.ev * 3 * Nq + node_of<N1>(d, j, oth);
      dv[0] += w * r[0];
      dv[1] += w * r[Nq];
      dv[2] += w * r[2 * Nq];
    }
#pragma unroll
    for (int c = 0; c < 3; ++c) { tx[c] += gx * dv[c]; ty[c] += gy * dv[c]; }
    // lift of the half jumps on the two faces at the ends of this line
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      const int f = sInt[L.FN + (d * 2 + t) * N1 + oth];
      const double lw = sTab[L.PF + (d * 2 + t) * N1 + pos] * sTab[L.PTF + (d * 2 + t) * N1 + oth] * sTab[L.WFAC + f];
      const double* gn = fn3e + 3 * f;   // the face node's own normal (the reference multiplies the jump by nxJ per node)
      const double lx = lw * gn[0], ly = lw * gn[1];
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        const double dj = sDv[(ln.ev * 3 + c) * Nfq + f];
        tx[c] += lx * dj;
        ty[c] += ly * dj;
      }
    }
  }
  const double iJ = rcp_refined(g[4]);
#pragma unroll
  for (int c = 0; c < 3; ++c) { tx[c] *= iJ; ty[c] *= iJ; }
  const double* r = sVn + ln.ev * 3 * Nq + ln.q;
  const double v[3] = {r[0], r[Nq], r[2 * Nq]};
  viscous_stress(v, tx, ty, -ph.lambda, ph.mu, ph.kappa, sgx, sgy);
  if (gradx) {
#pragma unroll
    for (int c = 0; c < 3; ++c) { gradx[c] = tx[c]; grady[c] = ty[c]; }
  }
}

// face lanes: own normal stress (Ef*sigma_x)*nxJ + (Ef*sigma_y)*nyJ;  sS: [E][Nq][6]
template <int N1, class LN>
__device__ __forceinline__ void face_normal_stress(const LN& ln, const double* sTab, const int* sInt,
                                                   const double* sS, double nxJ, double nyJ, double* sn,
                                                   double* fx, double* fy) {
  constexpr int Nq = N1 * N1;
  constexpr TensorLayout L(N1);
  int d, t, o;
  face_dto(sInt, L.FINV, ln.fn, d, t, o);
#pragma unroll
  for (int c = 0; c < 3; ++c) { fx[c] = 0.0; fy[c] = 0.0; }
#pragma unroll 1
  for (int j = 0; j < N1; ++j) {
    const double w = sTab[L.EE + (d * 2 + t) * N1 + j];
    const double2* r = reinterpret_cast<const double2*>(sS + (ln.ef * Nq + node_of<N1>(d, j, o)) * 6);
    const double2 s0 = r[0], s1 = r[1], s2 = r[2];
    fx[0] += w * s0.x; fx[1] += w * s0.y; fx[2] += w * s1.x;
    fy[0] += w * s1.y; fy[1] += w * s2.x; fy[2] += w * s2.y;
  }
#pragma unroll
  for (int c = 0; c < 3; ++c) sn[c] = fx[c] * nxJ + fy[c] * nyJ;
}

// ---------------------------------------------------------------------------------------------
// phase 1 (CNS): sigma = K(v) grad v, normal stress traces -> B
// ---------------------------------------------------------------------------------------------
// DIAG: also reduce visc_test = sum(wJq .* (VUx .* sigma_x + VUy .* sigma_y)) (rhs_viscous! :802-806) per workgroup
// DIVV (meshes without walls): instead of sigma itself, store the volume part of its divergence,
//   (rxJ Dr + sxJ Ds) sigma_x + (ryJ Dr + syJ Ds) sigma_y   (3 numbers per node instead of 6),
// which is all the last phase needs of sigma at the nodes (its face part comes from the normal-stress traces B)
template <int N1, bool DIAG, bool DIVV>
__global__ __launch_bounds__(WgS<N1>::TPB) void kt_sigma(TensorTables TT, MeshDev M, Phys ph, const double* __restrict__ Q,
                                               const double* __restrict__ A_U, double* __restrict__ B,
                                               double* __restrict__ SG, double* __restrict__ vt_partial) {
  using W = WgS<N1>;
  constexpr int Nq = N1 * N1, Nfq = 4 * N1, E = W::G::E;
  constexpr TensorLayout L(N1);
  __shared__ double sTab[L.NDBL];
  __shared__ int sInt[L.NINT];
  constexpr int NG = W::NG, TPB = W::TPB;
  constexpr int nB = NG * E * 6 * Nq > TPB ? NG * E * 6 * Nq : TPB;   // also holds the DIAG reduction
  __shared__ __align__(16) double sA_[NG * E * 4 * Nq];   // interp scratch, then sVn [E][Nq][4]
  __shared__ __align__(16) double sB_[nB];                // interp scratch, then sS [E][Nq][6]
  __shared__ double sDv_[NG * E * 3 * Nfq];
  const Lane<N1, W::GW> ln;
  double* sA = sA_ + ln.grp * (E * 4 * Nq);
  double* sB = sB_ + ln.grp * (E * 6 * Nq);
  double* sDv = sDv_ + ln.grp * (E * 3 * Nfq);
  stage_tables<N1>(TT, sTab, sInt);
  __syncthreads();
  const int64_t e0 = M.e_begin + ((int64_t)blockIdx.x * W::NG + ln.grp) * E;
  const int nE = (int)max((int64_t)0, min((int64_t)E, M.e_begin + M.e_count - e0));
  const bool vactive = ln.tid < nE * Nq, factive = ln.tid < nE * Nfq;
  const int64_t e0s = min(e0, M.e_begin + M.e_count - 1);   // in-range base for the geometry reads of idle lanes / idle groups

  double x[4];
  issue_state_loads<N1>(Q, M.K, e0, vactive, ln.tid, x);
  double vP[3] = {0, 0, 0};
  if (factive) {
    const double* up = A_U + (int64_t)M.mapP[(e0 + ln.ef) * Nfq + ln.fn] * FAU_NC;   // (rho,u,v,beta,...)
    const double b2 = 2 * Gas<true>::GM1 * up[3];
    vP[0] = b2 * up[1]; vP[1] = b2 * up[2]; vP[2] = -b2;
  }
  double U[4];
  state_at_quad<N1, true>(ln, sTab, sA, sB, x, U);
  double qh[6], V[4];
  prim_logs<true>(U, qh);
  v_of_prim<true>(qh, V);
  __syncthreads();
  if (ln.vin) {
#pragma unroll
    for (int c = 0; c < 3; ++c) sA[(ln.ev * 3 + c) * Nq + ln.q] = V[c + 1];
  }
  __syncthreads();
  if (ln.fin) {
    const int64_t nn = (e0 + (factive ? ln.ef : 0)) * Nfq + ln.fn;
    const int bc = (M.bc && factive) ? M.bc[nn] : 0;
    const double vlid = (bc == 2 && M.vlid) ? M.vlid[nn] : 1.0;
    visc_face_jumps<N1>(ln, sTab, sInt, sA, vP, bc, vlid, M.fnrm + ((e0s + (factive ? ln.ef : 0)) * Nfq + ln.fn) * 3, ph, sDv, nullptr);
  }
  __syncthreads();
  double vt = 0.0;
  if (ln.vin) {
    double sgx[3], sgy[3], gx[3], gy[3];
    visc_sigma<N1>(ln, sTab, sInt, TT, ph, M.geo + (e0s + (vactive ? ln.ev : 0)) * GEO_STRIDE,
                   M.fnrm + (e0s + (vactive ? ln.ev : 0)) * Nfq * 3, sA, sDv, sgx, sgy,
                   DIAG ? gx : nullptr, DIAG ? gy : nullptr);
    double2* r = reinterpret_cast<double2*>(sB + (ln.ev * Nq + ln.q) * 6);
    r[0] = make_double2(sgx[0], sgx[1]);
    r[1] = make_double2(sgx[2], sgy[0]);
    r[2] = make_double2(sgy[1], sgy[2]);
    if (!DIVV && vactive) {   // sigma at the Gauss nodes, kept for the divergence in the last phase: SG[6][K][Nq], coalesced
      const int64_t n = e0 * Nq + ln.tid, KN = M.K * Nq;
      SG[n] = sgx[0]; SG[KN + n] = sgx[1]; SG[2 * KN + n] = sgx[2];
      SG[3 * KN + n] = sgy[0]; SG[4 * KN + n] = sgy[1]; SG[5 * KN + n] = sgy[2];
    }
    if (DIAG && vactive)
      vt = M.wJq[(e0 + ln.ev) * Nq + ln.q] * (gx[0] * sgx[0] + gx[1] * sgx[1] + gx[2] * sgx[2] + gy[0] * sgy[0] + gy[1] * sgy[1] + gy[2] * sgy[2]);
  }
  __syncthreads();
  if (DIVV && ln.vin) {   // volume part of div sigma (dg_div! :590-611 without the lift), from sigma in LDS
    const double* g = M.geo + (e0s + (vactive ? ln.ev : 0)) * GEO_STRIDE;
    double dv[3] = {0, 0, 0};
#pragma unroll 1   // rolled: unrolled, the hoisted LDS reads cost 30-60 VGPRs and two occupancy steps (+3 % per RHS)
    for (int d = 0; d < 2; ++d) {
      const int op = d == 0 ? TT.op0 : TT.op1;
      const double gx = g[op], gy = g[2 + op];
      const int pos = ln.pos(d), oth = ln.oth(d);
      // walk the tensor line with pointer steps (no index arithmetic in the loop)
      const double* wp = sTab + L.DG + (d * N1 + pos) * N1;
      const double2* r = reinterpret_cast<const double2*>(sB + (ln.ev * Nq + (d == 0 ? N1 * oth : oth)) * 6);
      const int step = d == 0 ? 3 : 3 * N1;   // in double2
#pragma unroll 1
      for (int j = 0; j < N1; ++j, r += step) {
        const double w = wp[j];
        const double2 s0 = r[0], s1 = r[1], s2 = r[2];
        const double wx = w * gx, wy = w * gy;
        dv[0] += wx * s0.x + wy * s1.y;
        dv[1] += wx * s0.y + wy * s2.x;
        dv[2] += wx * s1.x + wy * s2.y;
      }
    }
    if (vactive) {   // SG[3][K][Nq], coalesced
      const int64_t n = e0 * Nq + ln.tid, KN = M.K * Nq;
      SG[n] = dv[0]; SG[KN + n] = dv[1]; SG[2 * KN + n] = dv[2];
    }
  }
  if (factive) {
    const double* gn = M.fnrm + ((e0 + ln.ef) * Nfq + ln.fn) * 3;
    double sn[3], fx[3], fy[3];
    face_normal_stress<N1>(ln, sTab, sInt, sB, gn[0], gn[1], sn, fx, fy);
    double* bb = B + ((e0 + ln.ef) * Nfq + ln.fn) * B_NC;
    bb[0] = sn[0]; bb[1] = sn[1]; bb[2] = sn[2];
  }
  if (DIAG) {   // fixed-order tree reduction over the workgroup
    __syncthreads();
    double* red = sB_;
    red[threadIdx.x] = vt;
    __syncthreads();
    for (int w = TPB / 2; w > 0; w >>= 1) {
      if ((int)threadIdx.x < w) red[threadIdx.x] += red[threadIdx.x + w];
      __syncthreads();
    }
    if (threadIdx.x == 0) vt_partial[blockIdx.x] = red[0];
  }
}

// ---------------------------------------------------------------------------------------------
// last phase: interface + volume flux differencing (+ viscous divergence and penalty) -> rhs
// ---------------------------------------------------------------------------------------------
template <int N1, bool VISC>
struct RhsLds {
  using G = typename WgR<N1, VISC>::G;
  static constexpr int Nq = N1 * N1, Nfq = 4 * N1, Nh = Nq + Nfq, E = G::E;
  static constexpr int nQh = E * Nh * 6;                 // prims+logs of all hybrid nodes; also interp scratch
  static constexpr int nAcc1 = 4 * Nq + 4 * Nfq;         // accumulators of one element: sAcc[4][Nq] + sG[4][Nfq]
  static constexpr int nFlux = (E + G::NS) * nAcc1; // sAcc + sG, and a second copy for every straddling element
  static constexpr int nVisc = VISC ? E * 3 * Nfq : 0;   // sSj(3) per face node (sVn and sS live in the sQh region)
  static_assert(!VISC || 6 * Nq <= 6 * Nh, "sS must fit in the sQh region");
  static constexpr int nR2 = nFlux > nVisc ? nFlux : nVisc;
  static_assert(8 * Nq <= 6 * Nh, "interp scratch must fit in the sQh region");
};

template <int N1, bool MODAL, bool VISC, bool WALLS>
__global__ __launch_bounds__((WgR<N1, VISC>::TPB)) void kt_rhs(TensorTables TT, MeshDev M, Phys ph, const double* Q,
                                             const double* __restrict__ A_U, const double* __restrict__ SG,
                                             const double* __restrict__ B, double* rhs, LsrkFuse lf) {
  using LD = RhsLds<N1, VISC>;
  constexpr int Nq = LD::Nq, Nfq = LD::Nfq, Nh = LD::Nh, E = LD::E, NF = N1 / 2;
  constexpr int NFULL = (N1 - 1) / 2;   // full circulant rounds per direction (the antipodal round of even N1 is separate)
  constexpr int UNR_K = VISC ? (NFULL > 0 ? NFULL : 1) : 1, UNR_T = VISC ? 2 : 1, UNR_D = N1 <= 4 ? 2 : 1;
  constexpr TensorLayout L(N1);
  __shared__ double sTab[L.NDBL];
  __shared__ int sInt[L.NINT];
  using W = WgR<N1, VISC>;
  using G = typename W::G;
  __shared__ __align__(16) double sQh_[W::NG * LD::nQh];
  __shared__ __align__(16) double sR2_[W::NG * LD::nR2];
  const Lane<N1, W::GW> ln;
  double* sQh = sQh_ + ln.grp * LD::nQh;
  double* sR2 = sR2_ + ln.grp * LD::nR2;
  double* sAcc = sR2;                  // [E][4][Nq]   partner contributions (volume nodes)
  double* sG = sR2 + E * 4 * Nq;       // [E][4][Nfq]  face-node sums, then QF_f + wfac*flux_f
  // ds_add_f64 order is fixed within a wave but not between waves: an element whose lanes straddle two waves gets a
  // second accumulator copy (sAcc2 | sG2) that the lanes of its second wave add into; the copies are summed in a fixed
  // order afterwards, so the result does not depend on wave scheduling.
  double* myAcc = sAcc + ln.ev * 4 * Nq;     // where this volume lane adds partner contributions
  double* myG = sG + ln.ev * 4 * Nfq;
  const double* acc2 = nullptr;              // second copy of this volume lane's element (null: not straddling)
  double* zeroF2 = nullptr;                  // second sG copy this face lane initialises
  if (G::NS > 0) {
    double* sec = sR2 + E * LD::nAcc1;
    int so_v = 0, so_f = 0;
    bool sv = false, sf = false;
#pragma unroll
    for (int e = 0; e < E; ++e) {
      if (G::strad(e)) {
        if (e < ln.ev) ++so_v;
        if (e < ln.ef) ++so_f;
        if (e == ln.ev) sv = true;
        if (e == ln.ef) sf = true;
      }
    }
    if (sv && ln.vin) {
      double* c2 = sec + so_v * LD::nAcc1;
      acc2 = c2;
      if (ln.tid / TW != (ln.ev * Nq) / TW) { myAcc = c2; myG = c2 + 4 * Nq; }
    }
    if (sf && ln.fin) zeroF2 = sec + so_f * LD::nAcc1 + 4 * Nq;
  }
  stage_tables<N1>(TT, sTab, sInt);
  __syncthreads();
  const int64_t e0 = M.e_begin + ((int64_t)blockIdx.x * W::NG + ln.grp) * E;
  const int nE = (int)max((int64_t)0, min((int64_t)E, M.e_begin + M.e_count - e0));
  const bool vactive = ln.tid < nE * Nq, factive = ln.tid < nE * Nfq;
  const int64_t e0s = min(e0, M.e_begin + M.e_count - 1);   // in-range base for the geometry reads of idle lanes / idle groups
  const double* g = M.geo + (e0s + (vactive ? ln.ev : 0)) * GEO_STRIDE;

  // ---- every global load of this workgroup is issued before any arithmetic -------------------
  double x[4];
  issue_state_loads<N1>(Q, M.K, e0, vactive, ln.tid, x);
  double qM[8], qP[8], pnr[3] = {0, 0, 0}, bPn[3] = {0, 0, 0}, bOwn[3] = {0, 0, 0};   // pnr: penalty tau*[[v]] of the face node
  int64_t mpk = 0;
  int bcf = 0;
  double vlid = 1.0;
#pragma unroll
  for (int c = 0; c < 8; ++c) { qM[c] = 1.0; qP[c] = 1.0; }
  if (factive) {
    const int64_t n = (e0 + ln.ef) * Nfq + ln.fn;
    const int64_t mp = M.mapP[n];
    if (WALLS) bcf = M.bc[n];   // WALLS <=> M.bc != null (periodic meshes compile the wall branches away)
    if (WALLS && bcf == 2 && M.vlid) vlid = M.vlid[n];
    const double2* aM = reinterpret_cast<const double2*>(A_U + n * FAU_NC);
    const double2* aP = reinterpret_cast<const double2*>(A_U + mp * FAU_NC);
#pragma unroll
    for (int c = 0; c < 2; ++c) {
      const double2 m = aM[c], p = aP[c];
      qM[2 * c] = m.x; qM[2 * c + 1] = m.y;
      qP[2 * c] = p.x; qP[2 * c + 1] = p.y;
    }
    mpk = mp;
    // logs, energy and wavespeed of the two trace states (devmath::trace_rest), with the face means of the geometry record
    // as phase 0 used to: the wavespeed only scales the LF term, itself a small jump
    const double* gmn = M.geo + (e0 + ln.ef) * GEO_STRIDE + 5 + 3 * (ln.fn / N1);
    const double isJm = rcp_refined(gmn[2]);
    trace_rest(qM, gmn[0], gmn[1], isJm, Gas<MODAL>::GM1);
    trace_rest(qP, gmn[0], gmn[1], isJm, Gas<MODAL>::GM1);
  }

  // ---- state at the Gauss node -> primitives + logs in registers and LDS ---------------------
  double U[4];
  state_at_quad<N1, MODAL>(ln, sTab, sQh, sQh + E * 4 * Nq, x, U);
  double qh[6];
  prim_logs<MODAL>(U, qh);
  __syncthreads();   // interp scratch (aliases sQh) is dead
  if (ln.vin) {
    double2* dq = reinterpret_cast<double2*>(sQh + (ln.ev * Nh + ln.q) * 6);
    dq[0] = make_double2(qh[0], qh[1]);
    dq[1] = make_double2(qh[2], qh[3]);
    dq[2] = make_double2(qh[4], qh[5]);
#pragma unroll
    for (int c = 0; c < 4; ++c) sAcc[(ln.ev * 4 + c) * Nq + ln.q] = 0.0;
    if (G::NS > 0 && acc2) {
      double* z = const_cast<double*>(acc2);
#pragma unroll
      for (int c = 0; c < 4; ++c) z[c * Nq + ln.q] = 0.0;
    }
  }
  // ---- face lanes: interface flux (euler_quad.jl:158-169 / update_flux! :308-324), kept in registers
  const bool inviscid = (ph.parts & 1) != 0;
  if (ln.fin) {
    double2* dq = reinterpret_cast<double2*>(sQh + (ln.ef * Nh + Nq + ln.fn) * 6);
    dq[0] = make_double2(qM[0], qM[1]);
    dq[1] = make_double2(qM[2], qM[3]);
    dq[2] = make_double2(qM[4], qM[5]);
    const double* gn = M.fnrm + ((e0s + (factive ? ln.ef : 0)) * Nfq + ln.fn) * 3;
    if (VISC && ph.viscous_dissp) {
      // penalty tau*[[v]] (:817-837): own and neighbour projected entropy variables are the entropy variables of the
      // two trace states (see kt_project), so no interpolation of nodal values is needed here
      const double bM = 2 * Gas<MODAL>::GM1 * qM[3], bP = 2 * Gas<MODAL>::GM1 * qP[3];
      const double vf[3] = {bM * qM[1], bM * qM[2], -bM}, vPn[3] = {bP * qP[1], bP * qP[2], -bP};
      double dV[3];
      face_jump_and_penalty(vf, vPn, bcf, vlid, gn, ph, dV, pnr);
    }
    if (bcf >= 3) {   // shock-tube closures (dg2D_CNS_modalESDG.jl:168-185): Dirichlet state / copy, lam = lamP = 0
#pragma unroll
      for (int c = 0; c < 6; ++c) qP[c] = bcf == 3 ? ph.inflow_q[c] : qM[c];
      qM[6] = 0.0; qP[6] = 0.0;
    } else if (bcf) {   // wall: mirror state rho+ = rho, beta+ = beta, u+ = u - 2 (u.n) n  (impose_BCs_inviscid! :157-176)
      const double is = rcp_refined(gn[2]);
      const double nx = gn[0] * is, ny = gn[1] * is;
      const double un = qM[1] * nx + qM[2] * ny;
#pragma unroll
      for (int c = 0; c < 8; ++c) qP[c] = qM[c];
      qP[1] = qM[1] - 2 * un * nx;
      qP[2] = qM[2] - 2 * un * ny;
    }

    double Fn[4];
    ec_flux_dir<MODAL>(qM, qP, gn[0], gn[1], Fn);
    const double LFc = ph.inviscid_dissp ? ph.lf_scale * fmax(qM[6], qP[6]) * gn[2] : 0.0;
    // LF jump uses Uf[mapP] - Uf, which vanishes at walls (mapP = self), cavity :511-513
    double dU[4] = {qP[0] - qM[0], qP[0] * qP[1] - qM[0] * qM[1], qP[0] * qP[2] - qM[0] * qM[2], qP[7] - qM[7]};
    if (bcf) { dU[0] = 0.0; dU[1] = 0.0; dU[2] = 0.0; dU[3] = 0.0; }
    // the face accumulator starts from the lifted interface flux: G_f = wfac_f * flux_f + QF_f (QF_f is added by
    // the volume lanes below)
    const double wf = inviscid ? sTab[L.WFAC + ln.fn] : 0.0;
#pragma unroll
    for (int c = 0; c < 4; ++c) sG[(ln.ef * 4 + c) * Nfq + ln.fn] = wf * (Fn[c] - LFc * dU[c]);
    if (G::NS > 0 && zeroF2) {
#pragma unroll
      for (int c = 0; c < 4; ++c) zeroF2[c * Nfq + ln.fn] = 0.0;
    }
  }
  __syncthreads();

  // ---- flux differencing along the tensor lines (sparse_hadamard_sum :102-138 / flux_differencing! :326-348)
  double acc[4] = {0, 0, 0, 0};
  if (ln.vin && inviscid && !(ph.dbg & 1)) {
#pragma unroll UNR_D   // both directions inlined up to N1 = 4 (N=3: -1 %); from N1 = 5 it costs the viscous kernel an occupancy step (+5 %)
    for (int d = 0; d < 2; ++d) {
      const int op = d == 0 ? TT.op0 : TT.op1;
      const double gx = 2 * g[op], gy = 2 * g[2 + op];
      const int pos = ln.pos(d), oth = ln.oth(d), stride = d == 0 ? 1 : N1;
      const double wt = sTab[L.WT + d * N1 + oth];
#pragma unroll UNR_K   // unrolled only where it does not cost an occupancy step (A/B: CNS -0.7 %, Euler +3 %)
      for (int k = 0; k < NFULL; ++k) {   // full circulant rounds: pair (pos, pos+k+1 mod N1), every lane busy
        int pp = pos + k + 1;
        if (pp >= N1) pp -= N1;
        const int pid = ln.q + (pp - pos) * stride;
        const double cw = sTab[L.S + (d * N1 + pos) * N1 + pp] * wt;
        const double2* pr = reinterpret_cast<const double2*>(sQh + (ln.ev * Nh + pid) * 6);
        const double2 p0 = pr[0], p1 = pr[1], p2 = pr[2];
        const double qj[6] = {p0.x, p0.y, p1.x, p1.y, p2.x, p2.y};
        double Fd[4];
        ec_flux_dir<MODAL>(qh, qj, cw * gx, cw * gy, Fd);
        double* tgt = myAcc + pid;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          acc[c] += Fd[c];
          lds_add(tgt + c * Nq, -Fd[c]);
        }
      }
#pragma unroll UNR_T
      for (int t = 0; t < 2; ++t) {
        const int f = sInt[L.FN + (d * 2 + t) * N1 + oth];
        const double cw = sTab[L.SF + (d * 2 + t) * N1 + pos] * sTab[L.WTF + (d * 2 + t) * N1 + oth];
        const double2* pr = reinterpret_cast<const double2*>(sQh + (ln.ev * Nh + Nq + f) * 6);
        const double2 p0 = pr[0], p1 = pr[1], p2 = pr[2];
        const double qj[6] = {p0.x, p0.y, p1.x, p1.y, p2.x, p2.y};
        double vv[4];
        ec_flux_dir<MODAL>(qh, qj, cw * gx, cw * gy, vv);
        double* tgt = myG + f;
#pragma unroll
        for (int c = 0; c < 4; ++c) acc[c] += vv[c];
        // all N1 lanes of the line add into the same face node: rotate the field order by the lane's
        // position so that one ds_add_f64 instruction hits (almost) distinct addresses instead of N1 equal ones
        const int rot = pos & 3;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const double w = rot == 0 ? vv[i] : (rot == 1 ? vv[(i + 1) & 3] : (rot == 2 ? vv[(i + 2) & 3] : vv[(i + 3) & 3]));
          lds_add(tgt + ((i + rot) & 3) * Nfq, -w);
        }
      }
    }
    // even N1: the antipodal pairs (pos, pos + N1/2) exist once per two lanes of a line.  With b_d = (pos_d >= N1/2)
    // and S = {b_0 != b_1} every such pair has exactly one endpoint in S, so ONE round serves both directions:
    // S-lanes take their d = 0 pair, the other lanes their d = 1 pair.
    if (N1 % 2 == 0) {
      constexpr int H = N1 / 2;
      const bool inS = (ln.a >= H) != (ln.b >= H);
      const int d = inS ? 0 : 1;
      const int op = d == 0 ? TT.op0 : TT.op1;
      const double gx = 2 * g[op], gy = 2 * g[2 + op];
      const int pos = ln.pos(d), oth = ln.oth(d), stride = d == 0 ? 1 : N1;
      int pp = pos + H;
      if (pp >= N1) pp -= N1;
      const int pid = ln.q + (pp - pos) * stride;
      const double cw = sTab[L.S + (d * N1 + pos) * N1 + pp] * sTab[L.WT + d * N1 + oth];
      const double2* pr = reinterpret_cast<const double2*>(sQh + (ln.ev * Nh + pid) * 6);
      const double2 p0 = pr[0], p1 = pr[1], p2 = pr[2];
      const double qj[6] = {p0.x, p0.y, p1.x, p1.y, p2.x, p2.y};
      double Fd[4];
      ec_flux_dir<MODAL>(qh, qj, cw * gx, cw * gy, Fd);
      double* tgt = myAcc + pid;
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        acc[c] += Fd[c];
        lds_add(tgt + c * Nq, -Fd[c]);
      }
    }
  }
  __syncthreads();
  // ---- collocated rhs: -(Ph*QF + Lf*flux)/J  (euler_quad.jl:170-184 / cavity :514-518) -------
  const double iJ = rcp_refined(g[4]);
  double R[4];
  {
    const double pd = sTab[L.PD + ln.q];
    double r[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      double part = sAcc[(ln.ev * 4 + c) * Nq + ln.q];
      if (G::NS > 0 && acc2) part += acc2[c * Nq + ln.q];
      r[c] = pd * (acc[c] + part);
    }
#pragma unroll 1
    for (int d = 0; d < 2; ++d) {
      const int pos = ln.pos(d), oth = ln.oth(d);
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        const int f = sInt[L.FN + (d * 2 + t) * N1 + oth];
        const double w = sTab[L.PF + (d * 2 + t) * N1 + pos] * sTab[L.PTF + (d * 2 + t) * N1 + oth];
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          double gf = sG[(ln.ev * 4 + c) * Nfq + f];
          if (G::NS > 0 && acc2) gf += acc2[4 * Nq + c * Nfq + f];
          r[c] += w * gf;
        }
      }
    }
#pragma unroll
    for (int c = 0; c < 4; ++c) R[c] = -r[c] * iJ;
  }
  // ---- viscous terms (rhs_viscous! :749-849 in collocated form) ------------------------------------
  if (VISC && (ph.parts & 2) && !(ph.dbg & 2)) {
    __syncthreads();   // sAcc / sG are dead; sR2 becomes the viscous scratch
    if (factive) {   // neighbour traces of the viscous part
      const double* bp = B + mpk * B_NC;
#pragma unroll
      for (int c = 0; c < 3; ++c) bPn[c] = bp[c];
      if (!WALLS) {   // own normal stress: phase 1 already computed it (no walls: the x/y split is not needed)
        const double* bo = B + ((e0 + ln.ef) * Nfq + ln.fn) * B_NC;
#pragma unroll
        for (int c = 0; c < 3; ++c) bOwn[c] = bo[c];
      }
    }
    // sigma = K(v) grad v at the Gauss nodes was computed (and its face traces exchanged) by phase 1: reload it
    // instead of recomputing gradient and stress (HBM has headroom here, the LDS does not)
    double* sSj = sR2;                       // [E][3][Nfq]  stress jump (+ J * penalty, see below)
    double* sS = sQh;                        // [E][Nq][6]   in the dead primitive region (WALLS only)
    double dv[3] = {0, 0, 0};                // divergence of sigma at this node
    if (WALLS) {   // sigma itself: the wall closures need its face values split into x and y parts
      double2 sg0 = make_double2(0, 0), sg1 = sg0, sg2 = sg0;
      if (vactive) {
        const int64_t n = e0 * Nq + ln.tid, KN = M.K * Nq;
        sg0 = make_double2(SG[n], SG[KN + n]);
        sg1 = make_double2(SG[2 * KN + n], SG[3 * KN + n]);
        sg2 = make_double2(SG[4 * KN + n], SG[5 * KN + n]);
      }
      if (ln.vin) {
        double2* r = reinterpret_cast<double2*>(sS + (ln.ev * Nq + ln.q) * 6);
        r[0] = sg0; r[1] = sg1; r[2] = sg2;
      }
      __syncthreads();
    } else if (vactive) {   // phase 1 stored the volume part of the divergence (kt_sigma<.., DIVV>)
      const int64_t n = e0 * Nq + ln.tid, KN = M.K * Nq;
      dv[0] = SG[n]; dv[1] = SG[KN + n]; dv[2] = SG[2 * KN + n];
    }
    // stress jumps .5*((sxP-sxf)*nxJ + (syP-syf)*nyJ): the neighbour's normal stress from B carries
    // its own outward normal = minus ours (dg_div! :606)
    if (ln.fin) {
      const double* gn = M.fnrm + ((e0s + (factive ? ln.ef : 0)) * Nfq + ln.fn) * 3;
      double sn[3], fx[3] = {0, 0, 0}, fy[3] = {0, 0, 0}, sj[3];
      if (WALLS) {
        face_normal_stress<N1>(ln, sTab, sInt, sS, gn[0], gn[1], sn, fx, fy);
      } else {
#pragma unroll
        for (int c = 0; c < 3; ++c) sn[c] = bOwn[c];
      }
#pragma unroll
      for (int c = 0; c < 3; ++c) sj[c] = .5 * (-bPn[c] - sn[c]);
      if (bcf >= 3) {   // sigma+ = sigma- (dg2D_CNS_modalESDG.jl:205-216)
        sj[0] = 0.0; sj[1] = 0.0; sj[2] = 0.0;
      } else if (bcf) {   // impose_BCs_stress! :218-262
        if (ph.BCTYPE == 1) {
          sj[0] = 0.0; sj[1] = 0.0;
          sj[2] = bcf == 2 ? -sn[2] + vlid * sn[0] : -sn[2];
        } else if (ph.BCTYPE == 2) {
          sj[0] = 0.0; sj[1] = 0.0; sj[2] = 0.0;
        } else {
          const double is = rcp_refined(gn[2]);
          const double n1 = gn[0] * is, n2 = gn[1] * is;
          const double snx = fx[0] * n1 + fx[1] * n2, sny = fy[0] * n1 + fy[1] * n2;
          sj[0] = .5 * ((-2 * fx[0] + 2 * n1 * snx) * gn[0] + (-2 * fy[0] + 2 * n1 * sny) * gn[1]);
          sj[1] = .5 * ((-2 * fx[1] + 2 * n2 * snx) * gn[0] + (-2 * fy[1] + 2 * n2 * sny) * gn[1]);
          sj[2] = -sn[2];
        }
      }
      // the penalty is lifted WITHOUT the 1/J of the divergence (:839-845, quirk Q3): fold it into the stress jump
      // as J*pn so one lifted array serves both (J*(1/J) differs from 1 by one rounding of the penalty only)
      const double Jf = ph.viscous_dissp ? M.geo[(e0s + (factive ? ln.ef : 0)) * GEO_STRIDE + 4] : 0.0;
#pragma unroll
      for (int c = 0; c < 3; ++c) sSj[(ln.ef * 3 + c) * Nfq + ln.fn] = sj[c] + Jf * pnr[c];
    }
    __syncthreads();
    // divergence + penalty (dg_div! :590-611, penalty :817-845: NOT scaled by 1/J, quirk Q3)
    if (ln.vin) {
#pragma unroll 1
      for (int d = 0; d < 2; ++d) {
        const int op = d == 0 ? TT.op0 : TT.op1;
        const double gx = g[op], gy = g[2 + op];
        const int pos = ln.pos(d), oth = ln.oth(d);
#pragma unroll 1
        for (int j = 0; WALLS && j < N1; ++j) {
          const double w = sTab[L.DG + (d * N1 + pos) * N1 + j];
          const double2* r = reinterpret_cast<const double2*>(sS + (ln.ev * Nq + (d == 0 ? j + N1 * oth : oth + N1 * j)) * 6);
          const double2 s0 = r[0], s1 = r[1], s2 = r[2];
          const double wx = w * gx, wy = w * gy;
          dv[0] += wx * s0.x + wy * s1.y;
          dv[1] += wx * s0.y + wy * s2.x;
          dv[2] += wx * s1.x + wy * s2.y;
        }
#pragma unroll
        for (int t = 0; t < 2; ++t) {
          const int f = sInt[L.FN + (d * 2 + t) * N1 + oth];
          const double lw = sTab[L.PF + (d * 2 + t) * N1 + pos] * sTab[L.PTF + (d * 2 + t) * N1 + oth] * sTab[L.WFAC + f];
#pragma unroll
          for (int c = 0; c < 3; ++c) dv[c] += lw * sSj[(ln.ev * 3 + c) * Nfq + f];
        }
      }
#pragma unroll
      for (int c = 0; c < 3; ++c) R[c + 1] += dv[c] * iJ;
    }
  }
  __syncthreads();   // sQh is dead: it becomes the Pq scratch
  store_rhs_from_quad<N1, MODAL>(ln, sTab, rhs, lf, M.K, e0, vactive, sQh, sQh + E * 4 * Nq, R);
}

__global__ void kt_log_test(const double* __restrict__ x, double* __restrict__ y, int64_t n) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) y[i] = tdev::log_pos(x[i]);
}

int launch_log_test(const double* x, double* y, int64_t n, hipStream_t s) {
  hipLaunchKernelGGL(kt_log_test, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, x, y, n);
  return (int)hipGetLastError();
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

int launch_project_tensor(int N1v, const TensorTables& TT, const MeshDev& M, const Phys& ph, const double* Q,
                          double* A_U, double* A_v, hipStream_t s) {
  if (M.e_count <= 0) return 0;
  const bool modal = ph.formulation != 0, visc = ph.formulation == 1;
  ESDG_DISPATCH_N1(N1v, {
    using W = WgP<N1>;
    constexpr int EPB = W::EPB;
    constexpr int TPB = W::TPB;
    const int nb = (int)((M.e_count + EPB - 1) / EPB);
    if (!modal)
      hipLaunchKernelGGL((kt_project<N1, false, false>), dim3(nb), dim3(TPB), 0, s, TT, M, ph, Q, A_U, A_v);
    else if (visc)
      hipLaunchKernelGGL((kt_project<N1, true, true>), dim3(nb), dim3(TPB), 0, s, TT, M, ph, Q, A_U, A_v);
    else
      hipLaunchKernelGGL((kt_project<N1, true, false>), dim3(nb), dim3(TPB), 0, s, TT, M, ph, Q, A_U, A_v);
  });
  return (int)hipGetLastError();
}

int sigma_tensor_blocks(int N1v, int64_t K) {
  ESDG_DISPATCH_N1(N1v, {
    constexpr int EPB = WgS<N1>::EPB;
    return (int)((K + EPB - 1) / EPB);
  });
  return 0;
}

int launch_sigma_tensor(int N1v, const TensorTables& TT, const MeshDev& M, const Phys& ph, const double* Q,
                        const double* A_U, double* B, double* SG, double* vt_partial, hipStream_t s) {
  if (M.e_count <= 0) return 0;
  ESDG_DISPATCH_N1(N1v, {
    using W = WgS<N1>;
    constexpr int EPB = W::EPB;
    constexpr int TPB = W::TPB;
    const int nb = (int)((M.e_count + EPB - 1) / EPB);
    const bool divv = M.bc == nullptr;   // must match WALLS of launch_rhs_tensor: kt_rhs<.., WALLS = false> reads the divergence form
    if (vt_partial && divv)
      hipLaunchKernelGGL((kt_sigma<N1, true, true>), dim3(nb), dim3(TPB), 0, s, TT, M, ph, Q, A_U, B, SG, vt_partial);
    else if (vt_partial)
      hipLaunchKernelGGL((kt_sigma<N1, true, false>), dim3(nb), dim3(TPB), 0, s, TT, M, ph, Q, A_U, B, SG, vt_partial);
    else if (divv)
      hipLaunchKernelGGL((kt_sigma<N1, false, true>), dim3(nb), dim3(TPB), 0, s, TT, M, ph, Q, A_U, B, SG, vt_partial);
    else
      hipLaunchKernelGGL((kt_sigma<N1, false, false>), dim3(nb), dim3(TPB), 0, s, TT, M, ph, Q, A_U, B, SG, vt_partial);
  });
  return (int)hipGetLastError();
}

int launch_rhs_tensor(int N1v, const TensorTables& TT, const MeshDev& M, const Phys& ph, const double* Q,
                      const double* A_U, const double* SG, const double* B, double* rhs, const LsrkFuse& lf,
                      hipStream_t s) {
  if (M.e_count <= 0) return 0;
  const bool modal = ph.formulation != 0, visc = ph.formulation == 1;
  // grid and workgroup shape depend on the instantiation (WgR<N1, VISC>)
#define ESDG_LAUNCH_RHS(MODALv, VISCv, WALLSv)                                                                   \
  do {                                                                                                          \
    using W = WgR<N1, VISCv>;                                                                                   \
    const int nb = (int)((M.e_count + W::EPB - 1) / W::EPB);                                                    \
    hipLaunchKernelGGL((kt_rhs<N1, MODALv, VISCv, WALLSv>), dim3(nb), dim3(W::TPB), 0, s, TT, M, ph, Q, A_U, SG, \
                       B, rhs, lf);                                                                             \
  } while (0)
  ESDG_DISPATCH_N1(N1v, {
    const bool walls = M.bc != nullptr;
    if (!modal) ESDG_LAUNCH_RHS(false, false, false);
    else if (visc && walls) ESDG_LAUNCH_RHS(true, true, true);
    else if (visc) ESDG_LAUNCH_RHS(true, true, false);
    else if (walls) ESDG_LAUNCH_RHS(true, false, true);
    else ESDG_LAUNCH_RHS(true, false, false);
  });
#undef ESDG_LAUNCH_RHS
  return (int)hipGetLastError();
}

}  // namespace esdg
