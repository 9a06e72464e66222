// esdg_kernels_tensor3.hip -- third generation of the last-phase kernel of the 2D tensor path for gfx950 (MI355X / CDNA4).
//
// Same formulas, tables and face-trace protocol as kt2_rhs (esdg_kernels_tensor2.hip; reference: euler_quad.jl:141-194 /
// rhs_inviscid! cavity_optimized.jl:447-528, update_flux! :308-324, flux_differencing! :326-348, dg_div! :590-611), a different
// mapping of the flux differencing.  kt2_rhs gives every Gauss node a lane and returns the partner's share of each flux through
// LDS (ds_add_f64 into two-add cells): its LDS is ~80 % busy, a quarter of that in bank conflicts of the 8-byte atomics on the
// 16-byte pair planes, two waves per group meet at eight barriers, and the VALU sits at 73 % (profiles/r03z_sq_counters.txt).
// Here
//   * a workgroup is ONE wave and owns E = 64 / (2 N1) elements: no inter-wave barrier exists, every wave runs through its
//     load -> Vq -> primitives -> fluxes -> Pq -> store sequence on its own, and the waves of a SIMD are in different stages;
//   * the flux differencing is LINE PER LANE: lane (element, direction d, line o) owns the N1 Gauss nodes of one tensor line and
//     the two face nodes at its ends, evaluates all C(N1,2) volume-volume pairs, the 2 N1 volume-face pairs and the two interface
//     fluxes of its line, and keeps the N1 + 2 accumulators (4 components each) in registers: both shares of a flux are added
//     in the lane that computed it -- no LDS atomics, no accumulator planes, no face-total planes; the partner records are
//     streamed from LDS (3 ds_read_b128 per flux, the only LDS traffic of the stage);
//   * the SBP weight of a pair is folded into the accumulation (acc_i += S_ij F, acc_j -= S_ij F: the flux is linear in the metric
//     vector, which carries the line's transverse weight), and the lift of the two face totals is applied by the line lane, so a
//     node's result is the sum of what its two lines hold for it: r = r_0 + r_1, one LDS exchange (direction 0 writes, direction 1 adds);
//   * node-wise work (Vq, primitives + logs, gather, viscous divergence, Pq, store) runs in ceil(E N1^2 / 64) rounds of the same
//     wave with the node-per-lane layout of kt2_rhs.
// An element's result depends on nothing but its own data and its neighbours' traces (no cross-lane reduction whose order
// depends on the slot), so ranged, sharded and full launches agree bit for bit as before.
// Meshes with walls: the WALLS instantiation applies the closures of init_BC_funs / the shock-tube driver in the face turns and
// the nodal-basis correction of the wall elements after Pq, both as kt2_rhs does (formulas and citations there); CNS wall meshes
// from N1 = 6 on run kt2_rhs (launch_rhs3 below).
// Round 5: the instruction stream was attributed at ISA level (tools/isa_buckets.py) and dieted -- a smooth
// wave calls the series flux directly, the flux works on sums with exact power-of-two folds, its series constants are pinned in
// VGPRs (flux_dir / ec_flux_core) --, and the LDS block shrank from 13.2 to 10.1 KB at N1 = 5 (Vq / Pq in place, one pair of exchange
// planes, compact tables).
#include "esdg_dev.hpp"
#include "esdg_tensor_tables.hpp"
#include "esdg_devmath.hpp"
#include "esdg_t2_physics.hpp"

namespace esdg {
namespace t3 {

using namespace devmath;
using t2::d2;
using t2::ec_flux_dir;
using t2::Gas2;
using t2::prim_logs;

template <int N1> struct G3 {
  static constexpr int TW = 64, Nq = N1 * N1, Nfq = 4 * N1, NLN = 2 * N1;
  static constexpr int E = TW / NLN;                  // elements of a wave
  static constexpr int NV = E * Nq, NF = E * Nfq, LL = E * NLN;
  static constexpr int NR = (NV + TW - 1) / TW;       // node rounds
  static_assert(E >= 1, "a wave holds an element");
};

// The directional flux.  `series` (wave-uniform): the wave is known to be smooth (every pair all-series, see the kernel's smooth-wave
// test) -- the series variant directly, without ec_flux_dir's per-flux series tests and ballots (which in such a wave would pick
// the same variant flux by flux: same bits).
template <bool MODAL>
__device__ __forceinline__ void flux_dir(bool series, const double* qL, const double* qR, double gx, double gy, double* F, const t2::SeriesK& sk) {
  if (series) {
    t2::ec_flux_core<MODAL, 1>(qL, qR, gx, gy, F, qR[0] - qL[0], qR[0] + qL[0], qR[3] - qL[3], qR[3] + qL[3], true, true, sk);
  } else {
    ec_flux_dir<MODAL>(qL, qR, gx, gy, F, sk);
  }
}

// Compiler-only memory fence in front of every record read of the line stage: the records are read-only there, so without it
// hipcc merges the repeated reads of a node's record (the node is a partner in N1 + 1 pairs) and keeps all N1 records of the
// line -- 12 VGPRs each -- live across the whole stage on top of the accumulators: 284 registers at N1 = 5, spills under any cap.
#define T3_FENCE() do { __builtin_amdgcn_sched_barrier(0); asm volatile("" ::: "memory"); } while (0)

// ... and a pin behind every flux: the updated accumulators pass through an empty asm statement, which hipcc cannot move relative
// to the fences.  Without it the second half of every flux (averages, the four components, the accumulation) is sunk towards
// the stage's end, where the accumulators are first read, and ten doubles of intermediates per flux stay live until there.
#define T3_PIN4(a) asm volatile("" : "+v"((a)[0]), "+v"((a)[1]), "+v"((a)[2]), "+v"((a)[3]))

// Workgroup -> element group, XCD-contiguous (A/B hook, see xcd_group in esdg_kernels_tensor2.hip): the dispatcher deals
// consecutive workgroups to the 8 XCDs in turn; with the remap the workgroups of XCD x = blockIdx % 8 walk the x-th contiguous
// eighth of the groups, so that the rows above and below a group -- whose traces it reads -- belong to the same L2.
#ifndef ESDG_T3_XCD_REMAP
#define ESDG_T3_XCD_REMAP 0
#endif
__device__ __forceinline__ int64_t xcd_group3(int64_t b, int64_t n) {
  if (!ESDG_T3_XCD_REMAP || n < 64) return b;
  const int64_t x = b % 8, q = n / 8, r = n % 8;
  return x * q + (x < r ? x : r) + b / 8;
}

#ifndef ESDG_T3_PIN_OUT
#define ESDG_T3_PIN_OUT 1   // (A/B hook: 0 = only the DOPRI45 instantiation pins its results ahead of the epilogue)
#endif
#ifndef ESDG_T3_LAZY_LOGS
#define ESDG_T3_LAZY_LOGS 1   // (A/B hook: 0 = the logarithms of every node and trace state, always)
#endif
// waves per SIMD asked of the register allocator: three up to N1 = 6 (168 VGPRs: the N1 + 2 accumulators of a line take 8 (N1 + 2)
// of them), two at N1 = 7, 8 (180-196 VGPRs with Pq's results pinned, ESDG_T3_PIN_OUT) and for the CNS wall instantiation from
// N1 = 5 (its correction planes and penalty shares; one at N1 = 8, which is not launched).  ESDG_T3_WPE overrides (A/B hook)
#ifdef ESDG_T3_WPE
constexpr int wpe3(int, bool) { return ESDG_T3_WPE; }
#else
constexpr int wpe3(int N1, bool walls_cns) { return N1 <= 4 ? 3 : (N1 <= 6 ? (walls_cns ? 2 : 3) : (N1 == 7 || !walls_cns ? 2 : 1)); }
#endif
template <int N1, bool MODAL, bool VISC, bool WALLS, bool STG = false>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(wpe3(N1, WALLS && VISC && MODAL)))) void kt3_rhs(TensorTables TT, MeshDev M, Phys ph, const double* Q,
                                                          const double* __restrict__ A_U, const double* __restrict__ SG,
                                                          const double* __restrict__ B, double* rhs, LsrkFuse lf, StageFuse sf) {
  // (Q and rhs are NOT restrict-qualified: the fused RK forms write the state in place -- lf.Qw, StageFuse::y may be the array Q
  // points to -- and the DOPRI error stage re-reads rhs.  The in-place update is safe because a wave reads its elements' state at
  // entry only; the compiler fence in front of the epilogue keeps every Q load above every store.)
  using G = G3<N1>;
  constexpr int TW = G::TW, Nq = G::Nq, Nfq = G::Nfq, NLN = G::NLN, E = G::E, NV = G::NV, LL = G::LL, NR = G::NR;
  constexpr TensorLayout TL(N1);
  constexpr int NGEO = E * GEO_STRIDE, GPT = (NGEO + TW - 1) / TW;
  constexpr double GM1 = Gas2<MODAL>::GM1;
  // LDS (round 5: 10.1 instead of 13.2 KB at N1 = 5 -- sixteen workgroups per CU fit): ONE block of
  //   arena  6 NV doubles: in turn Vq's buffer ([2][NV] pair planes, both stages in place: a wave reads everything a stage needs
  //          into registers before it writes), the node records ([3][NV] pair planes: (rho,u) (v,beta) (log rho, log beta)), the
  //          lines' results per node ([2][NV]: direction 0 writes, direction 1 adds) and Pq's buffer (in place as well);
  //   sTab   the 1D operator tables this kernel reads (TensorLayout without its EE and DG blocks: NTAB doubles);
  //   sGeo   the geometry records of the wave's elements, laid over the padding of sTab's last staging round (staged after it).
  constexpr int cIQ = TL.EE, cIP = TL.EE + N1 * N1, NTAB = TL.EE + 2 * N1 * N1;   // compact offsets of IQ, IP; table doubles staged
  static_assert(TL.DG == TL.EE + 4 * N1 && TL.IQ == TL.DG + 2 * N1 * N1 && TL.IP == TL.IQ + N1 * N1, "TensorLayout order: ..., EE, DG, IQ, IP");
  constexpr int TPTc = (NTAB + TW - 1) / TW;
  constexpr int OFF_TAB = 6 * NV, OFF_GEO = OFF_TAB + NTAB, NLDS = OFF_GEO + GPT * TW;
  static_assert(OFF_TAB + TPTc * TW <= NLDS, "the table staging stays inside the block");
  __shared__ __align__(16) double lds[NLDS];
  double* arena = lds;
  double* sTab = lds + OFF_TAB;
  double* sGeo = lds + OFF_GEO;
  // WALLS (CNS): the lines' shares of the lifted penalty per node, [2 directions][3][NV], and the elements that have a boundary node
  constexpr bool WCORR = WALLS && VISC && MODAL;
  __shared__ double sX[WCORR ? 6 * NV : 1];
  __shared__ int sEb[WALLS ? E : 1];
  d2* sA = reinterpret_cast<d2*>(arena);                // [2][NV]
  d2* sRec = reinterpret_cast<d2*>(arena);              // [3][NV]
  d2* sS = reinterpret_cast<d2*>(arena);                // [2][NV]

  const unsigned tid = threadIdx.x;
  const int64_t e0r = M.e_begin + xcd_group3(blockIdx.x, gridDim.x) * E;
  const int nE = (int)min((int64_t)E, M.e_begin + M.e_count - e0r);
  const int64_t e0 = ESDG_EW(e0r);
  const int64_t KN = M.K * Nq;
#ifdef ESDG_T3_ATTR   // ISA-attribution builds only (tools/isa_buckets.py): the uniform switches as constants -> straight-line code
  const bool inviscid = true, viscous = VISC;
  lf.Qw = nullptr;
#else
  const bool inviscid = (ph.parts & 1) != 0, viscous = VISC && (ph.parts & 2) != 0;
#endif

  // ---- this lane's line: element el, direction d, transverse index o; nodes n0 + i st; face nodes fA (end 0), fB (end 1) ----
  const unsigned ln = tid < (unsigned)LL ? tid : tid - LL;   // (lanes beyond the lines duplicate a line: same LDS writes)
  const unsigned el = ln / NLN, lr = ln - el * NLN, d = lr / N1, o = lr - d * N1;
  const unsigned elc = el < (unsigned)nE ? el : 0u;          // (elements beyond the range: the data of the first one)
  const unsigned n0 = el * Nq + (d ? o : N1 * o), st = d ? N1 : 1;
  const int fA = TT.ints[TL.FN + (2 * d) * N1 + o], fB = TT.ints[TL.FN + (2 * d + 1) * N1 + o];
  const int64_t nfA = (e0 + elc) * Nfq + fA, nfB = (e0 + elc) * Nfq + fB;
  const unsigned mpA = ESDG_EWN((unsigned)M.mapP[nfA], Nfq), mpB = ESDG_EWN((unsigned)M.mapP[nfB], Nfq);
  const int64_t nsA = trace_slot<N1>(M, e0 + elc, (unsigned)fA), nsB = trace_slot<N1>(M, e0 + elc, (unsigned)fB);   // own records (MeshDev::bf)

  // ---- every global load of the inputs, unconditionally ------------------------------------------------------------------
  double x[NR][4], geo[GPT], tab[TPTc];
#pragma unroll
  for (int r = 0; r < NR; ++r) {
    const unsigned n = tid + r * TW, s = n < (unsigned)NV ? n : n - NV, sl = s < (unsigned)(nE * Nq) ? s : 0u;
#pragma unroll
    for (int f = 0; f < 4; ++f) x[r][f] = Q[f * KN + e0 * Nq + sl];
  }
#pragma unroll
  for (int i = 0; i < GPT; ++i) { const unsigned n = tid + i * TW; geo[i] = M.geo[e0 * GEO_STRIDE + (n < (unsigned)(nE * GEO_STRIDE) ? n : 0u)]; }
#pragma unroll
  for (int i = 0; i < TPTc; ++i) {   // (entries [0, EE) and [IQ, NDBL) of the tables: EE and DG are not read here)
    const unsigned n = tid + i * TW;
    tab[i] = TT.dbl[n < (unsigned)TL.EE ? n : (n < (unsigned)NTAB ? n + (unsigned)(TL.IQ - TL.EE) : 0u)];
  }
  // traces of the line's two face nodes, both sides: (rho, u, v, beta); their logs, energy and wavespeed are rebuilt below
  double qMA[8], qPA[8], qMB[8], qPB[8];
  {
    const d2* aMA = reinterpret_cast<const d2*>(A_U + nsA * FAU_NC);
    const d2* aMB = reinterpret_cast<const d2*>(A_U + nsB * FAU_NC);
    const d2* aPA = reinterpret_cast<const d2*>(A_U + (size_t)mpA * FAU_NC);
    const d2* aPB = reinterpret_cast<const d2*>(A_U + (size_t)mpB * FAU_NC);
#pragma unroll
    for (int c = 0; c < 2; ++c) {
      const d2 ma = aMA[c], mb = aMB[c], pa = aPA[c], pb = aPB[c];
      qMA[2 * c] = ma.x; qMA[2 * c + 1] = ma.y; qMB[2 * c] = mb.x; qMB[2 * c + 1] = mb.y;
      qPA[2 * c] = pa.x; qPA[2 * c + 1] = pa.y; qPB[2 * c] = pb.x; qPB[2 * c + 1] = pb.y;
    }
  }
  const float2 ndA = reinterpret_cast<const float2*>(M.fnd)[nfA], ndB = reinterpret_cast<const float2*>(M.fnd)[nfB];
  int bcA = 0, bcB = 0;              // WALLS: boundary flag (1 wall, 2 lid, 3 inflow, 4 copy), lid velocity, sJ - face mean
  double vlA = 1.0, vlB = 1.0;
  float sdA = 0.f, sdB = 0.f;
  if (WALLS) {
    bcA = M.bc[nfA]; bcB = M.bc[nfB];
    if (M.vlid) { vlA = M.vlid[nfA]; vlB = M.vlid[nfB]; }
    sdA = M.fsd[nfA]; sdB = M.fsd[nfB];
    if (tid < (unsigned)E) sEb[tid] = 0;
  }
  double bsA[3] = {0, 0, 0}, bsB[3] = {0, 0, 0};   // central stress jump .5*((sxP-sxf)*nxJ + (syP-syf)*nyJ) of the two face nodes
  if (VISC) {
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      bsA[c] = .5 * (-B[(size_t)mpA * B_NC + c] - B[nsA * B_NC + c]);
      bsB[c] = .5 * (-B[(size_t)mpB * B_NC + c] - B[nsB * B_NC + c]);
    }
  }

  // ---- staging: geometry, tables, nodal values ------------------------------------------------------------------------------
#pragma unroll
  for (int i = 0; i < TPTc; ++i) sTab[tid + i * TW] = tab[i];
#pragma unroll
  for (int i = 0; i < GPT; ++i) sGeo[tid + i * TW] = geo[i];   // (after the tables: it lies over their padding)
  unsigned slot[NR], nq[NR];
#pragma unroll
  for (int r = 0; r < NR; ++r) {
    const unsigned n = tid + r * TW;
    slot[r] = n < (unsigned)NV ? n : n - NV;       // (slots beyond the nodes duplicate one: same values, same LDS writes)
    nq[r] = slot[r] % Nq;
  }
  double U[NR][4];
  if (MODAL) {
#pragma unroll
    for (int r = 0; r < NR; ++r) {
      sA[slot[r]] = make_double2(x[r][0], x[r][1]);
      sA[NV + slot[r]] = make_double2(x[r][2], x[r][3]);
    }
    __syncthreads();
    // Uq = Vq Qn by sum factorisation, exactly as t2::vq_apply: W[a + N1 b] = sum_i IQ[a,i] Qn[i + N1 b], then
    // Uq[a + N1 b] = sum_j IQ[a,j] W[b + N1 j]; W goes back into the buffer it was read from, once every round has read
    double W[NR][4];
#pragma unroll
    for (int r = 0; r < NR; ++r) {
      const unsigned ev = slot[r] / Nq, a = nq[r] % N1, b = nq[r] / N1;
      const double* c = sTab + cIQ + a * N1;
      const d2* rw = sA + ev * Nq + N1 * b;
      d2 p = rw[0], t = rw[NV];
      const double c0 = c[0];
      double w0 = c0 * p.x, w1 = c0 * p.y, w2 = c0 * t.x, w3 = c0 * t.y;
#pragma unroll
      for (int i = 1; i < N1; ++i) {
        p = rw[i]; t = rw[NV + i];
        const double ci = c[i];
        w0 = __builtin_fma(ci, p.x, w0); w1 = __builtin_fma(ci, p.y, w1);
        w2 = __builtin_fma(ci, t.x, w2); w3 = __builtin_fma(ci, t.y, w3);
      }
      W[r][0] = w0; W[r][1] = w1; W[r][2] = w2; W[r][3] = w3;
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < NR; ++r) {
      sA[slot[r]] = make_double2(W[r][0], W[r][1]);
      sA[NV + slot[r]] = make_double2(W[r][2], W[r][3]);
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < NR; ++r) {
      const unsigned ev = slot[r] / Nq, a = nq[r] % N1, b = nq[r] / N1;
      const double* c = sTab + cIQ + a * N1;
      const d2* rw = sA + ev * Nq + b;
      d2 p = rw[0], t = rw[NV];
      const double c0 = c[0];
      U[r][0] = c0 * p.x; U[r][1] = c0 * p.y; U[r][2] = c0 * t.x; U[r][3] = c0 * t.y;
#pragma unroll
      for (int j = 1; j < N1; ++j) {
        p = rw[N1 * j]; t = rw[NV + N1 * j];
        const double cj = c[j];
        U[r][0] = __builtin_fma(cj, p.x, U[r][0]); U[r][1] = __builtin_fma(cj, p.y, U[r][1]);
        U[r][2] = __builtin_fma(cj, t.x, U[r][2]); U[r][3] = __builtin_fma(cj, t.y, U[r][3]);
      }
    }
    __syncthreads();   // the records below overwrite both buffers
  } else {
#pragma unroll
    for (int r = 0; r < NR; ++r)
#pragma unroll
      for (int f = 0; f < 4; ++f) U[r][f] = x[r][f];
  }
  // Primitives; the logarithms only where a flux of this wave can read them.  The log-mean of a pair takes the reference's series
  // branch (logmean.jl:23-27, no logarithms) when |a_j - a_i| < 1e-4 (a_i + a_j)/2.  If every density and every beta this wave
  // holds -- its nodes' and the four trace states of every lane -- lies within 0.49e-4 (relative) of lane 0's first node, every pair
  // the wave evaluates satisfies that with 2 % to spare (|a_i - a_j| <= 0.98e-4 a_0 < 1e-4 (1 - 0.49e-4) a_0 <= 1e-4 (a_i + a_j)/2;
  // the rounding of the test itself is ~1e-16): ec_flux_dir then runs its all-series variant for every flux, which reads no
  // logarithm, so none is computed -- 14 logs per lane, a sixth of the kernel's instructions in smooth regions.  The
  // results are bit for bit those of the general path (the variants evaluate identical expressions, esdg_t2_physics.hpp).
  bool ok = true;
  double a0 = 0.0, b0 = 0.0;
#pragma unroll
  for (int r = 0; r < NR; ++r) {
    double qh[4];
    t2::prims<MODAL>(U[r], qh);
    sRec[slot[r]] = make_double2(qh[0], qh[1]);
    sRec[NV + slot[r]] = make_double2(qh[2], qh[3]);
    if (r == 0) { a0 = first_lane(qh[0]); b0 = first_lane(qh[3]); }
    ok = ok && fabs(qh[0] - a0) <= 0.49e-4 * a0 && fabs(qh[3] - b0) <= 0.49e-4 * b0;
    U[r][0] = qh[0]; U[r][3] = qh[3];   // (kept for the logs below)
  }
  {
    const double ta = 0.49e-4 * a0, tb = 0.49e-4 * b0;
    ok = ok && fabs(qMA[0] - a0) <= ta && fabs(qPA[0] - a0) <= ta && fabs(qMB[0] - a0) <= ta && fabs(qPB[0] - a0) <= ta;
    ok = ok && fabs(qMA[3] - b0) <= tb && fabs(qPA[3] - b0) <= tb && fabs(qMB[3] - b0) <= tb && fabs(qPB[3] - b0) <= tb;
  }
  if (WALLS) ok = ok && bcA != 3 && bcB != 3;   // (a Dirichlet inflow state takes the place of a neighbour's and is not part of this test)
#ifdef ESDG_T3_ATTR
  const bool smooth = ESDG_T3_ATTR == 1 && (ok || !ok);
#else
  // (through readfirstlane: a scalar the compiler KNOWS to be uniform -- as a lane mask it came back through a v_cndmask / v_cmp pair in
  // front of every flux)
  const bool smooth = __builtin_amdgcn_readfirstlane(
      (int)(ESDG_T3_LAZY_LOGS && !(ph.dbg & 32) && __builtin_amdgcn_ballot_w64(ok) == __builtin_amdgcn_ballot_w64(true))) != 0;
#endif
  if (!smooth) {
#pragma unroll
    for (int r = 0; r < NR; ++r) sRec[2 * NV + slot[r]] = make_double2(log_pos(U[r][0]), log_pos(U[r][3]));
  }
  __syncthreads();

  // ---- line stage ----------------------------------------------------------------------------------------------------------------
  double acc[N1][4], GfA[4], GfB[4], gpA[3] = {0, 0, 0}, gpB[3] = {0, 0, 0};   // (gp: WALLS, the penalty's share of the face totals)
#pragma unroll
  for (int i = 0; i < N1; ++i)
#pragma unroll
    for (int c = 0; c < 4; ++c) acc[i][c] = 0.0;
  const double* g = sGeo + el * GEO_STRIDE;
  {
    const t2::SeriesK sk = t2::series_k_pinned();   // (the two series constants that must sit in VGPRs, held for the stage)
    const int opd = d ? TT.op1 : TT.op0;
    const double gxd = 2 * g[opd], gyd = 2 * g[2 + opd];     // metric vector of the line's direction (affine: one per element)
    // one face turn: interface flux + penalty + stress jump of face node f (end t of the line), then its N1 volume-face pairs
    auto face_turn = [&](int t, int f, double* qM, double* qP, const float2 nd, const double* bs, double* Gf, int bc, double vlid, float sd,
                         double* gpen) {
      const double* gm = g + 5 + 3 * (f / N1);             // face means of the record; + this node's difference = its own normal
      // (sJ: the face mean unless a wall closure turns it into a unit normal -- it only scales the LF term, a small jump)
      const double gn[3] = {gm[0] + (double)nd.x, gm[1] + (double)nd.y, WALLS ? gm[2] + (double)sd : gm[2]};
      if (WALLS && bc) sEb[el] = 1;
      {
        const double isJm = rcp_refined(gm[2]);
        if (!smooth) { qM[4] = log_pos(qM[0]); qM[5] = log_pos(qM[3]); qP[4] = log_pos(qP[0]); qP[5] = log_pos(qP[3]); }
        else { qM[4] = 0.0; qM[5] = 0.0; qP[4] = 0.0; qP[5] = 0.0; }   // (never read: every flux of a smooth wave is all-series)
        trace_rest_nolog(qM, gm[0], gm[1], isJm, GM1);
        trace_rest_nolog(qP, gm[0], gm[1], isJm, GM1);
      }
      double pnr[3] = {0, 0, 0};
      if (VISC) {   // penalty tau*[[v]] (:817-837): the projected entropy variables are those OF the trace states
        const double bM = 2 * GM1 * qM[3], bP = 2 * GM1 * qP[3];
        const double tau = ph.viscous_dissp ? -rcp_refined(-bM) * ph.inv_Re : 0.0;
        if (WALLS && bc) {   // exterior values by the wall closure; third component overridden as in :827-837 (see kt2_rhs)
          const double vf[3] = {bM * qM[1], bM * qM[2], -bM};
          double vP[3];
          t2::wall_exterior_v(vf, bc, vlid, gn, ph, vP);
          const double dV[3] = {vP[0] - vf[0], vP[1] - vf[1], vP[2] - vf[2]};
          const double a2 = .5 * (vP[0] + vf[0]), a3 = .5 * (vP[1] + vf[1]);
          double sq = a2 * dV[0] + a3 * dV[1];
          if (ph.BCTYPE != 1) sq += dV[2] * dV[2] * .5;
          pnr[0] = tau * dV[0];
          pnr[1] = tau * dV[1];
          pnr[2] = -tau * sq * rcp_refined(vf[2]);
        } else {
          pnr[0] = tau * (bP * qP[1] - bM * qM[1]);
          pnr[1] = tau * (bP * qP[2] - bM * qM[2]);
          pnr[2] = tau * (bM - bP);
        }
      }
      if (WALLS && bc >= 3) {   // shock-tube closures (dg2D_CNS_modalESDG.jl:168-185): Dirichlet state / copy, lam = lamP = 0
#pragma unroll
        for (int c = 0; c < 6; ++c) qP[c] = bc == 3 ? ph.inflow_q[c] : qM[c];
        qM[6] = 0.0; qP[6] = 0.0;
      } else if (WALLS && bc) {   // wall: mirror state rho+ = rho, beta+ = beta, u+ = u - 2 (u.n) n  (impose_BCs_inviscid! :157-176)
        const double is = rcp_refined(gn[2]);
        const double nx = gn[0] * is, ny = gn[1] * is;
        const double un = qM[1] * nx + qM[2] * ny;
#pragma unroll
        for (int c = 0; c < 8; ++c) qP[c] = qM[c];
        qP[1] = qM[1] - 2 * un * nx;
        qP[2] = qM[2] - 2 * un * ny;
      }
      double Fn[4];
      flux_dir<MODAL>(smooth, qM, qP, gn[0], gn[1], Fn, sk);
      const double LFc = ph.inviscid_dissp ? ph.lf_scale * fmax(qM[6], qP[6]) * gn[2] : 0.0;
      // (the LF jump uses Uf[mapP] - Uf, which vanishes at boundary nodes: mapP = self, cavity :511-513)
      const double dz = (WALLS && bc) ? 0.0 : 1.0;
      const double dU[4] = {dz * (qP[0] - qM[0]), dz * (qP[0] * qP[1] - qM[0] * qM[1]), dz * (qP[0] * qP[2] - qM[0] * qM[2]), dz * (qP[7] - qM[7])};
      const double wfac = sTab[TL.WFAC + f];
      const double wf = inviscid ? wfac : 0.0;
#pragma unroll
      for (int c = 0; c < 4; ++c) Gf[c] = wf * (Fn[c] - LFc * dU[c]);
      if (VISC) {   // stress jump + J * penalty (lifted WITHOUT 1/J, quirk Q3) ride in the face total with the opposite sign
        const double Jf = g[4], ws = viscous ? wfac : 0.0;
#pragma unroll
        for (int c = 0; c < 3; ++c) Gf[c + 1] = __builtin_fma(-ws, __builtin_fma(Jf, pnr[c], bs[c]), Gf[c + 1]);
        if (WALLS) {
#pragma unroll
          for (int c = 0; c < 3; ++c) gpen[c] = ws * Jf * pnr[c];
        }
      }
      if (inviscid) {   // (uniform)
        const double wt = sTab[TL.WTF + (2 * d + t) * N1 + o];
        const double gxf = gxd * wt, gyf = gyd * wt;
        const double* SFk = sTab + TL.SF + (2 * d + t) * N1;
#pragma unroll
        for (int j = 0; j < N1; ++j) {
          const unsigned n = n0 + j * st;
          T3_FENCE();
          const d2 p0 = sRec[n], p1 = sRec[NV + n], p2 = sRec[2 * NV + n];
          const double qj[6] = {p0.x, p0.y, p1.x, p1.y, p2.x, p2.y};
          double F[4];
          flux_dir<MODAL>(smooth, qj, qM, gxf, gyf, F, sk);
          const double c = SFk[j];
#pragma unroll
          for (int k = 0; k < 4; ++k) { acc[j][k] = __builtin_fma(c, F[k], acc[j][k]); Gf[k] = __builtin_fma(-c, F[k], Gf[k]); }
          T3_PIN4(acc[j]); T3_PIN4(Gf);
        }
      }
    };
    face_turn(0, fA, qMA, qPA, ndA, bsA, GfA, bcA, vlA, sdA, gpA);
    face_turn(1, fB, qMB, qPB, ndB, bsB, GfB, bcB, vlB, sdB, gpB);
    if (inviscid) {   // volume-volume pairs of the line, each once
      const double wt = sTab[TL.WT + d * N1 + o];
      const double gxv = gxd * wt, gyv = gyd * wt;
      const double* Sd = sTab + TL.S + d * N1 * N1;
#pragma unroll
      for (int i = 0; i < N1 - 1; ++i) {
        const unsigned ni = n0 + i * st;
        T3_FENCE();
        const d2 a0 = sRec[ni], a1 = sRec[NV + ni], a2 = sRec[2 * NV + ni];
        const double qi[6] = {a0.x, a0.y, a1.x, a1.y, a2.x, a2.y};
#pragma unroll
        for (int j = i + 1; j < N1; ++j) {
          const unsigned n = n0 + j * st;
          T3_FENCE();
          const d2 p0 = sRec[n], p1 = sRec[NV + n], p2 = sRec[2 * NV + n];
          const double qj[6] = {p0.x, p0.y, p1.x, p1.y, p2.x, p2.y};
          double F[4];
          flux_dir<MODAL>(smooth, qi, qj, gxv, gyv, F, sk);
          const double c = Sd[i * N1 + j];
#pragma unroll
          for (int k = 0; k < 4; ++k) { acc[i][k] = __builtin_fma(c, F[k], acc[i][k]); acc[j][k] = __builtin_fma(-c, F[k], acc[j][k]); }
          T3_PIN4(acc[i]); T3_PIN4(acc[j]);
        }
      }
    }
  }
  // The node-layout ids pass through an empty asm statement: what the compiler derived from them before the line stage (LDS byte
  // addresses of the Vq / Pq rounds, element and row ids -- some twenty VGPRs it kept alive across the stage, one of them spilled once
  // the series constants were pinned) is derived again behind it from these six registers.
#pragma unroll
  for (int r = 0; r < NR; ++r) asm volatile("" : "+v"(slot[r]), "+v"(nq[r]));
  // viscous volume divergence of the wave's nodes (phase 1): requested here, consumed after the exchange below
  double dvs[NR][3];
  if (VISC) {
#pragma unroll
    for (int r = 0; r < NR; ++r) {
      const unsigned sl = slot[r] < (unsigned)(nE * Nq) ? slot[r] : 0u;
#pragma unroll
#ifdef ESDG_EXP_NOSG   // bound experiment (wrong results): the stored divergence read from an L2-resident window = what removing its 600 B / element buys at most
      for (int c = 0; c < 3; ++c) dvs[r][c] = SG[c * KN + (e0 & 1023) * Nq + sl];
#else
      for (int c = 0; c < 3; ++c) dvs[r][c] = SG[c * KN + e0 * Nq + sl];
#endif
    }
  }
  __syncthreads();   // every lane is past its reads of the records, whose space takes the lines' results
  {   // collocated projection and lift along the line: r_d[node i] = PD[node i] acc_i + PW_A[i] Gf_A + PW_B[i] Gf_B
    const double ptA = sTab[TL.PTF + (2 * d) * N1 + o], ptB = sTab[TL.PTF + (2 * d + 1) * N1 + o];
    const double* PFA = sTab + TL.PF + (2 * d) * N1;
    const double* PFB = sTab + TL.PF + (2 * d + 1) * N1;
    const unsigned q0 = n0 - el * Nq;
#pragma unroll
    for (int i = 0; i < N1; ++i) {
      const double pd = sTab[TL.PD + q0 + i * st], pa = PFA[i] * ptA, pb = PFB[i] * ptB;
      double rr[4];
#pragma unroll
      for (int c = 0; c < 4; ++c) rr[c] = __builtin_fma(pb, GfB[c], __builtin_fma(pa, GfA[c], pd * acc[i][c]));
      const unsigned n = n0 + i * st;
#pragma unroll
      for (int c = 0; c < 4; ++c) acc[i][c] = rr[c];   // (the line's share of node i, kept for the two passes below)
      if (WCORR) {
#pragma unroll
        for (int c = 0; c < 3; ++c) sX[(3 * d + c) * NV + n] = __builtin_fma(pb, gpB[c], pa * gpA[c]);
      }
    }
    // a node's result = the shares of its two lines: the direction-0 lines write, then the direction-1 lines add (one pair of planes
    // instead of two; lanes that duplicate a line -- tid >= LL -- write duplicates but never add)
    if (d == 0) {
#pragma unroll
      for (int i = 0; i < N1; ++i) {
        const unsigned n = n0 + i * st;
        sS[n] = make_double2(acc[i][0], acc[i][1]);
        sS[NV + n] = make_double2(acc[i][2], acc[i][3]);
      }
    }
    __syncthreads();
    if (d == 1 && tid < (unsigned)LL) {
#pragma unroll
      for (int i = 0; i < N1; ++i) {
        const unsigned n = n0 + i * st;
        const d2 a0 = sS[n], a1 = sS[NV + n];
        sS[n] = make_double2(a0.x + acc[i][0], a0.y + acc[i][1]);
        sS[NV + n] = make_double2(a1.x + acc[i][2], a1.y + acc[i][3]);
      }
    }
  }
  __syncthreads();
  // Meshes with walls: the reference divides the nodal coefficients of everything but the penalty by J[i,e] NODE BY NODE; in the
  // elements with a boundary node that shows and the result is corrected after Pq exactly as in kt2_rhs:
  // out_i (1 + g_i) - g_i (Pq X)_i, g_i = J / J[i,e] - 1, X = lift of the penalty + volume divergence at the Gauss nodes.
  const bool gb = WCORR && M.wgeo && __builtin_amdgcn_ballot_w64((bcA | bcB) != 0) != 0;   // (uniform: the workgroup is this wave)

  // ---- node rounds: rhs at the Gauss nodes = -(r_0 + r_1)/J (+ viscous volume divergence / J), then Pq -----------------------
  double R[NR][4], RX[WCORR ? NR : 1][4];
#pragma unroll
  for (int r = 0; r < NR; ++r) {
    const unsigned s = slot[r], ev = s / Nq;
    const double iJ = rcp_refined(sGeo[ev * GEO_STRIDE + 4]);
    const d2 a0 = sS[s], a1 = sS[NV + s];
    R[r][0] = -a0.x * iJ; R[r][1] = -a0.y * iJ; R[r][2] = -a1.x * iJ; R[r][3] = -a1.y * iJ;
    if (VISC) {
      const double vs = viscous ? iJ : 0.0;
#pragma unroll
      for (int c = 0; c < 3; ++c) R[r][c + 1] = __builtin_fma(dvs[r][c], vs, R[r][c + 1]);
      if (WCORR && gb) {
        RX[r][0] = 0.0;
#pragma unroll
        for (int c = 0; c < 3; ++c) RX[r][c + 1] = __builtin_fma(dvs[r][c], vs, (sX[c * NV + s] + sX[(3 + c) * NV + s]) * iJ);
      }
    }
  }
  double out[NR][4];
  if (MODAL) {
    // o = Pq X as in kt2_rhs: W[a + N1 b] = sum_j IP[a,j] X[b + N1 j], o[a + N1 b] = sum_i IP[b,i] W[a + N1 i]
    auto pq_apply = [&](const double (*X)[4], double (*o)[4]) {
      __syncthreads();   // every lane is past its reads of what the buffer held before
#pragma unroll
      for (int r = 0; r < NR; ++r) {
        sA[slot[r]] = make_double2(X[r][0], X[r][1]);
        sA[NV + slot[r]] = make_double2(X[r][2], X[r][3]);
      }
      __syncthreads();
      double W[NR][4];   // (first stage into registers, then back into the same buffer: in place, as Vq above)
#pragma unroll
      for (int r = 0; r < NR; ++r) {
        const unsigned ev = slot[r] / Nq, a = nq[r] % N1, b = nq[r] / N1;
        const double* c = sTab + cIP + a * N1;
        const d2* rw = sA + ev * Nq + b;
        d2 p = rw[0], t = rw[NV];
        const double c0 = c[0];
        double w0 = c0 * p.x, w1 = c0 * p.y, w2 = c0 * t.x, w3 = c0 * t.y;
#pragma unroll
        for (int j = 1; j < N1; ++j) {
          p = rw[N1 * j]; t = rw[NV + N1 * j];
          const double cj = c[j];
          w0 = __builtin_fma(cj, p.x, w0); w1 = __builtin_fma(cj, p.y, w1);
          w2 = __builtin_fma(cj, t.x, w2); w3 = __builtin_fma(cj, t.y, w3);
        }
        W[r][0] = w0; W[r][1] = w1; W[r][2] = w2; W[r][3] = w3;
      }
      __syncthreads();
#pragma unroll
      for (int r = 0; r < NR; ++r) {
        sA[slot[r]] = make_double2(W[r][0], W[r][1]);
        sA[NV + slot[r]] = make_double2(W[r][2], W[r][3]);
      }
      __syncthreads();
#pragma unroll
      for (int r = 0; r < NR; ++r) {
        const unsigned ev = slot[r] / Nq, a = nq[r] % N1, b = nq[r] / N1;
        const double* c = sTab + cIP + b * N1;
        const d2* rw = sA + ev * Nq + a;
        d2 p = rw[0], t = rw[NV];
        const double c0 = c[0];
        o[r][0] = c0 * p.x; o[r][1] = c0 * p.y; o[r][2] = c0 * t.x; o[r][3] = c0 * t.y;
#pragma unroll
        for (int i = 1; i < N1; ++i) {
          p = rw[N1 * i]; t = rw[NV + N1 * i];
          const double ci = c[i];
          o[r][0] = __builtin_fma(ci, p.x, o[r][0]); o[r][1] = __builtin_fma(ci, p.y, o[r][1]);
          o[r][2] = __builtin_fma(ci, t.x, o[r][2]); o[r][3] = __builtin_fma(ci, t.y, o[r][3]);
        }
      }
    };
    pq_apply(R, out);
    if (WCORR && gb) {   // (uniform) second product for the part that keeps the record's J; correction per element and node
      double ox[NR][4];
      pq_apply(RX, ox);
#pragma unroll
      for (int r = 0; r < NR; ++r) {
        const unsigned ev = slot[r] / Nq, evc = ev < (unsigned)nE ? ev : 0u;
        if (sEb[ev]) {
          const double Jn = M.wgeo[((e0 + evc) * 5 + 4) * Nq + nq[r]];
          const double gam = __builtin_fma(sGeo[ev * GEO_STRIDE + 4], rcp_refined(Jn), -1.0);
#pragma unroll
          for (int f = 0; f < 4; ++f) out[r][f] = __builtin_fma(gam, out[r][f] - ox[r][f], out[r][f]);
        }
      }
    }
  } else {
#pragma unroll
    for (int r = 0; r < NR; ++r)
#pragma unroll
      for (int f = 0; f < 4; ++f) out[r][f] = R[r][f];
  }
  asm volatile("" ::: "memory");   // (no load of the state sinks below this point: the epilogue may overwrite it in place)
  if (STG || ESDG_T3_PIN_OUT) {   // (the results in registers before any of the epilogue's loads is issued)
#pragma unroll
    for (int r = 0; r < NR; ++r) T3_PIN4(out[r]);
  }
  // ---- store, or fused low-storage RK stage (dg2D_euler_quad.jl:204-205) -------------------------------------------------------
#pragma unroll
  for (int r = 0; r < NR; ++r) {
    const unsigned n = tid + r * TW;
    if (STG) {   // DOPRI45: the store of k_s plus the next stage's state / the error norm (StageFuse, esdg_dev.hpp)
      T3_FENCE();   // (one round's loads at a time: hoisted together they spill)
      if (n < (unsigned)(nE * Nq)) {
        const int64_t i0 = e0 * Nq + n;
        if (sf.y) {   // (uniform)
          double xo[4], kk[6][4];
#pragma unroll
          for (int f = 0; f < 4; ++f) xo[f] = sf.x0[f * KN + i0];
#pragma unroll
          for (int j = 0; j < 6; ++j)
            if (j < sf.ns) {   // (uniform)
#pragma unroll
              for (int f = 0; f < 4; ++f) kk[j][f] = sf.k[j][f * KN + i0];
            }
#pragma unroll
          for (int f = 0; f < 4; ++f) rhs[f * KN + i0] = out[r][f];
#pragma unroll
          for (int f = 0; f < 4; ++f) {
            double a = 0.0, ee = 0.0;
#pragma unroll
            for (int j = 0; j < 6; ++j)
              if (j < sf.ns) { a = __builtin_fma(sf.c[j], kk[j][f], a); ee = __builtin_fma(sf.ce[j], kk[j][f], ee); }
            a = __builtin_fma(sf.c_last, out[r][f], a);
            sf.y[f * KN + i0] = __builtin_fma(sf.dt, a, xo[f]);
            if (sf.e_out) sf.e_out[f * KN + i0] = __builtin_fma(sf.ce_last, out[r][f], ee);
          }
        } else {
          double xo[4], ei[4];
#pragma unroll
          for (int f = 0; f < 4; ++f) { xo[f] = sf.x0[f * KN + i0]; ei[f] = sf.err ? rhs[f * KN + i0] : 0.0; }
          double t = 0.0;                 // the norm's term of this node (k_dopri_err's chain) at the node's own index: the host adds
#pragma unroll
          for (int f = 0; f < 4; ++f) {   // the terms in ONE order (k_chunk_sum), whatever launches the phase was cut into
            rhs[f * KN + i0] = out[r][f];
            const double e = __builtin_fma(sf.ce_last, out[r][f], ei[f]);
            const double sc = fabs(e) / (sf.tol * (1 + fabs(xo[f])));
            t = __builtin_fma(sc, sc, t);
          }
          if (sf.err) sf.partial[i0] = t;   // (uniform)
        }
      }
      continue;
    }
    if (n < (unsigned)(nE * Nq)) {
      if (lf.Qw) {   // (uniform)
        double ro[4], qo[4];
#pragma unroll
        for (int f = 0; f < 4; ++f) { const int64_t idx = f * KN + e0 * Nq + n; ro[f] = lf.res[idx]; qo[f] = lf.Qw[idx]; }
#pragma unroll
        for (int f = 0; f < 4; ++f) {
          const int64_t idx = f * KN + e0 * Nq + n;
          const double rr = __builtin_fma(lf.a, ro[f], lf.dt * out[r][f]);
          lf.res[idx] = rr;
          lf.Qw[idx] = __builtin_fma(lf.b, rr, qo[f]);
        }
      } else {
#pragma unroll
        for (int f = 0; f < 4; ++f) rhs[f * KN + e0 * Nq + n] = out[r][f];
      }
    }
  }
}

}  // namespace t3

#ifndef ESDG_T3_NO_DISPATCH   // (tools/isa_buckets.py includes this file for one explicit instantiation)
#if ESDG_MAX_N1 >= 12
#define ESDG_T3_DISPATCH_HI(...) case 9: { constexpr int N1 = 9; __VA_ARGS__; } break; case 10: { constexpr int N1 = 10; __VA_ARGS__; } break; \
  case 11: { constexpr int N1 = 11; __VA_ARGS__; } break; case 12: { constexpr int N1 = 12; __VA_ARGS__; } break;
#elif ESDG_MAX_N1 >= 10
#define ESDG_T3_DISPATCH_HI(...) case 9: { constexpr int N1 = 9; __VA_ARGS__; } break; case 10: { constexpr int N1 = 10; __VA_ARGS__; } break;
#else
#define ESDG_T3_DISPATCH_HI(...)
#endif
#define ESDG_T3_DISPATCH(N1v, BODY)                  \
  switch (N1v) {                                     \
    case 2: { constexpr int N1 = 2; BODY; } break;   \
    case 3: { constexpr int N1 = 3; BODY; } break;   \
    case 4: { constexpr int N1 = 4; BODY; } break;   \
    case 5: { constexpr int N1 = 5; BODY; } break;   \
    case 6: { constexpr int N1 = 6; BODY; } break;   \
    case 7: { constexpr int N1 = 7; BODY; } break;   \
    case 8: { constexpr int N1 = 8; BODY; } break;   \
    ESDG_T3_DISPATCH_HI(BODY)                          \
    default: return -1;                              \
  }

// CNS on meshes with walls from N1 = 6 on: kt2_rhs (measured in round 5 on the lid-driven cavity, 128 x 128: N = 5 0.1021 vs 0.1047 ms,
// N = 6 0.1652 vs 0.1910 ms -- the wall instantiation of kt3_rhs spills 54-64 registers there).  Not instantiated -- except at N1 = 10,
// which kt2_rhs does not reach (its packed rows hold eight partner ids): there the wall instantiation of this kernel runs, one wave
// per SIMD (512 registers, 22 spilled in the CNS form), so that every degree the library serves is served with walls too.
template <int N1, bool MODAL, bool VISC> constexpr bool kt3_serves_walls() { return (N1 < 8 && !(MODAL && VISC && N1 >= 6)) || N1 >= 10; }

template <int N1, bool MODAL, bool VISC>
static int launch_rhs3(const TensorTables& TT, const MeshDev& M, const Phys& ph, const double* Q, const double* A_U, const double* SG,
                       const double* B, double* rhs, const LsrkFuse& lf, hipStream_t s, const StageFuse* sf) {
  using G = t3::G3<N1>;
  const int nb = (int)((M.e_count + G::E - 1) / G::E);
  const StageFuse sf0{};
  if (M.bc) {
    if constexpr (kt3_serves_walls<N1, MODAL, VISC>()) {
      if (sf) {
        hipLaunchKernelGGL((t3::kt3_rhs<N1, MODAL, VISC, true, true>), dim3(nb), dim3(G::TW), 0, s, TT, M, ph, Q, A_U, SG, B, rhs, lf, *sf);
      } else {
        hipLaunchKernelGGL((t3::kt3_rhs<N1, MODAL, VISC, true>), dim3(nb), dim3(G::TW), 0, s, TT, M, ph, Q, A_U, SG, B, rhs, lf, sf0);
      }
      return 0;
    } else {
      return -1;
    }
  }
  if (sf) {   // DOPRI45 stage (esdg_dopri45_attempt); the reference integrates CNS so (cavity_optimized.jl:999-1037), the library every 2D formulation
    hipLaunchKernelGGL((t3::kt3_rhs<N1, MODAL, VISC, false, true>), dim3(nb), dim3(G::TW), 0, s, TT, M, ph, Q, A_U, SG, B, rhs, lf, *sf);
    return 0;
  }
  hipLaunchKernelGGL((t3::kt3_rhs<N1, MODAL, VISC, false>), dim3(nb), dim3(G::TW), 0, s, TT, M, ph, Q, A_U, SG, B, rhs, lf, sf0);
  return 0;
}

// last phase on meshes without walls; returns -1 where the v3 kernel does not apply (caller falls back to kt2_rhs)
int launch_rhs_tensor3(int N1v, const TensorTables& TT, const MeshDev& M, const Phys& ph, const double* Q, const double* A_U,
                       const double* SG, const double* B, double* rhs, const LsrkFuse& lf, hipStream_t s, const StageFuse* sf) {
  if (M.e_count <= 0) return 0;
  const bool modal = ph.formulation != 0, visc = ph.formulation == 1;
  int rc = 0;
  ESDG_T3_DISPATCH(N1v, {
    if (!modal) rc = (launch_rhs3<N1, false, false>)(TT, M, ph, Q, A_U, SG, B, rhs, lf, s, sf);
    else if (visc) rc = (launch_rhs3<N1, true, true>)(TT, M, ph, Q, A_U, SG, B, rhs, lf, s, sf);
    else rc = (launch_rhs3<N1, true, false>)(TT, M, ph, Q, A_U, SG, B, rhs, lf, s, sf);
  });
  if (rc) return rc;   // (-1: a wall mesh this kernel does not serve -- the caller takes kt2_rhs)
  return (int)hipGetLastError();
}

int rhs_tensor3_blocks(int N1v, int64_t e_count) {
  ESDG_T3_DISPATCH(N1v, { return (int)((e_count + t3::G3<N1>::E - 1) / t3::G3<N1>::E); });
}
#endif  // ESDG_T3_NO_DISPATCH

}  // namespace esdg
