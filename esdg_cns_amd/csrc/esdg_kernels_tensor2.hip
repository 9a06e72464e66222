// esdg_kernels_tensor2.hip -- tensor-product kernels of the 2D path for gfx950 (MI355X / CDNA4): phase 0 (kt2_project), phase 1
// (kt2_sigma) and the node-per-lane last phase (kt2_rhs: wall meshes from N = 7 on, A/B partner of kt3_rhs).  They run whenever
// the driver's operators factor into 1D tables (esdg_tensor_tables.hpp; always the case for init_reference_quad with a Gauss
// rule -- verified entry by entry in esdg_api.hip, otherwise the generic pair-list kernels of esdg_kernels.hip run).
//
// Reference: `rhs` / sparse_hadamard_sum of examples/dg2D_euler_quad.jl:102-194; rhs_inviscid! :447-528, update_flux! :308-324,
// flux_differencing! :326-348, rhs_viscous! :749-849, dg_grad! :548-569, dg_div! :590-611, viscous_matrices! :613-645 and the wall
// closures init_BC_funs :135-265 of examples/CompressibleNS/dg2D_CNS_cavity_optimized.jl; pointwise physics
// examples/EntropyStableEuler/{logmean,euler_fluxes,euler_variables}.jl (cited at each device function, esdg_t2_physics.hpp).
//
// Algorithm and face-trace protocol (shared with the generic kernels and kt3_rhs): phase 0 projects the entropy variables to the
// faces and writes one (rho, u, v, beta) record per face node (A_U); phase 1 (CNS) forms sigma = K(v) grad v with the BR1
// gradient, writes the normal-stress traces B and the volume part of div sigma (SG); the last phase evaluates every unordered
// flux pair of the 2 N1 tensor lines once (200 EC fluxes per element at N = 4, the reference visits 825 / 400), the interface
// fluxes with the LF penalty, lifts, the viscous divergence and Pq.  Mapping: a group of GW waves owns E elements, lane t of the
// group is Gauss node t % Nq of element t / Nq and face node t % Nfq of element t / Nfq (elements may straddle the waves of
// their group: all cross-lane traffic goes through LDS between workgroup barriers); Vq / Pq by sum factorisation.  One refined
// v_rcp_f64 serves the three quotients of an EC flux; pointwise work stays in (rho,u,v,beta,log rho,log beta).
//
// What the kernels of this file do about everything around the arithmetic (the round-1 kernels -- in the history up to round 4,
// esdg_kernels_tensor.hip -- were bound by instruction issue with only half of their VALU instructions doing fp64 math, the rest
// index arithmetic on the 1D tables, 64-bit address formation, runtime-direction selects and the EXEC bookkeeping of many small
// divergent regions):
//   * every per-node quantity (operator rows, lift / projection / SBP weights, partner and face ids) comes from ONE row of
//     host-built per-node tables (NodeLayout / FaceLayout, esdg_tensor_tables.hpp), loaded straight into registers;
//   * both tensor directions are compile-time: LDS addresses are a per-lane base plus an immediate offset;
//   * all LDS arrays are structure-of-arrays planes [component][E * nodes], so that consecutive lanes touch consecutive
//     8-byte slots (conflict-free ds_read_b64 / ds_write_b64);
//   * the per-element geometry records of a group are staged in LDS once;
//   * lanes without a node compute on duplicated data instead of branching; only stores are masked.
#include "esdg_dev.hpp"
#include "esdg_tensor_tables.hpp"
#include "esdg_devmath.hpp"
#include "esdg_t2_physics.hpp"

namespace esdg {
namespace t2 {

using namespace devmath;

constexpr int TW = 64;

// waves per group by degree (measured with the round-1 kernels, kept)
template <int N1> struct Cfg { static constexpr int GW = 1; };
template <> struct Cfg<5> { static constexpr int GW = 2; };
template <> struct Cfg<6> { static constexpr int GW = 4; };
template <> struct Cfg<7> { static constexpr int GW = 2; };   // 2 elements per group: the accumulator planes stay under 64 KB
template <> struct Cfg<8> { static constexpr int GW = 2; };
// N1 = 9, 10 (round 4, late; only these two choices were measured): three waves hold two elements at N1 = 9 (phases 0 / 1 at 256 x 256:
// 0.094 / 0.244 ms against 0.104 / 0.269 with two waves and one element); at N1 = 10 four waves would hold two elements but push
// kt2_sigma past 256 registers, i.e. to one workgroup per CU -- two waves, one element (78 % of the lanes)
#ifndef ESDG_T2_GW9
#define ESDG_T2_GW9 3
#endif
#ifndef ESDG_T2_GW10
#define ESDG_T2_GW10 2
#endif
template <> struct Cfg<9> { static constexpr int GW = ESDG_T2_GW9; };
template <> struct Cfg<10> { static constexpr int GW = ESDG_T2_GW10; };
template <> struct Cfg<11> { static constexpr int GW = 2; };   // (round 5: one element per group, 121 of 128 / 144 of 192 lanes)
template <> struct Cfg<12> { static constexpr int GW = 3; };

// ... and of the last-phase kernel where its measured optimum differs (N=5, 384x384, same box: kt2_rhs 0.400 ms with 4
// waves per group -- 58 KB LDS, 2 workgroups per CU -- 0.358 ms with 2; kt2_sigma the other way round, 0.161 vs 0.179 ms)
template <int N1> struct CfgRhs { static constexpr int GW = Cfg<N1>::GW; };
template <> struct CfgRhs<6> { static constexpr int GW = 2; };

template <int N1, int GWv = Cfg<N1>::GW> struct Geo {
  static constexpr int GW = GWv, GT = TW * GW, Nq = N1 * N1, Nfq = 4 * N1;
  static constexpr int E = (GT / Nq) < (GT / Nfq) ? (GT / Nq) : (GT / Nfq);
  static constexpr int NV = E * Nq, NF = E * Nfq;   // volume / face lanes of a group
  static_assert(E >= 1, "group does not hold an element");
};

template <int N1> using GeoR = Geo<N1, CfgRhs<N1>::GW>;   // group geometry of kt2_rhs
// ... and of kt2_sigma (A/B hook ESDG_T2_SIGMA_GW5: waves per group at N1 = 5)
#ifndef ESDG_T2_PROJECT_GRID_NUM
#define ESDG_T2_PROJECT_GRID_NUM 1   // persistent grid of kt2_project = resident workgroups x NUM / DEN (A/B hooks)
#endif
#ifndef ESDG_T2_PROJECT_GRID_DEN
#define ESDG_T2_PROJECT_GRID_DEN 1
#endif
#ifndef ESDG_T2_PROJECT_PERSIST
#define ESDG_T2_PROJECT_PERSIST 4   // (measured: 12.8 groups per resident workgroup at cfg3 -15 %, 3.2 at cfg2 +5 %) persistent kt2_project when the launch has more than that many groups per resident workgroup (0 = always one-shot)
#endif
#ifndef ESDG_T2_SIGMA_GW5
#define ESDG_T2_SIGMA_GW5 2
#endif
template <int N1> struct CfgSigmaG { static constexpr int GW = Cfg<N1>::GW; };
template <> struct CfgSigmaG<5> { static constexpr int GW = ESDG_T2_SIGMA_GW5; };
template <int N1> using GeoS = Geo<N1, CfgSigmaG<N1>::GW>;

// LDS layout.  Measured on MI355X (tools/ubench/lds_read.hip, 2 waves per SIMD): a ds_read_b64 and a ds_read_b128
// wave-instruction cost the same LDS time (~4.5 cycles; 119 vs 223 B/clk/CU), ds_read2_b64 costs two.  All arrays are
// therefore planes of double2 ("pair planes": two components of one node side by side, [pairs][E * nodes], lane stride
// 16 B = conflict-free ds_read_b128 / ds_write_b128) plus a plane of doubles for an odd component.

// Workgroup -> element group, XCD-contiguous (ESDG_T2_XCD_REMAP, A/B hook): the dispatcher deals consecutive workgroups to the 8
// XCDs in turn, so with group = blockIdx the five elements next door (and the row above, 102 groups away at cfg3) are always
// another XCD's and their traces another L2's.  With the remap the workgroups of XCD x = blockIdx % 8 walk the x-th contiguous
// eighth of the groups (a bijection for every grid size: XCD x holds q + (x < r) workgroups, q = n / 8, r = n % 8).
// Measured in round 3 (profiles/experiments/r03_xcd_remap_ab.log): cfg3 kt2_rhs -1.5 %, kt2_sigma +3 %, RHS +-0; N = 6: kt2_sigma
// +38 %; N = 2: RHS -3.6 %, N = 3: +1.5 % -- no consistent gain, off.
#ifndef ESDG_T2_XCD_REMAP
#define ESDG_T2_XCD_REMAP 0
#endif
constexpr int T2_NXCD = 8;
__device__ __forceinline__ int64_t xcd_group(int64_t b, int64_t n) {
  if (!ESDG_T2_XCD_REMAP || n < 8 * T2_NXCD) return b;
  const int64_t x = b % T2_NXCD, q = n / T2_NXCD, r = n % T2_NXCD;
  return x * q + (x < r ? x : r) + b / T2_NXCD;
}

// Uq = Vq Qn by sum factorisation through the pair planes sA, sB ([2][NV] each).  Nodal values are r-fastest, Gauss
// nodes s-fastest (SetupDG.jl:244): Vq[(a + N1 b), (i + N1 j)] = IQ[b,i] IQ[a,j].  c = IQ[a][:]; rowb = first slot of
// row b, colq = slot b of row 0 (both of this lane's element).
template <int N1, int NV>
__device__ __forceinline__ void vq_apply(const double* c, d2* sA, d2* sB0, d2* sB1, unsigned tv, unsigned rowb, unsigned colq,
                                         const double* x, double* U) {
  if (x) {
    sA[tv] = make_double2(x[0], x[1]);
    sA[NV + tv] = make_double2(x[2], x[3]);
  }
  __syncthreads();
  // stage 1: W[a + N1 b] = sum_i IQ[a,i] Qn[i + N1 b]   (this lane: row b of the nodal values)
  {
    const d2* r = sA + rowb;
    d2 p = r[0], t = r[NV];
    double w0 = c[0] * p.x, w1 = c[0] * p.y, w2 = c[0] * t.x, w3 = c[0] * t.y;
#pragma unroll
    for (int i = 1; i < N1; ++i) {
      p = r[i]; t = r[NV + i];
      w0 = __builtin_fma(c[i], p.x, w0); w1 = __builtin_fma(c[i], p.y, w1);
      w2 = __builtin_fma(c[i], t.x, w2); w3 = __builtin_fma(c[i], t.y, w3);
    }
    sB0[tv] = make_double2(w0, w1);
    sB1[tv] = make_double2(w2, w3);
  }
  __syncthreads();
  // stage 2: Uq[a + N1 b] = sum_j IQ[a,j] W[b + N1 j]   (this lane: column b of W)
  {
    const d2* r = sB0 + colq;
    const d2* r1 = sB1 + colq;
    d2 p = r[0], t = r1[0];
    U[0] = c[0] * p.x; U[1] = c[0] * p.y; U[2] = c[0] * t.x; U[3] = c[0] * t.y;
#pragma unroll
    for (int j = 1; j < N1; ++j) {
      p = r[N1 * j]; t = r1[N1 * j];
      U[0] = __builtin_fma(c[j], p.x, U[0]); U[1] = __builtin_fma(c[j], p.y, U[1]);
      U[2] = __builtin_fma(c[j], t.x, U[2]); U[3] = __builtin_fma(c[j], t.y, U[3]);
    }
  }
}

// The kernels are PERSISTENT: a workgroup (= one group of GW waves) walks the element groups blockIdx.x, blockIdx.x +
// gridDim.x, ... and issues the global loads of its next group before it computes on the current one.  One-shot
// workgroups had only the loads of the groups that happened to be in their first phase in flight (measured: kt2_sigma
// at 3.4 TB/s of its own traffic with 12-16 waves per CU, no faster with more waves); with the prefetch every resident
// workgroup keeps ~8 KB in flight all the time, and the per-lane table rows and index arithmetic are paid once.
// The loads of the NEXT group go into the registers of the current one right after their last use (no second register
// set, no copies that would have to wait).  What is left on the table: hipcc drains all outstanding vector-memory
// operations (s_waitcnt vmcnt(0)) at the loop head, i.e. also this iteration's stores (~19 % of an iteration in
// kt2_sigma by s_memtime stamps); inline-asm loads with counted waits remove that wait, but hipcc then copies the asm
// destinations between registers before the data has landed (audited in the .s), so the loads stay compiler-managed.
// grid = what the device holds at once (occupancy query; ESDG_T2_WG_PER_CU overrides for experiments)
// Tuning knobs of the A/B builds (esdg_api.hip under -DESDG_AB_HOOKS sets them from the environment through ab_tuning_t2; the
// shipped library never changes them): persistent workgroups per CU (0 = the occupancy query), slots the interior launch of a
// sharded schedule leaves to its boundary strips.
int g_wg_per_cu = 0, g_reserve = 64;
template <auto kernel>
__host__ inline int persistent_grid(int threads, int64_t ngroups) {
  static int per_cu = 0, cus = 0;     // one pair per kernel (the kernel is the template argument)
  if (!per_cu) {
    int dev = 0;
    cus = 256;
    if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    int n = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, kernel, threads, 0) != hipSuccess || n < 1) n = 2;
    if (g_wg_per_cu > 0) n = g_wg_per_cu;   // (A/B builds only: esdg_ab_tuning)
    per_cu = n;
  }
  const int64_t g = (int64_t)cus * per_cu;
  return (int)(ngroups < g ? ngroups : g);
}

// Diagnostic build only (-DESDG_T2_STAMP): per-phase wave cycles (s_memtime) summed into a buffer nothing else reads.
#ifdef ESDG_T2_STAMP
__device__ unsigned long long g_stamp[16];
#define T2_STAMP(i)                                                                          \
  do {                                                                                       \
    unsigned long long t_;                                                                   \
    __builtin_amdgcn_sched_barrier(0);                                                       \
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");              \
    __builtin_amdgcn_sched_barrier(0);                                                       \
    t_acc[i] += t_ - t_prev;                                                                 \
    t_prev = t_;                                                                             \
  } while (0)
#define T2_STAMP_INIT                                                                        \
  unsigned long long t_prev, t_acc[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};               \
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_prev)::"memory")
// (a sample of the workgroups: 12 same-address atomics from each of the 52429 workgroups of a one-shot launch at cfg3
// would take ~7 ms and distort every wait)
#define T2_STAMP_FLUSH                                                                       \
  if (threadIdx.x == 0 && (blockIdx.x & 31) == 0)                                            \
    for (int i_ = 0; i_ < 12; ++i_) atomicAdd(&g_stamp[i_], t_acc[i_])
#else
#define T2_STAMP(i)
#define T2_STAMP_INIT
#define T2_STAMP_FLUSH
#endif

// ---------------------------------------------------------------------------------------------------------------------
// phase 1 (CNS, meshes without walls): sigma = K(v) grad v -> normal-stress traces B and the volume part of div sigma
// (rhs_viscous! :749-815 in collocated form, dg_grad! :548-569)
// ---------------------------------------------------------------------------------------------------------------------
// ---- meshes with walls: gradient and volume divergence of the elements that touch a wall, in the NODAL basis --------------
// dg_grad! / dg_div! (cavity_optimized.jl:549-611) act on nodal coefficients: Dr, Ds, LIFT, then rows 1:Np of the metric
// arrays and 1/J[i,e] node by node, then Vq.  Everywhere else the kernel works at the Gauss nodes with one geometry record
// per element (Vq Dr Pq as a 1D operator along the lines) -- the same up to the round-off of the driver's set-up in those
// arrays (5e-14 relative at 8x8 elements, 5e-13 at 64x64), which shows only where the lifted wall jump dominates and the
// reference is almost exact: the elements with a boundary node (MeshDev::wgeo).  For those, kt2_sigma takes its Gauss-node
// pieces -- the two line derivatives and the lift, Vq (Dr VU), Vq (Ds VU), Vq (LIFT ...) -- back to nodal coefficients (Pq),
// scales them node by node as the reference does, and returns to the Gauss nodes (Vq); the same for the volume part of the
// divergence.  Pq = IP (x) IP and Vq = IQ (x) IQ by sum factorisation, the 1D operators in LDS (staged at kernel entry).
//
// One such product for NP pair planes: stage 1 reads `in` at o1 + k s1 with the weights w1[k], writes `mid` at the lane's own
// slot; stage 2 reads `mid` at o2 + k s2 with w2[k].  The caller's barrier precedes the call (in complete); one barrier inside.
template <int N1, int NV, int NP>
__device__ __forceinline__ void tp_apply(const double* w1, const double* w2, const d2* in, d2* mid, unsigned tv, unsigned o1, unsigned s1,
                                         unsigned o2, unsigned s2, double* out) {
  double t[2 * NP];
#pragma unroll
  for (int c = 0; c < 2 * NP; ++c) t[c] = 0.0;
#pragma unroll
  for (int k = 0; k < N1; ++k) {
    const double m = w1[k];
#pragma unroll
    for (int p = 0; p < NP; ++p) {
      const d2 v = in[p * NV + o1 + k * s1];
      t[2 * p] = __builtin_fma(m, v.x, t[2 * p]); t[2 * p + 1] = __builtin_fma(m, v.y, t[2 * p + 1]);
    }
  }
#pragma unroll
  for (int p = 0; p < NP; ++p) mid[p * NV + tv] = make_double2(t[2 * p], t[2 * p + 1]);
  __syncthreads();
#pragma unroll
  for (int c = 0; c < 2 * NP; ++c) out[c] = 0.0;
#pragma unroll
  for (int k = 0; k < N1; ++k) {
    const double m = w2[k];
#pragma unroll
    for (int p = 0; p < NP; ++p) {
      const d2 v = mid[p * NV + o2 + k * s2];
      out[2 * p] = __builtin_fma(m, v.x, out[2 * p]); out[2 * p + 1] = __builtin_fma(m, v.y, out[2 * p + 1]);
    }
  }
}

#ifndef ESDG_T2_SIGMA_WPE
#define ESDG_T2_SIGMA_WPE 1   // minimum waves per SIMD asked of the register allocator (A/B hook)
#endif
// kt2_sigma at N1 = 5 without walls: the per-lane rows of IQ and of the face extrapolation (20 VGPRs held across the persistent
// loop) live in LDS and are read at the point of use, which brings the kernel from 184 to <= 168 VGPRs = three waves per SIMD
// (ESDG_T2_SIGMA_ROWS_LDS=0: rows in registers, two waves per SIMD, the form of every other instantiation).
#ifndef ESDG_T2_SIGMA_DEFER_STORES
#define ESDG_T2_SIGMA_DEFER_STORES 1   // (see the comment at the kernel's store stage)
#endif
#ifndef ESDG_T2_SIGMA_COALESCE
#define ESDG_T2_SIGMA_COALESCE 0   // (measured late in round 3: phase 1 +2 % at N=4, -2 ... -5 % at N=2, 3, 6: hook, off)
#endif
#ifndef ESDG_T2_SIGMA_ROWS_LDS
#define ESDG_T2_SIGMA_ROWS_LDS 0
#endif
template <int N1, bool WALLS> struct SigmaCfg {
  static constexpr bool ROWS_LDS = ESDG_T2_SIGMA_ROWS_LDS && N1 == 5 && !WALLS;
  static constexpr int WPE = ROWS_LDS ? 3 : ESDG_T2_SIGMA_WPE;
};
// FULL = true: persistent over the complete groups of the launch's element range; every lane holds valid data (lanes
// beyond a group's slots duplicate slot tid - NV / tid % NF), so nothing is masked -- duplicate lanes store the same
// value to the same address -- and no branch hides the outstanding-store count from the compiler's s_waitcnt placement.
// FULL = false: the one partial group at the end of the range (one workgroup, masked stores); same arithmetic.
// WALLS: boundary nodes (M.bc): exterior entropy variables by the wall closure; at a boundary node B holds MINUS the
// prescribed stress jump, so that the last phase's .5*(-B[mapP] - B[own]) with mapP = own gives the jump unchanged.
template <int N1, bool FULL, bool WALLS>
__global__ __launch_bounds__(GeoS<N1>::GT, (SigmaCfg<N1, WALLS>::WPE)) void kt2_sigma(TensorTables TT, MeshDev M, Phys ph, const double* __restrict__ Q,
                                                        const double* __restrict__ A_U, double* __restrict__ B,
                                                        double* __restrict__ SG, double* __restrict__ vt_partial) {
  using G = GeoS<N1>;
  constexpr int Nq = G::Nq, Nfq = G::Nfq, E = G::E, NV = G::NV, NF = G::NF;
  constexpr NodeLayout NL(N1);
  constexpr FaceLayout FL(N1);
  // vt_partial != null (uniform; esdg_viscous_entropy_test only): the launch also reduces rhs_viscous!'s second return,
  // visc_test = sum(wJq .* (VUx .* sigma_x + VUy .* sigma_y)) (:802-806), over the elements it owns -- one partial per workgroup,
  // summed in a fixed order (lane, then the workgroup's waves), so a context's value is reproducible run to run.
  // LDS arena in doubles.  R0: Vq scratch A|B (4 pair planes), later sigma (3 pair planes: (sx0,sx1) (sx2,sy0) (sy1,sy2))
  // and the pair plane of S^0; R1: pair plane of S^1, single planes S^0_2, S^1_2; then V (pair (v2,v3) + single v4),
  // the half jumps (pair + single, face nodes) and the geometry records of the group's elements.
  constexpr int NVP = NV + (NV & 1), NFP = NF + (NF & 1);   // single planes padded to an even length: pair planes stay 16-B aligned
  constexpr int NGEO = E * GEO_STRIDE, GPT = (NGEO + G::GT - 1) / G::GT;   // geometry doubles of a group / per thread
  // (RG: from N1 = 9 on the six face planes are shorter than the three node planes the wall scratch borrows from them)
  constexpr int R0 = 0, R1 = 8 * NV, RV = R1 + 2 * NV + 2 * NVP, RD = RV + 2 * NV + NVP,
                RG = (RD + 6 * NF < 18 * NV) ? 18 * NV : RD + 6 * NF,
                RE = RG + GPT * G::GT, RW = RE + (WALLS ? (E + 2) / 2 : 0), RT = RW + (WALLS ? 2 * N1 * N1 : 0);
  constexpr bool ROWS_LDS = SigmaCfg<N1, WALLS>::ROWS_LDS;
  constexpr int N1P = N1 + (N1 & 1);                   // rows padded to an even length: N1P / 2 ds_read_b128 per row
  constexpr int RTQ = RT + (RT & 1), RTE = RTQ + Nq * N1P, RBS = ROWS_LDS ? RTE + Nfq * N1P : RT;
  // COAL (ESDG_T2_SIGMA_COALESCE, with the deferred stores): a group's normal-stress records [NF][3] leave through an LDS block, so
  // that a store instruction writes 1 KB of consecutive doubles instead of 8 B per lane at a stride of 24 B
  constexpr bool COAL = ESDG_T2_SIGMA_COALESCE && ESDG_T2_SIGMA_DEFER_STORES;
  constexpr int NBS = B_NC * NF, BPT = (NBS + G::GT - 1) / G::GT, NLDS = RBS + (COAL ? NBS : 0);
  __shared__ __align__(16) double lds[NLDS];
  int* sEb = reinterpret_cast<int*>(lds + RE);        // [E] WALLS: element has a boundary node (see wall_dense above)
  double* sW = lds + RW;                              // WALLS: the 1D operators IQ | IP, row-major (for tp_apply above)
  static_assert(R0 + 18 * NV <= RG, "scratch of the nodal-basis divergence ends before the geometry records");
  d2* sSg = reinterpret_cast<d2*>(lds + R0);          // [3][NV] sigma pairs
  d2* sS0 = reinterpret_cast<d2*>(lds + R0 + 6 * NV); // [NV] (S^0_0, S^0_1)
  d2* sS1 = reinterpret_cast<d2*>(lds + R1);          // [NV] (S^1_0, S^1_1)
  double* sS02 = lds + R1 + 2 * NV;                   // [NV] S^0_2
  double* sS12 = lds + R1 + 2 * NV + NVP;             // [NV] S^1_2
  d2* sVp = reinterpret_cast<d2*>(lds + RV);          // [NV] (v2, v3)
  double* sV4 = lds + RV + 2 * NV;                    // [NV] v4
  // half jumps times the face node's own normal (the reference multiplies per node, dg_grad! :560-566), three pair planes:
  d2* sDx = reinterpret_cast<d2*>(lds + RD);          // [NF] nxJ * half jump of (v2, v3)
  d2* sD4 = reinterpret_cast<d2*>(lds + RD + 2 * NF); // [NF] (nxJ, nyJ) * half jump of v4
  d2* sDy = reinterpret_cast<d2*>(lds + RD + 4 * NF); // [NF] nyJ * half jump of (v2, v3)
  double* sGeo = lds + RG;
  static_assert((R1 % 2) == 0 && (RV % 2) == 0 && (RD % 2) == 0 && (RG % 2) == 0, "pair planes must be 16-byte aligned");

#ifdef ESDG_T2_POISON   // diagnostic build: LDS starts as NaN, so a read of a slot nobody wrote shows in the result
  for (int i = threadIdx.x; i < NLDS; i += G::GT) lds[i] = __builtin_nan("");
  __syncthreads();
#endif
  const unsigned tid = threadIdx.x;
  // Lanes beyond the group's NV volume / NF face slots redo the work of slot tid - NV / tid % NF (same loads, same
  // values: their LDS writes are duplicates), slots of elements beyond the mesh compute on the data of slot 0: no
  // divergent regions, only the global stores are masked.
  const unsigned tv = tid < (unsigned)NV ? tid : tid - NV;              // volume slot
  const unsigned tf = tid < (unsigned)NF ? tid : tid % NF;              // face slot
  const unsigned ev = tv / Nq, q = tv - ev * Nq, a = q % N1, b = q / N1;
  const unsigned ef = tf / Nfq, fn = tf - ef * Nfq;
  const unsigned rowb = ev * Nq + N1 * b, colb = ev * Nq + a;           // first node of this lane's d = 0 / d = 1 line
  const int64_t KN = M.K * Nq;

  // per-lane table rows, once per workgroup
  const double* nd_ = TT.node_d + q;      // entry-major tables: entry i of node q at [i * Nq + q]
  const int* ni_ = TT.node_i + q;
  const double* fd_ = TT.face_d + fn;
  const int* fi_ = TT.face_i + fn;
  double cq_r[N1], dg0[N1], dg1[N1], ee_r[N1], lw[4];
#pragma unroll
  for (int i = 0; i < N1; ++i) { cq_r[i] = nd_[(NL.IQ + i) * Nq]; dg0[i] = nd_[(NL.DG0 + i) * Nq]; dg1[i] = nd_[(NL.DG1 + i) * Nq]; ee_r[i] = fd_[(FL.EE + i) * Nfq]; }
  if (ROWS_LDS) {   // row q of IQ's per-node table and row fn of the face extrapolation, by every lane that holds one (same values)
#pragma unroll
    for (int i = 0; i < N1; ++i) { lds[RTQ + q * N1P + i] = cq_r[i]; lds[RTE + fn * N1P + i] = ee_r[i]; }
    if (N1P > N1) { lds[RTQ + q * N1P + N1] = 0.0; lds[RTE + fn * N1P + N1] = 0.0; }
  }
  // (rows in registers: the arrays themselves; rows in LDS: re-read where they are used, first after the loop's first barrier)
  auto row_of = [&](const double* reg, int base, double* out) {
    if (ROWS_LDS) {
      const d2* r = reinterpret_cast<const d2*>(lds + base);
#pragma unroll
      for (int i = 0; i < N1P / 2; ++i) { const d2 t = r[i]; out[2 * i] = t.x; if (2 * i + 1 < N1) out[2 * i + 1] = t.y; }
    } else {
#pragma unroll
      for (int i = 0; i < N1; ++i) out[i] = reg[i];
    }
  };
  unsigned fq[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) { lw[k] = nd_[(NL.LW + k) * Nq]; fq[k] = ev * Nfq + ni_[(NL.FQ + k) * Nq]; }
  const unsigned fnode0 = ef * Nq + fi_[(FL.NODE0) * Nfq], fstride = fi_[(FL.STRIDE) * Nfq];

  const int64_t e_end = M.e_begin + M.e_count;
  // FULL: ceil(e_count / E) groups, the last one shifted back so that it is complete too: it recomputes up to E - 1
  // elements of the group before it and stores the same values to the same addresses (plain stores of slot-independent
  // results) -- no masked partial group, no second launch.  FULL = false serves ranges of fewer than E elements only.
  const int64_t nfull = FULL ? (M.e_count + E - 1) / E : 0;
  const int64_t ngrp = FULL ? nfull : nfull + 1;
  const int64_t e_last = M.e_begin + M.e_count - E;   // base of the shifted last group (FULL)
  // ---- prologue: loads of the first group ---------------------------------------------------------------------------
  // Prefetch without a second register set: the loads of the NEXT group go into the registers of the current one right
  // after their last use (x and the geometry are written to LDS first thing; the neighbour traces are consumed by the
  // face-jump stage), so that every load has most of an iteration to land and no copy ever waits for one.
  // group of this workgroup in round k: k G + w, or (MeshDev::wall_rot) k G + ((w + k rot) mod G)
  const int64_t rot = (WALLS && FULL) ? (int64_t)M.wall_rot : 0;
  // (ESDG_T2_XCD_REMAP, meshes without walls: the workgroups of XCD x = blockIdx % 8 share the x-th contiguous eighth of the groups,
  // workgroup i of the XCD's nx takes its groups i, i + nx, ...; exhausted -> ngrp, which ends the loop)
  const bool xr = ESDG_T2_XCD_REMAP && FULL && !WALLS && nfull >= 8 * T2_NXCD && gridDim.x >= (unsigned)T2_NXCD;
  const int64_t xx = blockIdx.x % T2_NXCD, xi = blockIdx.x / T2_NXCD;
  const int64_t xnw = gridDim.x / T2_NXCD + (xx < (int64_t)(gridDim.x % T2_NXCD) ? 1 : 0);           // workgroups on this XCD
  const int64_t xq = nfull / T2_NXCD, xrm = nfull % T2_NXCD;
  const int64_t xs = xx * xq + (xx < xrm ? xx : xrm), xc = xq + (xx < xrm ? 1 : 0);                     // this XCD's groups: start, count
  auto group_of = [&](int64_t k) -> int64_t {
    if (xr) { const int64_t j = xi + k * xnw; return j < xc ? xs + j : ngrp; }
    return k * (int64_t)gridDim.x + ((int64_t)blockIdx.x + k * rot) % (int64_t)gridDim.x;
  };
  int64_t rnd = 0;
  int64_t grp = FULL ? group_of(0) : nfull;
  double x[4], geo[GPT];
  float2 ndn;        // this lane's face-node normal (nxJ, nyJ) minus the face mean (MeshDev::fnd), prefetched like the rest
  float sdn = 0.f;   // the same for sJ (wall closures only)
  d2 up0, up1;
  {
    const int64_t e0 = FULL ? min(M.e_begin + grp * E, e_last) : M.e_begin;
    const int nE = FULL ? E : (int)(e_end - e0);
    const unsigned tvl = tv < (unsigned)(nE * Nq) ? tv : 0u, tfl = tf < (unsigned)(nE * Nfq) ? tf : 0u;
    const unsigned mp = ESDG_EWN((unsigned)M.mapP[ESDG_EW(e0) * Nfq + tfl], Nfq);
#pragma unroll
    for (int f = 0; f < 4; ++f) x[f] = Q[f * KN + ESDG_EW(e0) * Nq + tvl];
#pragma unroll
    for (int i = 0; i < GPT; ++i) { const unsigned n = tid + i * G::GT; geo[i] = M.geo[ESDG_EW(e0) * GEO_STRIDE + (n < (unsigned)(nE * GEO_STRIDE) ? n : 0u)]; }
    const d2* up = reinterpret_cast<const d2*>(A_U + (size_t)mp * FAU_NC);
    up0 = up[0]; up1 = up[1];
    ndn = reinterpret_cast<const float2*>(M.fnd)[ESDG_EW(e0) * Nfq + tfl];
    if (WALLS) sdn = M.fsd[ESDG_EW(e0) * Nfq + tfl];
  }
  const unsigned gfo = 5 + 3 * (fn / N1);

  // The stores of a group are issued at the top of the NEXT iteration (ESDG_T2_SIGMA_DEFER_STORES, default on): on gfx9
  // loads and stores share the vmcnt counter and complete out of order with respect to each other, so the wait for the
  // prefetched loads at the loop head is a vmcnt(0) that also drains every store issued since -- with the stores at the end
  // of the iteration that was ~19 % of an iteration (round-2 stamps).  Issued right after that wait, and before the next
  // prefetch loads, they have a whole iteration to drain.
#ifndef ESDG_T2_SIGMA_DEFER_STORES
#define ESDG_T2_SIGMA_DEFER_STORES 1
#endif
  if (WALLS && M.wgeo) {   // IQ | IP for the nodal-basis path (read many barriers further down)
    constexpr TensorLayout TL(N1);
    static_assert(TL.IP == TL.IQ + N1 * N1, "IQ and IP are adjacent in the 1D tables");
    for (int i = tid; i < 2 * N1 * N1; i += G::GT) sW[i] = TT.dbl[TL.IQ + i];
  }
  double vt = 0.0;   // this lane's share of visc_test
  double cdv[3] = {0, 0, 0}, csn[3] = {0, 0, 0};
  int64_t ce0 = 0;
  bool cva = false, cfa = false;
  int cnb = 0;                      // COAL: doubles of the stored group's B block (uniform)
  double* sBs = lds + RBS;
  T2_STAMP_INIT;
#pragma unroll 1
  for (; grp < ngrp; grp = FULL ? group_of(++rnd) : ngrp) {
    T2_STAMP(0);
    const int64_t e0 = FULL ? min(M.e_begin + grp * E, e_last) : M.e_begin;
    const int nE = FULL ? E : (int)(e_end - e0);
    const bool vact = FULL || tid < (unsigned)(nE * Nq), fact = FULL || tid < (unsigned)(nE * Nfq);
    // next group, clamped to the last one (harmless re-loads at the end)
    const int64_t gnx = min(FULL ? group_of(rnd + 1) : ngrp, ngrp - 1);
    const int64_t e0n = FULL ? min(M.e_begin + gnx * E, e_last) : M.e_begin;
    const int nEn = FULL ? E : (int)(e_end - e0n);
    const unsigned tvn = tv < (unsigned)(nEn * Nq) ? tv : 0u, tfn = tf < (unsigned)(nEn * Nfq) ? tf : 0u;
    int bcf = 0;
    double vlid = 1.0;
    if (WALLS) {   // boundary flag (and lid velocity) of this lane's face node
      const int64_t nfb = ESDG_EW(e0) * Nfq + (tf < (unsigned)(nE * Nfq) ? tf : 0u);
      bcf = M.bc[nfb];
      if (M.vlid) vlid = M.vlid[nfb];
    }

    // ---- this group's state and geometry to LDS; their registers take the next group's loads ---------------------------
    d2* sA = reinterpret_cast<d2*>(lds + R0);
    if (WALLS && tid < (unsigned)E) sEb[tid] = 0;   // (set by the face lanes two barriers further down)
    double bt[BPT];
    if (COAL) {   // the previous group's B block, consecutive doubles per lane (read before the staging writes below are queued)
#pragma unroll
      for (int r = 0; r < BPT; ++r) bt[r] = sBs[min((int)tid + r * G::GT, NBS - 1)];
    }
#pragma unroll
    for (int i = 0; i < GPT; ++i) sGeo[tid + i * G::GT] = geo[i];
    sA[tv] = make_double2(x[0], x[1]);
    sA[NV + tv] = make_double2(x[2], x[3]);
    __builtin_amdgcn_sched_barrier(0);
    if (ESDG_T2_SIGMA_DEFER_STORES) {   // the previous group's results (none in the first iteration: cva = cfa = false)
#ifdef ESDG_EXP_NOSG   // (bound experiment, wrong results: the divergence stored into an L2-resident window)
      if (cva) { double* o = SG + (ce0 & 1023) * Nq + tv; o[0] = cdv[0]; o[KN] = cdv[1]; o[2 * KN] = cdv[2]; }
#else
      if (cva) { double* o = SG + ESDG_EW(ce0) * Nq + tv; o[0] = cdv[0]; o[KN] = cdv[1]; o[2 * KN] = cdv[2]; }
#endif
      if (COAL) {
        double* bb = B + ESDG_EW(ce0) * Nfq * B_NC;
#pragma unroll
        for (int r = 0; r < BPT; ++r) { const int idx = (int)tid + r * G::GT; if (idx < cnb) bb[idx] = bt[r]; }
      } else if (cfa) { double* bb = B + trace_slot<N1>(M, ESDG_EW(ce0) + ef, fn) * B_NC; bb[0] = csn[0]; bb[1] = csn[1]; bb[2] = csn[2]; }
      __builtin_amdgcn_sched_barrier(0);
    }
    // mapP first: it is waited for first (vmcnt counts in issue order), the others may then still be in flight.
    // unsigned: a sign-extending load would put its shift, and with it the wait for the load, right here
    const unsigned mpn = ESDG_EWN((unsigned)M.mapP[ESDG_EW(e0n) * Nfq + tfn], Nfq);
    __builtin_amdgcn_sched_barrier(0);
    const float2 nd = ndn;   // this group's normal differences; the registers take the next group's
    const float sd = sdn;
    ndn = reinterpret_cast<const float2*>(M.fnd)[ESDG_EW(e0n) * Nfq + tfn];
    if (WALLS) sdn = M.fsd[ESDG_EW(e0n) * Nfq + tfn];
#pragma unroll
    for (int f = 0; f < 4; ++f) x[f] = Q[f * KN + ESDG_EW(e0n) * Nq + tvn];
#pragma unroll
    for (int i = 0; i < GPT; ++i) { const unsigned n = tid + i * G::GT; geo[i] = M.geo[ESDG_EW(e0n) * GEO_STRIDE + (n < (unsigned)(nEn * GEO_STRIDE) ? n : 0u)]; }
    __builtin_amdgcn_sched_barrier(0);
    T2_STAMP(1);

    // ---- state at the Gauss node, entropy variables 2..4 --------------------------------------------------------------
    double U[4];
    {   // (a lane reads the row it staged itself -- other lanes' copies of it hold the same values: no barrier needed)
      double cq[N1];
      row_of(cq_r, RTQ + q * N1P, cq);
      vq_apply<N1, NV>(cq, sA, reinterpret_cast<d2*>(lds + R0 + 4 * NV), reinterpret_cast<d2*>(lds + R0 + 6 * NV), tv, rowb, ev * Nq + b, nullptr, U);
    }
    T2_STAMP(2);
    double V[3];
    {
      const double m2 = U[1] * U[1] + U[2] * U[2];
      const double rre = __builtin_fma(U[0], U[3], -.5 * m2);      // rho * rhoe
      const double t = U[0] * rcp_refined(rre);                    // rho / (rho rhoe) = 1 / rhoe
      V[0] = U[1] * t; V[1] = U[2] * t; V[2] = -(U[0] * t);       // (rho u, rho v, -rho) / rhoe  (cavity :464-467)
    }
    sVp[tv] = make_double2(V[0], V[1]);
    sV4[tv] = V[2];
    __syncthreads();
    T2_STAMP(3);

    // ---- face lanes: projected entropy variables at the face node, half jump to the neighbour's ---------------------
    double nr[3];
    {
      const double b2 = 2 * GM1 * up1.y;                           // neighbour: (v2,v3,v4) = (b u, b v, -b), b = 2 (gamma-1) beta
      double vP[3] = {b2 * up0.y, b2 * up1.x, -b2};
      {   // the neighbour traces of the next group (its mapP entries were requested two barriers ago) into the registers
          // of this group's, whose last use is forced to lie above this point
        asm volatile("" : "+v"(vP[0]), "+v"(vP[1]), "+v"(vP[2]));
        __builtin_amdgcn_sched_barrier(0);
        const d2* upn = reinterpret_cast<const d2*>(A_U + (size_t)mpn * FAU_NC);
        up0 = upn[0]; up1 = upn[1];
        __builtin_amdgcn_sched_barrier(0);
      }
      double ee[N1];
      row_of(ee_r, RTE + fn * N1P, ee);
      d2 p = sVp[fnode0];
      double vf0 = ee[0] * p.x, vf1 = ee[0] * p.y, vf2 = ee[0] * sV4[fnode0];
#pragma unroll
      for (int j = 1; j < N1; ++j) {
        p = sVp[fnode0 + j * fstride];
        vf0 = __builtin_fma(ee[j], p.x, vf0); vf1 = __builtin_fma(ee[j], p.y, vf1);
        vf2 = __builtin_fma(ee[j], sV4[fnode0 + j * fstride], vf2);
      }
      const double* gm = sGeo + ef * GEO_STRIDE + gfo;   // face means; + this node's difference = its own normal, exactly
      nr[0] = gm[0] + (double)nd.x; nr[1] = gm[1] + (double)nd.y; nr[2] = WALLS ? gm[2] + (double)sd : gm[2];
      if (WALLS && bcf) {
        const double vfo[3] = {vf0, vf1, vf2};
        wall_exterior_v(vfo, bcf, vlid, nr, ph, vP);
        sEb[ef] = 1;
      }
      const double h0 = .5 * (vP[0] - vf0), h1 = .5 * (vP[1] - vf1), h2 = .5 * (vP[2] - vf2);
      sDx[tf] = make_double2(nr[0] * h0, nr[0] * h1);     // (duplicate lanes: duplicate writes)
      sD4[tf] = make_double2(nr[0] * h2, nr[1] * h2);
      sDy[tf] = make_double2(nr[1] * h0, nr[1] * h1);
    }
    __syncthreads();
    T2_STAMP(4);

    // ---- volume lanes: BR1 gradient, sigma = K(v) grad v --------------------------------------------------------------
    bool gb = false;   // WALLS: some element of the group has a boundary node (uniform)
    double wgm[5] = {0, 0, 0, 0, 1};   // ... and then this lane's nodal metric terms and J (MeshDev::wgeo)
    const double* g = sGeo + ev * GEO_STRIDE;     // elements beyond the mesh read the (clamped) staged values: finite, unused
    const double gx0 = g[TT.op0], gy0 = g[2 + TT.op0], gx1 = g[TT.op1], gy1 = g[2 + TT.op1];
    double sgx[3], sgy[3];
    {
      double d0[3], d1[3], tx[3], ty[3];
      {
        d2 p = sVp[rowb], t = sVp[colb];
        d0[0] = dg0[0] * p.x; d0[1] = dg0[0] * p.y; d0[2] = dg0[0] * sV4[rowb];
        d1[0] = dg1[0] * t.x; d1[1] = dg1[0] * t.y; d1[2] = dg1[0] * sV4[colb];
#pragma unroll
        for (int j = 1; j < N1; ++j) {
          p = sVp[rowb + j]; t = sVp[colb + N1 * j];
          d0[0] = __builtin_fma(dg0[j], p.x, d0[0]); d0[1] = __builtin_fma(dg0[j], p.y, d0[1]);
          d0[2] = __builtin_fma(dg0[j], sV4[rowb + j], d0[2]);
          d1[0] = __builtin_fma(dg1[j], t.x, d1[0]); d1[1] = __builtin_fma(dg1[j], t.y, d1[1]);
          d1[2] = __builtin_fma(dg1[j], sV4[colb + N1 * j], d1[2]);
        }
      }
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        tx[c] = __builtin_fma(gx1, d1[c], gx0 * d0[c]);
        ty[c] = __builtin_fma(gy1, d1[c], gy0 * d0[c]);
      }
      double lxs[3] = {0, 0, 0}, lys[3] = {0, 0, 0};   // WALLS: the lift on its own (the nodal-basis path below needs it)
#pragma unroll
      for (int k = 0; k < 4; ++k) {   // lift of the (normal x half jump)s on the four faces at the ends of this node's lines
        const d2 jx = sDx[fq[k]], j4 = sD4[fq[k]], jy = sDy[fq[k]];
        if (WALLS) {
          lxs[0] = __builtin_fma(lw[k], jx.x, lxs[0]); lys[0] = __builtin_fma(lw[k], jy.x, lys[0]);
          lxs[1] = __builtin_fma(lw[k], jx.y, lxs[1]); lys[1] = __builtin_fma(lw[k], jy.y, lys[1]);
          lxs[2] = __builtin_fma(lw[k], j4.x, lxs[2]); lys[2] = __builtin_fma(lw[k], j4.y, lys[2]);
        } else {
          tx[0] = __builtin_fma(lw[k], jx.x, tx[0]); ty[0] = __builtin_fma(lw[k], jy.x, ty[0]);
          tx[1] = __builtin_fma(lw[k], jx.y, tx[1]); ty[1] = __builtin_fma(lw[k], jy.y, ty[1]);
          tx[2] = __builtin_fma(lw[k], j4.x, tx[2]); ty[2] = __builtin_fma(lw[k], j4.y, ty[2]);
        }
      }
      if (WALLS) {
#pragma unroll
        for (int c = 0; c < 3; ++c) { tx[c] += lxs[c]; ty[c] += lys[c]; }
      }
      const double iJ = rcp_refined(g[4]);
#pragma unroll
      for (int c = 0; c < 3; ++c) { tx[c] *= iJ; ty[c] *= iJ; }
      if (WALLS && M.wgeo) {   // (uniform) elements with a boundary node: the gradient as dg_grad! forms it, in the nodal basis
        gb = __syncthreads_or(bcf != 0) != 0;
        if (gb) {
          // this lane's NODE: metric terms of tensor direction 0 / 1 (x, y parts) and 1/J as the driver holds them; requested
          // here, needed after the first product below (and once more by the divergence)
          const double* wg = M.wgeo + (ESDG_EW(e0) + (ev < (unsigned)nE ? ev : 0u)) * 5 * Nq + q;
          wgm[0] = wg[TT.op0 * Nq]; wgm[1] = wg[(2 + TT.op0) * Nq]; wgm[2] = wg[TT.op1 * Nq]; wgm[3] = wg[(2 + TT.op1) * Nq];
          wgm[4] = wg[4 * Nq];
          d2* sc = reinterpret_cast<d2*>(lds + R0);   // 3 + 3 pair planes: the Vq scratch and what will hold S^0, S^1 (all free here)
          d2* sm = sc + 3 * NV;
          const double* ipa = sW + N1 * N1 + a * N1;   // IP[a][:], IP[b][:], IQ[a][:]
          const double* ipb = sW + N1 * N1 + b * N1;
          const double* iqa = sW + a * N1;
          const unsigned colq = ev * Nq + b;
          double o[12];   // derivative of VU along tensor direction 0 [3], direction 1 [3] (Dr / Ds VU), LIFT (.5 (vP - vf) nxJ) [3],
                          // LIFT (... nyJ) [3], all at this lane's NODE
          sc[tv] = make_double2(d0[0], d0[1]); sc[NV + tv] = make_double2(d0[2], d1[0]); sc[2 * NV + tv] = make_double2(d1[1], d1[2]);
          __syncthreads();
          tp_apply<N1, NV, 3>(ipa, ipb, sc, sm, tv, colq, N1, colb, N1, o);
          __syncthreads();
          sc[tv] = make_double2(lxs[0], lxs[1]); sc[NV + tv] = make_double2(lxs[2], lys[0]); sc[2 * NV + tv] = make_double2(lys[1], lys[2]);
          __syncthreads();
          tp_apply<N1, NV, 3>(ipa, ipb, sc, sm, tv, colq, N1, colb, N1, o + 6);
          const double iJn = rcp_refined(wgm[4]);
          double th[6];
#pragma unroll
          for (int c = 0; c < 3; ++c) {   // (rxj*ur + sxj*us + surf) / J  (:559-566)
            th[c] = (__builtin_fma(wgm[2], o[3 + c], wgm[0] * o[c]) + o[6 + c]) * iJn;
            th[3 + c] = (__builtin_fma(wgm[3], o[3 + c], wgm[1] * o[c]) + o[9 + c]) * iJn;
          }
          __syncthreads();
          sc[tv] = make_double2(th[0], th[1]); sc[NV + tv] = make_double2(th[2], th[3]); sc[2 * NV + tv] = make_double2(th[4], th[5]);
          __syncthreads();
          double tn[6];   // VUx = Vq * VUx (:779-780)
          tp_apply<N1, NV, 3>(iqa, iqa, sc, sm, tv, rowb, 1, colq, N1, tn);
          if (sEb[ev]) {
#pragma unroll
            for (int c = 0; c < 3; ++c) { tx[c] = tn[c]; ty[c] = tn[3 + c]; }
          }
          __syncthreads();   // the scratch planes take sigma below
        }
      }
      viscous_stress(V, tx, ty, -ph.lambda, ph.mu, ph.kappa, sgx, sgy);
      if (vt_partial) {   // (uniform) every element once: not the duplicate lanes, not the elements a shifted last group repeats
        const int64_t ee = e0 + ev, enom = FULL ? M.e_begin + grp * E : e0;
        if (tid < (unsigned)NV && ee >= enom && (FULL || ev < (unsigned)nE)) {
          double t = tx[0] * sgx[0];
          t = __builtin_fma(tx[1], sgx[1], t); t = __builtin_fma(tx[2], sgx[2], t);
          t = __builtin_fma(ty[0], sgy[0], t); t = __builtin_fma(ty[1], sgy[1], t); t = __builtin_fma(ty[2], sgy[2], t);
          vt = __builtin_fma(M.wJq[ESDG_EW(ee) * Nq + q], t, vt);
        }
      }
    }
    {
      sSg[tv] = make_double2(sgx[0], sgx[1]);
      sSg[NV + tv] = make_double2(sgx[2], sgy[0]);
      sSg[2 * NV + tv] = make_double2(sgy[1], sgy[2]);
      // contravariant components: the divergence below differentiates these along the tensor lines
      sS0[tv] = make_double2(__builtin_fma(gy0, sgy[0], gx0 * sgx[0]), __builtin_fma(gy0, sgy[1], gx0 * sgx[1]));
      sS02[tv] = __builtin_fma(gy0, sgy[2], gx0 * sgx[2]);
      sS1[tv] = make_double2(__builtin_fma(gy1, sgy[0], gx1 * sgx[0]), __builtin_fma(gy1, sgy[1], gx1 * sgx[1]));
      sS12[tv] = __builtin_fma(gy1, sgy[2], gx1 * sgx[2]);
    }
    __syncthreads();
    T2_STAMP(5);

    // ---- volume part of div sigma (dg_div! :590-611 without the lift) -> SG[3][K][Nq] --------------------------------
    {
      d2 p = sS0[rowb];
      double dv0 = dg0[0] * p.x, dv1 = dg0[0] * p.y, dv2 = dg0[0] * sS02[rowb];
#pragma unroll
      for (int j = 1; j < N1; ++j) {
        p = sS0[rowb + j];
        dv0 = __builtin_fma(dg0[j], p.x, dv0); dv1 = __builtin_fma(dg0[j], p.y, dv1);
        dv2 = __builtin_fma(dg0[j], sS02[rowb + j], dv2);
      }
#pragma unroll
      for (int j = 0; j < N1; ++j) {
        p = sS1[colb + N1 * j];
        dv0 = __builtin_fma(dg1[j], p.x, dv0); dv1 = __builtin_fma(dg1[j], p.y, dv1);
        dv2 = __builtin_fma(dg1[j], sS12[colb + N1 * j], dv2);
      }
      if (WALLS && gb) {   // (uniform) the same for the volume part of dg_div! (:604-609)
        double ax[3], bx[3], ay[3], by[3];   // line derivatives of sigma_x, sigma_y at the Gauss nodes: Vq (Dr sigma), Vq (Ds sigma)
        {
          d2 p0 = sSg[rowb], p1 = sSg[NV + rowb], p2 = sSg[2 * NV + rowb];
          ax[0] = dg0[0] * p0.x; ax[1] = dg0[0] * p0.y; ax[2] = dg0[0] * p1.x;
          ay[0] = dg0[0] * p1.y; ay[1] = dg0[0] * p2.x; ay[2] = dg0[0] * p2.y;
          d2 r0 = sSg[colb], r1 = sSg[NV + colb], r2 = sSg[2 * NV + colb];
          bx[0] = dg1[0] * r0.x; bx[1] = dg1[0] * r0.y; bx[2] = dg1[0] * r1.x;
          by[0] = dg1[0] * r1.y; by[1] = dg1[0] * r2.x; by[2] = dg1[0] * r2.y;
#pragma unroll
          for (int j = 1; j < N1; ++j) {
            p0 = sSg[rowb + j]; p1 = sSg[NV + rowb + j]; p2 = sSg[2 * NV + rowb + j];
            ax[0] = __builtin_fma(dg0[j], p0.x, ax[0]); ax[1] = __builtin_fma(dg0[j], p0.y, ax[1]); ax[2] = __builtin_fma(dg0[j], p1.x, ax[2]);
            ay[0] = __builtin_fma(dg0[j], p1.y, ay[0]); ay[1] = __builtin_fma(dg0[j], p2.x, ay[1]); ay[2] = __builtin_fma(dg0[j], p2.y, ay[2]);
            r0 = sSg[colb + N1 * j]; r1 = sSg[NV + colb + N1 * j]; r2 = sSg[2 * NV + colb + N1 * j];
            bx[0] = __builtin_fma(dg1[j], r0.x, bx[0]); bx[1] = __builtin_fma(dg1[j], r0.y, bx[1]); bx[2] = __builtin_fma(dg1[j], r1.x, bx[2]);
            by[0] = __builtin_fma(dg1[j], r1.y, by[0]); by[1] = __builtin_fma(dg1[j], r2.x, by[1]); by[2] = __builtin_fma(dg1[j], r2.y, by[2]);
          }
        }
        __syncthreads();   // every lane is past its reads of the contravariant planes
        d2* sc = reinterpret_cast<d2*>(lds + R0 + 6 * NV);   // 3 + 3 pair planes behind sigma (S^0, S^1, V, half jumps: all dead here)
        d2* sm = sc + 3 * NV;
        const double* ipa = sW + N1 * N1 + a * N1;
        const double* ipb = sW + N1 * N1 + b * N1;
        const double* iqa = sW + a * N1;
        const unsigned colq = ev * Nq + b;
        double o[12];   // derivatives of sigma_x along direction 0 [3], 1 [3], of sigma_y [3], [3] at this lane's NODE
        sc[tv] = make_double2(ax[0], ax[1]); sc[NV + tv] = make_double2(ax[2], bx[0]); sc[2 * NV + tv] = make_double2(bx[1], bx[2]);
        __syncthreads();
        tp_apply<N1, NV, 3>(ipa, ipb, sc, sm, tv, colq, N1, colb, N1, o);
        __syncthreads();
        sc[tv] = make_double2(ay[0], ay[1]); sc[NV + tv] = make_double2(ay[2], by[0]); sc[2 * NV + tv] = make_double2(by[1], by[2]);
        __syncthreads();
        tp_apply<N1, NV, 3>(ipa, ipb, sc, sm, tv, colq, N1, colb, N1, o + 6);
        const double iJn = rcp_refined(wgm[4]);
        double w[3];
#pragma unroll
        for (int c = 0; c < 3; ++c)   // (rxj*uxr + sxj*uxs + ryj*uyr + syj*uys) / J
          w[c] = __builtin_fma(wgm[3], o[9 + c], __builtin_fma(wgm[1], o[6 + c], __builtin_fma(wgm[2], o[3 + c], wgm[0] * o[c]))) * iJn;
        __syncthreads();
        sc[tv] = make_double2(w[0], w[1]); sc[NV + tv] = make_double2(w[2], 0.0);
        __syncthreads();
        double dn[4];
        tp_apply<N1, NV, 2>(iqa, iqa, sc, sm, tv, rowb, 1, colq, N1, dn);
        if (sEb[ev]) {   // times the record's J: the last phase multiplies by its reciprocal before Pq
          dv0 = g[4] * dn[0]; dv1 = g[4] * dn[1]; dv2 = g[4] * dn[2];
        }
      }
      if (ESDG_T2_SIGMA_DEFER_STORES) {
        cdv[0] = dv0; cdv[1] = dv1; cdv[2] = dv2; cva = vact;
      } else if (vact) {
        double* o = SG + ESDG_EW(e0) * Nq + tv;
        o[0] = dv0; o[KN] = dv1; o[2 * KN] = dv2;
      }
    }
    T2_STAMP(6);
    // ---- face lanes: own normal stress (Ef*sigma_x)*nxJ + (Ef*sigma_y)*nyJ -> B ----------------------------------------
    {
      const double* gn = nr;
      const double nx = gn[0], ny = gn[1];
      double ee[N1];
      row_of(ee_r, RTE + fn * N1P, ee);
      d2 p0 = sSg[fnode0], p1 = sSg[NV + fnode0], p2 = sSg[2 * NV + fnode0];
      double fx0 = ee[0] * p0.x, fx1 = ee[0] * p0.y, fx2 = ee[0] * p1.x, fy0 = ee[0] * p1.y, fy1 = ee[0] * p2.x, fy2 = ee[0] * p2.y;
#pragma unroll
      for (int j = 1; j < N1; ++j) {
        const unsigned n = fnode0 + j * fstride;
        p0 = sSg[n]; p1 = sSg[NV + n]; p2 = sSg[2 * NV + n];
        fx0 = __builtin_fma(ee[j], p0.x, fx0); fx1 = __builtin_fma(ee[j], p0.y, fx1); fx2 = __builtin_fma(ee[j], p1.x, fx2);
        fy0 = __builtin_fma(ee[j], p1.y, fy0); fy1 = __builtin_fma(ee[j], p2.x, fy1); fy2 = __builtin_fma(ee[j], p2.y, fy2);
      }
      double sn[3] = {__builtin_fma(fy0, ny, fx0 * nx), __builtin_fma(fy1, ny, fx1 * nx), __builtin_fma(fy2, ny, fx2 * nx)};
      if (WALLS && bcf) {   // minus the prescribed stress jump (see the kernel's header comment)
        const double fx[3] = {fx0, fx1, fx2}, fy[3] = {fy0, fy1, fy2};
        double sj[3];
        wall_stress_jump(sn, fx, fy, bcf, vlid, gn, ph, sj);
        sn[0] = -sj[0]; sn[1] = -sj[1]; sn[2] = -sj[2];
      }
      if (COAL) {   // (duplicate lanes write duplicates; slots of elements beyond the mesh are never stored)
        sBs[tf * B_NC] = sn[0]; sBs[tf * B_NC + 1] = sn[1]; sBs[tf * B_NC + 2] = sn[2];
        cnb = (FULL ? E : nE) * Nfq * B_NC; ce0 = e0;
      } else if (ESDG_T2_SIGMA_DEFER_STORES) {
        csn[0] = sn[0]; csn[1] = sn[1]; csn[2] = sn[2]; cfa = fact; ce0 = e0;
      } else if (fact) {
        double* bb = B + trace_slot<N1>(M, ESDG_EW(e0) + ef, fn) * B_NC;
        bb[0] = sn[0]; bb[1] = sn[1]; bb[2] = sn[2];
      }
    }
    T2_STAMP(7);
    __syncthreads();   // the LDS planes are rewritten by the next iteration
    T2_STAMP(8);
  }
  if (ESDG_T2_SIGMA_DEFER_STORES) {   // the last group's results
    if (cva) { double* o = SG + ESDG_EW(ce0) * Nq + tv; o[0] = cdv[0]; o[KN] = cdv[1]; o[2 * KN] = cdv[2]; }
    if (COAL) {
      double* bb = B + ESDG_EW(ce0) * Nfq * B_NC;
#pragma unroll
      for (int r = 0; r < BPT; ++r) { const int idx = (int)tid + r * G::GT; if (idx < cnb) bb[idx] = sBs[idx]; }
    } else if (cfa) { double* bb = B + trace_slot<N1>(M, ESDG_EW(ce0) + ef, fn) * B_NC; bb[0] = csn[0]; bb[1] = csn[1]; bb[2] = csn[2]; }
  }
  if (vt_partial) {   // (uniform) lanes of a wave in lane order, then the waves in order
    __syncthreads();
    lds[tid] = vt;
    __syncthreads();
    if (tid == 0) {
      double t = 0.0;
      for (int i = 0; i < G::GT; ++i) t += lds[i];
      vt_partial[blockIdx.x] = t;
    }
  }
  T2_STAMP_FLUSH;
}


// ---------------------------------------------------------------------------------------------------------------------
// last phase (meshes without walls): interface + volume flux differencing (+ viscous divergence and penalty) -> rhs
// (euler_quad.jl:141-194 / rhs_inviscid! :447-528, update_flux! :308-324, flux_differencing! :326-348, dg_div! :590-611;
// formulas at the statements below)
//
// Flux pairs.  Volume-volume pairs: circulant rounds, each unordered pair once, by the volume lanes (the partner's share
// goes to an LDS accumulator with ds_add_f64).  Volume-face pairs: by the FACE lanes, which walk the N1 volume nodes
// of their line with their own trace state in registers, keep the face node's sum in registers and push the volume
// node's share with ds_add_f64.  Every accumulator cell receives at most TWO adds, on a cell zeroed beforehand: one plane
// set per direction for the volume-volume shares (rounds i = 0, 1 of a direction), one per direction for the shares
// coming from the two faces at the ends of a line, one for the antipodal round of even N1.  0 + x + y does not depend on
// the order of the two adds, so the result is bit-for-bit the same wherever the element sits in its group and however
// the waves of a group are scheduled (elements may straddle two waves) -- no per-wave accumulator copies.
// ---------------------------------------------------------------------------------------------------------------------
// ---------------------------------------------------------------------------------------------------------------------
// phase 0: entropy projection to the faces -> the (rho, u, v, beta) trace records A_U
// (euler_quad.jl:141-157 / rhs_inviscid! :447-495: VU = v(Vq u), Uf = u(Vf Pq VU))
// A workgroup per group of E elements -- one-shot on small launches, persistent from ESDG_T2_PROJECT_PERSIST groups per resident
// workgroup on (round 5) --, pair planes, per-node table rows; one logarithm per node and a refined rsqrt instead of a logarithm in
// the inverse map (v_of_state_onelog / prim_of_v2_fast, esdg_t2_physics.hpp).  The round-1 kernel kept its LDS
// arrays as [element][component][node] planes of doubles: a wave waited 38 % of its life on LDS instructions, a third of its
// LDS cycles were bank conflicts (profiles/r03u_sq_counters.txt).
// ---------------------------------------------------------------------------------------------------------------------
template <int N1, bool MODAL>
__global__ __launch_bounds__(Geo<N1>::GT) void kt2_project(TensorTables TT, MeshDev M, const double* __restrict__ Q, double* __restrict__ A_U) {
  using G = Geo<N1>;
  constexpr int Nq = G::Nq, Nfq = G::Nfq, E = G::E, NV = G::NV, NF = G::NF;
  constexpr NodeLayout NL(N1);
  constexpr FaceLayout FL(N1);
  __shared__ __align__(16) double lds[8 * NV];   // Vq scratch A | B0 | B1 (pair planes), then the entropy variables (2 pair planes)
  prio_entry_begin();
  d2* sA = reinterpret_cast<d2*>(lds);
  d2* sB0 = reinterpret_cast<d2*>(lds + 4 * NV);
  d2* sB1 = reinterpret_cast<d2*>(lds + 6 * NV);
  const unsigned tid = threadIdx.x;
  const unsigned tv = tid < (unsigned)NV ? tid : tid - NV;              // (lanes beyond the slots duplicate a slot, see kt2_sigma)
  const unsigned tf = tid < (unsigned)NF ? tid : tid % NF;
  const unsigned ev = tv / Nq, q = tv - ev * Nq, a = q % N1, b = q / N1;
  const unsigned ef = tf / Nfq, fn = tf - ef * Nfq;
  const unsigned rowb = ev * Nq + N1 * b, colq = ev * Nq + b;
  const int64_t KN = M.K * Nq;
  // Element groups blockIdx.x, blockIdx.x + gridDim.x, ...: one group per workgroup when the launch brings a workgroup per group
  // (the one-shot form), several with the persistent grid of ESDG_T2_PROJECT_PERSIST (A/B hook): the per-node table rows are then
  // fetched once per workgroup and the next group's state is requested into the registers of the current one after their last use.
  const int64_t ngroups = (M.e_count + E - 1) / E;
  int64_t grp = gridDim.x == (unsigned)ngroups ? xcd_group(blockIdx.x, gridDim.x) : (int64_t)blockIdx.x;
  // every global load at entry, unconditionally (state; per-node / per-face-node table rows)
  double x[4], cq[N1], ee[N1];
  {
    const int64_t e0 = M.e_begin + grp * E;
    const int nE = (int)min((int64_t)E, M.e_begin + M.e_count - e0);
    const unsigned tvl = tv < (unsigned)(nE * Nq) ? tv : 0u;
#pragma unroll
    for (int f = 0; f < 4; ++f) x[f] = Q[f * KN + ESDG_EW(e0) * Nq + tvl];
  }
  const double* nd_ = TT.node_d + q;
  const double* fd_ = TT.face_d + fn;
  const int* fi_ = TT.face_i + fn;
#pragma unroll
  for (int i = 0; i < N1; ++i) { cq[i] = MODAL ? nd_[(NL.IQ + i) * Nq] : 0.0; ee[i] = fd_[(FL.EE + i) * Nfq]; }
  const unsigned fnode0 = ef * Nq + fi_[(FL.NODE0) * Nfq], fstride = fi_[(FL.STRIDE) * Nfq];
  prio_entry_end();

#pragma unroll 1
  for (; grp < ngroups; grp += gridDim.x) {
    const int64_t e0 = M.e_begin + grp * E;
    const int nE = (int)min((int64_t)E, M.e_begin + M.e_count - e0);
    const bool fact = tid < (unsigned)(nE * Nfq);
    double U[4];
    if (MODAL) {
      vq_apply<N1, NV>(cq, sA, sB0, sB1, tv, rowb, colq, x, U);
    } else {
#pragma unroll
      for (int f = 0; f < 4; ++f) U[f] = x[f];
    }
    if (grp + gridDim.x < ngroups) {   // (uniform) the next group's state into the registers just consumed
      const int64_t e1 = M.e_begin + (grp + gridDim.x) * E;
      const int nE1 = (int)min((int64_t)E, M.e_begin + M.e_count - e1);
      const unsigned tv1 = tv < (unsigned)(nE1 * Nq) ? tv : 0u;
#pragma unroll
      for (int f = 0; f < 4; ++f) x[f] = Q[f * KN + ESDG_EW(e1) * Nq + tv1];
    }
    if (MODAL) __syncthreads();   // every lane is past its reads of the scratch planes, which take the entropy variables below
    double V[4];
    v_of_state_onelog<MODAL>(U, V);
    sA[tv] = make_double2(V[0], V[1]);
    sA[NV + tv] = make_double2(V[2], V[3]);
    __syncthreads();
    prio_exit();
    // face lanes: Vf = Ef * V along the node's line, then the primitive state of u(Vf)
    d2 p0 = sA[fnode0], p1 = sA[NV + fnode0];
    double Vf[4] = {ee[0] * p0.x, ee[0] * p0.y, ee[0] * p1.x, ee[0] * p1.y};
#pragma unroll
    for (int j = 1; j < N1; ++j) {
      p0 = sA[fnode0 + j * fstride]; p1 = sA[NV + fnode0 + j * fstride];
      Vf[0] = __builtin_fma(ee[j], p0.x, Vf[0]); Vf[1] = __builtin_fma(ee[j], p0.y, Vf[1]);
      Vf[2] = __builtin_fma(ee[j], p1.x, Vf[2]); Vf[3] = __builtin_fma(ee[j], p1.y, Vf[3]);
    }
    double qf[4];
    prim_of_v2_fast<MODAL>(Vf, qf);
    if (fact) {
      d2* rec = reinterpret_cast<d2*>(A_U + trace_slot<N1>(M, ESDG_EW(e0) + ef, fn) * FAU_NC);
      rec[0] = make_double2(qf[0], qf[1]);
      rec[1] = make_double2(qf[2], qf[3]);
    }
    __syncthreads();   // (the planes are rewritten by the next group)
  }
}

// kt2_rhs: partner records read one flux round ahead (A/B hook, off).  Measured in round 3 (profiles/experiments/
// r03_rhs_prefetch_ab.log): N = 4 nothing (0.392-0.397 vs 0.396-0.398 ms), N = 2, 3 -1.5 % of the kernel, N = 6 +15 % (its 12
// registers push the N1 = 7 instantiation from 166 to 178 VGPRs = from three waves per SIMD to two; N1 = 6 would spill).
#ifndef ESDG_T2_ACC_REUSE
#define ESDG_T2_ACC_REUSE 3   // bit 0: collocated Euler at N1 = 5; bit 1: every instantiation at N1 = 4
#endif
#ifndef ESDG_T2_PREFETCH_REC
#define ESDG_T2_PREFETCH_REC 0
#endif
template <int N1> struct RhsPrefetch { static constexpr bool ON = ESDG_T2_PREFETCH_REC && N1 <= 5; };
template <int N1, bool MODAL, bool VISC> struct RhsLds2 {
  using G = GeoR<N1>;
  static constexpr NodeLayout NL = NodeLayout(N1);
  static constexpr int NV = G::NV, NF = G::NF;
  static constexpr int NVV = (NL.NFULL + 1) / 2;                      // accumulator plane sets per direction, volume-volume
  // REUSE (the collocated Euler instantiation at N1 = 5, ESDG_T2_ACC_REUSE): the volume-face shares go to the plane sets of the volume-volume
  // shares, which every lane has gathered and zeroed again between two barriers -- the same sequence of additions per node, so
  // the same bits, two sets instead of four: 26.2 -> 18.2 KB of LDS = eight workgroups per CU, and with the collocated kernel's
  // 122 VGPRs four waves per SIMD.  (The CNS instantiation needs ~150 VGPRs: three waves either way; measured there in round 2.)
  static constexpr bool REUSE = ESDG_T2_ACC_REUSE && ((N1 == 5 && !VISC && !MODAL) || (N1 == 4 && (ESDG_T2_ACC_REUSE & 2)));
  static constexpr int WPE = N1 <= 6 ? ((REUSE && N1 == 5) ? 4 : 3) : 2;   // waves per SIMD asked of the register allocator
  static constexpr int NACCV = 2 * NVV + (N1 % 2 == 0 ? 1 : 0);       // volume-volume sets (+ antipodal)
  static constexpr int NACC = REUSE ? (NACCV > 2 ? NACCV : 2) : NACCV + 2;   // + the two volume-face sets
  static constexpr int FSET = REUSE ? 0 : NACCV;                      // first volume-face set
  static constexpr int REC = 0;                                       // 3 pair planes [NV]: (rho,u) (v,beta) (lrho,lbeta)
  // Before they are zeroed the accumulator planes hold Vq's second buffer (2 pair planes) and the 1D operator IQ (rows
  // padded to even length); after the flux rounds the (dead) third record plane holds IP.
  static constexpr int SGF = REC + 6 * NV;                            // face totals G_f: 2 pair planes [NF]
  static constexpr int ACC = SGF + 4 * NF;                            // NACC x 2 pair planes [NV]
  static constexpr int TABQ = ACC + 4 * NV, TABP = REC + 4 * NV;
  static constexpr int GEO = ACC + NACC * 4 * NV;
  // (geometry staged by every lane, padded to whole rounds of the group: an `if (n < NGEO)` around the LDS write lets
  // hipcc sink the global load into the branch and wait there with vmcnt(0) -- for every load in flight, traces included)
  static constexpr int NLDS = GEO + ((G::E * GEO_STRIDE + G::GT - 1) / G::GT) * G::GT;
  static_assert(N1 * (N1 + (N1 & 1)) <= 2 * NV, "operator fits the third record plane");
};

// One workgroup per group of E elements (one-shot).  Measured and rejected at cfg3 (profiles/experiments/README.md):
//   * persistent over the groups with next-group prefetch: rows + loop-carried loads = 314 VGPRs, 2 workgroups per CU, 0.504 ms
//     (capped at 256 VGPRs: spills, 0.727 ms);
//   * persistent without prefetch (rows fetched once per workgroup, in registers or in LDS): 0.46 ms -- hoisted addresses
//     spill, and resident workgroups that start together stay in step, so their load and compute phases do not overlap
//     the way consecutive one-shot workgroups' do.
template <int N1, bool MODAL, bool VISC, bool WALLS>
__global__ __launch_bounds__(GeoR<N1>::GT, (RhsLds2<N1, MODAL, VISC>::WPE)) void kt2_rhs(TensorTables TT, MeshDev M, Phys ph, const double* __restrict__ Q,
                                                                       const double* __restrict__ A_U, const double* __restrict__ SG,
                                                                       const double* __restrict__ B, double* __restrict__ rhs, LsrkFuse lf) {
  using G = GeoR<N1>;
  using LD = RhsLds2<N1, MODAL, VISC>;
  constexpr int Nq = G::Nq, Nfq = G::Nfq, E = G::E, NV = G::NV, NF = G::NF;
  constexpr NodeLayout NL(N1);
  constexpr FaceLayout FL(N1);
  constexpr int NFULL = NL.NFULL, NRND = NL.NRND, NVV = LD::NVV;
  constexpr bool PFR = RhsPrefetch<N1>::ON;
  constexpr int NGEO = E * GEO_STRIDE, GPT = (NGEO + G::GT - 1) / G::GT;
  // 1D operators IQ (Gauss nodes from nodal values) and IP (back), rows padded to an even length: a lane reads row a / b
  // as N1P / 2 ds_read_b128 at the point of use instead of holding per-lane copies in registers across the flux rounds.
  // Each is staged (from one register per lane, loaded with the group's data) into the third record plane while that
  // plane holds no records.
  constexpr int N1P = N1 + (N1 & 1);
  constexpr TensorLayout TL(N1);
  static_assert(TL.IP == TL.IQ + N1 * N1, "IQ and IP are adjacent in the 1D tables");
  __shared__ __align__(16) double lds[LD::NLDS];
  __shared__ int sEbR[WALLS ? GeoR<N1>::E : 1];   // WALLS: element has a boundary node (set by its face lanes; see the end of the kernel)
  if (WALLS && threadIdx.x < (unsigned)GeoR<N1>::E) sEbR[threadIdx.x] = 0;
  d2* sRec = reinterpret_cast<d2*>(lds + LD::REC);
  d2* sAcc = reinterpret_cast<d2*>(lds + LD::ACC);
  d2* sGf = reinterpret_cast<d2*>(lds + LD::SGF);
  double* sGeo = lds + LD::GEO;

  const unsigned tid = threadIdx.x;
  const unsigned tv = tid < (unsigned)NV ? tid : tid - NV;
  const unsigned tf = tid < (unsigned)NF ? tid : tid % NF;
  const unsigned ev = tv / Nq, q = tv - ev * Nq, a = q % N1, b = q / N1;
  const unsigned ef = tf / Nfq, fn = tf - ef * Nfq;
  const unsigned rowb = ev * Nq + N1 * b, colb = ev * Nq + a, colq = ev * Nq + b;
  const int64_t KN = M.K * Nq;
  const bool inviscid = (ph.parts & 1) != 0, viscous = VISC && (ph.parts & 2) != 0;
  const bool vown = NV == G::GT || tid < (unsigned)NV, fown = NF == G::GT || tid < (unsigned)NF;   // not a duplicate lane

  const int64_t e0r = M.e_begin + xcd_group(blockIdx.x, gridDim.x) * E;
  const int nE = (int)min((int64_t)E, M.e_begin + M.e_count - e0r);
  const int64_t e0 = ESDG_EW(e0r);
  const bool vact = tid < (unsigned)(nE * Nq);
  constexpr RhsRows RR(N1);
  prio_entry_begin();
  T2_STAMP_INIT;
#ifdef ESDG_T2_POISON   // diagnostic build: LDS starts as NaN, so a read of a slot nobody wrote shows in the result
  for (int i = tid; i < LD::NLDS; i += G::GT) lds[i] = __builtin_nan("");
  __syncthreads();
#endif

  // ---- global loads -------------------------------------------------------------------------------------------------
  // (the neighbour index first: the trace loads depend on it)
  double x[4], geo[GPT], qM[8], qP[8], bPn[3] = {0, 0, 0}, bOwn[3] = {0, 0, 0}, dvs[3] = {0, 0, 0};
  const unsigned tvl = tv < (unsigned)(nE * Nq) ? tv : 0u, tfl = tf < (unsigned)(nE * Nfq) ? tf : 0u, tvg = tv;
  const int64_t nf = e0 * Nfq + tfl;
  const int64_t nfs = trace_slot<N1>(M, e0 + tfl / Nfq, tfl % Nfq);   // this face node's record in the trace buffers (MeshDev::bf)
  const unsigned mp = ESDG_EWN((unsigned)M.mapP[nf], Nfq);
  // (Measured and removed in round 3 -- a structured-neighbour guess of mp from kernel arguments, the neighbour-trace loads
  // issued from it at entry and mp verified at first use: 4 % slower.  The traces are not needed until after the volume-volume
  // rounds; one round trip earlier they only queue ahead of the next workgroups' state loads.  profiles/experiments/README.md)
  int bcf = 0;
  double vlid = 1.0;
  if (WALLS) {   // boundary flag (1 wall, 2 lid, 3 inflow, 4 copy) and lid velocity of this lane's face node
    bcf = M.bc[nf];
    if (M.vlid) vlid = M.vlid[nf];
  }
  // this lane's face-node normal as the driver holds it = face mean of the record + float difference (MeshDev::fnd / fsd)
  const float2 nd = reinterpret_cast<const float2*>(M.fnd)[nf];
  const float sd = WALLS ? M.fsd[nf] : 0.f;
#pragma unroll
  for (int f = 0; f < 4; ++f) x[f] = Q[f * KN + e0 * Nq + tvl];
#pragma unroll
  for (int i = 0; i < GPT; ++i) { const unsigned n = tid + i * G::GT; geo[i] = M.geo[e0 * GEO_STRIDE + (n < (unsigned)(nE * GEO_STRIDE) ? n : 0u)]; }
  static_assert(N1 * N1 <= G::GT, "one operator entry per lane");
  double tabq = 0.0, tabp = 0.0;
  // (every lane loads and later stages an entry, lanes beyond the N1*N1 entries a duplicate: a store under `if (tid < N1*N1)`
  // lets hipcc sink the load into the branch and wait there with vmcnt(0), i.e. for every load in flight -- and the other
  // wave of the group then waits at the barrier behind it)
  const unsigned tabn = tid % (unsigned)(N1 * N1), tabo = (tabn / N1) * N1P + tabn % N1;
  if (MODAL) { tabq = TT.dbl[TL.IQ + tabn]; tabp = TT.dbl[TL.IP + tabn]; }
  // packed rows (RhsRows): NPV + NPF coalesced 16-byte loads and one int4 + one int per lane
  double svv[NRND > 0 ? NRND : 1], pw[4], svf[N1], pd, wfac;
  unsigned pid[NRND > 0 ? NRND : 1], fq[4], fnode0, fstride;
  int fdir, ad;
  {
    const d2* vd = reinterpret_cast<const d2*>(TT.rhs_vd) + q;
    const d2* fd = reinterpret_cast<const d2*>(TT.rhs_fd) + fn;
    double rv[2 * RR.NPV], rf[2 * RR.NPF];
#pragma unroll
    for (int i = 0; i < RR.NPV; ++i) { const d2 t = vd[i * Nq]; rv[2 * i] = t.x; rv[2 * i + 1] = t.y; }
#pragma unroll
    for (int i = 0; i < RR.NPF; ++i) { const d2 t = fd[i * Nfq]; rf[2 * i] = t.x; rf[2 * i + 1] = t.y; }
    const int4 wi = reinterpret_cast<const int4*>(TT.rhs_vi)[q];
    const unsigned wf = (unsigned)TT.rhs_fi[fn];
#pragma unroll
    for (int r = 0; r < NRND; ++r) {
      svv[r] = rv[r];
      pid[r] = ev * Nq + (((unsigned)(r < 4 ? wi.x : wi.y) >> (8 * (r % 4))) & 255u);
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) { pw[k] = rv[NRND + k]; fq[k] = ev * Nfq + (((unsigned)wi.z >> (8 * k)) & 255u); }
    pd = rv[NRND + 4];
#pragma unroll
    for (int j = 0; j < N1; ++j) svf[j] = rf[j];
    wfac = rf[N1];
    fnode0 = ef * Nq + (wf & 255u); fstride = (wf >> 8) & 255u;
    fdir = (int)(wf >> 17); ad = (N1 % 2 == 0) ? wi.w : 0;
  }
  const int opf = fdir ? TT.op1 : TT.op0;
  const unsigned gfo = 5 + 3 * (fn / N1);

    T2_STAMP(9);     // loads that need no neighbour index issued
#ifdef ESDG_T2_STAMP
    { unsigned mpw = mp; asm volatile("" : "+v"(mpw)); (void)mpw; }
    T2_STAMP(10);    // neighbour index landed
#endif
#ifdef ESDG_EXP_SMALLTRACE   // experiment: all trace reads from a cache-resident window (wrong results, same instruction stream)
    const int64_t nfx = nf & ESDG_EXP_SMALLTRACE; const size_t mpx = mp & ESDG_EXP_SMALLTRACE;
    const d2* aM = reinterpret_cast<const d2*>(A_U + nfx * FAU_NC);
    const d2* aP = reinterpret_cast<const d2*>(A_U + mpx * FAU_NC);
#else
    const d2* aM = reinterpret_cast<const d2*>(A_U + nfs * FAU_NC);
    const d2* aP = reinterpret_cast<const d2*>(A_U + (size_t)mp * FAU_NC);
#endif
#pragma unroll
    for (int c = 0; c < 2; ++c) {   // (rho, u, v, beta) of both sides; their logs, energy and wavespeed are rebuilt at first use
      const d2 m = aM[c], p = aP[c];
      qM[2 * c] = m.x; qM[2 * c + 1] = m.y; qP[2 * c] = p.x; qP[2 * c + 1] = p.y;
    }
  prio_entry_end();

  {
    // ---- state and geometry to LDS ----------------------------------------------------------------------------------------
    double U[4];
    T2_STAMP(0);     // table rows + load issue
#pragma unroll
    for (int i = 0; i < GPT; ++i) sGeo[tid + i * G::GT] = geo[i];
    if (MODAL) {
      lds[LD::TABQ + tabo] = tabq;
      sRec[tv] = make_double2(x[0], x[1]);
      sRec[NV + tv] = make_double2(x[2], x[3]);
    } else {
#pragma unroll
      for (int f = 0; f < 4; ++f) U[f] = x[f];
    }

    // ---- state at the Gauss node -> primitives + logs ---------------------------------------------------------------------
    if (MODAL) {
      __syncthreads();   // operator rows staged
      double cq[N1P];
      const d2* row = reinterpret_cast<const d2*>(lds + LD::TABQ + a * N1P);
#pragma unroll
      for (int i = 0; i < N1P / 2; ++i) { const d2 t = row[i]; cq[2 * i] = t.x; cq[2 * i + 1] = t.y; }
      vq_apply<N1, NV>(cq, sRec, sAcc, sAcc + NV, tv, rowb, colq, nullptr, U);
      __syncthreads();   // every lane is past its reads of the buffer in the accumulator planes, which are zeroed below
    }
    T2_STAMP(1);     // Q landed + Vq
    double qh[6];
    prim_logs<MODAL>(U, qh);
    // (modal: the Vq input planes alias the record planes; every lane is past its stage-1 reads, which the second barrier
    // inside vq_apply separates from here)
    sRec[tv] = make_double2(qh[0], qh[1]);
    sRec[NV + tv] = make_double2(qh[2], qh[3]);
    sRec[2 * NV + tv] = make_double2(qh[4], qh[5]);
#pragma unroll
    for (int p = 0; p < 2 * LD::NACC; ++p) sAcc[p * NV + tv] = make_double2(0.0, 0.0);
    __syncthreads();     // geometry, records and zeroed accumulators of every lane are in place
    T2_STAMP(2);     // prim_logs + records

    // ---- flux differencing ----------------------------------------------------------------------------------------------
    // (volume-volume rounds first: they need no trace data, whose loads were issued last)
    double acc[4] = {0, 0, 0, 0};
    if (inviscid) {   // uniform
      const double* g = sGeo + ev * GEO_STRIDE;
      const double gx0 = 2 * g[TT.op0], gy0 = 2 * g[2 + TT.op0], gx1 = 2 * g[TT.op1], gy1 = 2 * g[2 + TT.op1];
      // volume-volume rounds: pair (pos, pos + i + 1 mod N1) of direction d; share of the partner -> plane set d * NVV + i / 2
      // (ESDG_T2_PREFETCH_REC: the partner record of round r + 1 is read before round r's flux is evaluated -- the wave-uniform
      // variants of ec_flux_dir are branches, across which hipcc does not move the next round's ds_reads up by itself, so every
      // round started with an exposed LDS round trip)
      d2 nx0, nx1, nx2;
      if (PFR && NFULL > 0) { nx0 = sRec[pid[0]]; nx1 = sRec[NV + pid[0]]; nx2 = sRec[2 * NV + pid[0]]; }
#pragma unroll
      for (int r = 0; r < 2 * NFULL; ++r) {
        constexpr int NFD = NFULL > 0 ? NFULL : 1;     // (N1 = 2 has no full round: the loop is empty)
        const int d = r / NFD, i = r % NFD;
        d2 p0, p1, p2;
        if (PFR) {
          p0 = nx0; p1 = nx1; p2 = nx2;
          if (r + 1 < 2 * NFULL + (N1 % 2 == 0 ? 1 : 0)) { nx0 = sRec[pid[r + 1]]; nx1 = sRec[NV + pid[r + 1]]; nx2 = sRec[2 * NV + pid[r + 1]]; }
        } else { p0 = sRec[pid[r]]; p1 = sRec[NV + pid[r]]; p2 = sRec[2 * NV + pid[r]]; }
        const double qj[6] = {p0.x, p0.y, p1.x, p1.y, p2.x, p2.y};
        double Fd[4];
        ec_flux_dir<MODAL>(qh, qj, svv[r] * (d ? gx1 : gx0), svv[r] * (d ? gy1 : gy0), Fd);
        double* tgt = reinterpret_cast<double*>(sAcc + (d * NVV + i / 2) * 2 * NV + pid[r]);
#pragma unroll
        for (int c = 0; c < 4; ++c) acc[c] += Fd[c];
        if (vown) {   // duplicate lanes (slots beyond NV) must not add twice
#pragma unroll
          for (int c = 0; c < 4; ++c) lds_add(tgt + (c >> 1) * 2 * NV + (c & 1), -Fd[c]);
        }
      }
      if (N1 % 2 == 0) {   // antipodal pairs: a node serves direction `ad` (one endpoint of every such pair does)
        constexpr int r = 2 * NFULL;
        d2 p0, p1, p2;
        if (PFR && NFULL > 0) { p0 = nx0; p1 = nx1; p2 = nx2; }
        else { p0 = sRec[pid[r]]; p1 = sRec[NV + pid[r]]; p2 = sRec[2 * NV + pid[r]]; }
        const double qj[6] = {p0.x, p0.y, p1.x, p1.y, p2.x, p2.y};
        double Fd[4];
        ec_flux_dir<MODAL>(qh, qj, svv[r] * (ad ? gx1 : gx0), svv[r] * (ad ? gy1 : gy0), Fd);
        double* tgt = reinterpret_cast<double*>(sAcc + (2 * NVV) * 2 * NV + pid[r]);
#pragma unroll
        for (int c = 0; c < 4; ++c) acc[c] += Fd[c];
        if (vown) {
#pragma unroll
          for (int c = 0; c < 4; ++c) lds_add(tgt + (c >> 1) * 2 * NV + (c & 1), -Fd[c]);
        }
      }
    }
    if (LD::REUSE) {   // gather the volume-volume shares and hand the zeroed planes to the volume-face shares
      __syncthreads();
      if (vown) {   // (a duplicate lane would zero its owner's cells before the owner has read them; its own sums are never stored)
#pragma unroll
        for (int p = 0; p < LD::NACCV; ++p) {
          const d2 s0 = sAcc[(2 * p) * NV + tv], s1 = sAcc[(2 * p + 1) * NV + tv];
          acc[0] += s0.x; acc[1] += s0.y; acc[2] += s1.x; acc[3] += s1.y;
          sAcc[(2 * p) * NV + tv] = make_double2(0.0, 0.0); sAcc[(2 * p + 1) * NV + tv] = make_double2(0.0, 0.0);
        }
      }
      __syncthreads();
    }
    T2_STAMP(4);   // volume-volume rounds
    if (VISC) {   // needed after the volume-face pairs: issued here, their destinations are not live during the rounds above
#pragma unroll
      for (int c = 0; c < 3; ++c) { bPn[c] = B[(size_t)mp * B_NC + c]; bOwn[c] = B[nfs * B_NC + c]; dvs[c] = SG[c * KN + e0 * Nq + tvl]; }
    }
    // ---- face lanes: interface flux and penalty from the two trace states (registers only) -----------------------------------
    const double* gf = sGeo + ef * GEO_STRIDE;     // (slots of elements beyond the mesh hold the clamped loads: finite, unused)
    double Gf[4], pnr[3] = {0, 0, 0}, gpen[3] = {0, 0, 0};   // (gpen: WALLS, the penalty's share of the face total)
    {
      const double* gm = gf + gfo;
      // (sJ: the face mean unless a wall closure turns it into a unit normal -- it only scales the LF term, a small jump)
      const double gn[3] = {gm[0] + (double)nd.x, gm[1] + (double)nd.y, WALLS ? gm[2] + (double)sd : gm[2]};
      {   // logs, energy and wavespeed of the two trace states, with the face means of the record as phase 0 used to take them
        const double isJm = rcp_refined(gm[2]);
        trace_rest(qM, gm[0], gm[1], isJm, Gas2<MODAL>::GM1);
        trace_rest(qP, gm[0], gm[1], isJm, Gas2<MODAL>::GM1);
      }
      if (VISC) {   // penalty tau*[[v]] (:817-837): the projected entropy variables are those OF the trace states
        const double bM = 2 * Gas2<MODAL>::GM1 * qM[3], bP = 2 * Gas2<MODAL>::GM1 * qP[3];
        const double tau = ph.viscous_dissp ? -rcp_refined(-bM) * ph.inv_Re : 0.0;
        if (WALLS && bcf) sEbR[ef] = 1;
        if (WALLS && bcf) {   // exterior values by the wall closure; third component overridden as in :827-837
          const double vf[3] = {bM * qM[1], bM * qM[2], -bM};
          double vP[3];
          wall_exterior_v(vf, bcf, vlid, gn, ph, vP);
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
      if (WALLS && bcf >= 3) {   // shock-tube closures (dg2D_CNS_modalESDG.jl:168-185): Dirichlet state / copy, lam = lamP = 0
#pragma unroll
        for (int c = 0; c < 6; ++c) qP[c] = bcf == 3 ? ph.inflow_q[c] : qM[c];
        qM[6] = 0.0; qP[6] = 0.0;
      } else if (WALLS && bcf) {   // wall: mirror state rho+ = rho, beta+ = beta, u+ = u - 2 (u.n) n  (impose_BCs_inviscid! :157-176)
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
      // (the LF jump uses Uf[mapP] - Uf, which vanishes at boundary nodes: mapP = self, cavity :511-513)
      const double dz = (WALLS && bcf) ? 0.0 : 1.0;
      const double dU[4] = {dz * (qP[0] - qM[0]), dz * (qP[0] * qP[1] - qM[0] * qM[1]), dz * (qP[0] * qP[2] - qM[0] * qM[2]), dz * (qP[7] - qM[7])};
      const double wf = inviscid ? wfac : 0.0;
#pragma unroll
      for (int c = 0; c < 4; ++c) Gf[c] = wf * (Fn[c] - LFc * dU[c]);
    }

    T2_STAMP(3);     // traces landed + interface flux
    if (inviscid) {
      // volume-face pairs by the face lanes: share of the volume node -> plane set of the face's direction
      {
        const double gxf = 2 * gf[opf], gyf = 2 * gf[2 + opf];
        const d2* accf = sAcc + (LD::FSET + fdir) * 2 * NV;
        d2 nx0, nx1, nx2;
        if (PFR) { nx0 = sRec[fnode0]; nx1 = sRec[NV + fnode0]; nx2 = sRec[2 * NV + fnode0]; }
#pragma unroll
        for (int j = 0; j < N1; ++j) {
          const unsigned n = fnode0 + j * fstride;
          d2 p0, p1, p2;
          if (PFR) {
            p0 = nx0; p1 = nx1; p2 = nx2;
            if (j + 1 < N1) { const unsigned nn = n + fstride; nx0 = sRec[nn]; nx1 = sRec[NV + nn]; nx2 = sRec[2 * NV + nn]; }
          } else { p0 = sRec[n]; p1 = sRec[NV + n]; p2 = sRec[2 * NV + n]; }
          const double qj[6] = {p0.x, p0.y, p1.x, p1.y, p2.x, p2.y};
          double vv[4];
          ec_flux_dir<MODAL>(qj, qM, svf[j] * gxf, svf[j] * gyf, vv);
          double* tgt = const_cast<double*>(reinterpret_cast<const double*>(accf + n));
          if (fown) {   // duplicate face lanes must not add twice
#pragma unroll
            for (int c = 0; c < 4; ++c) lds_add(tgt + (c >> 1) * 2 * NV + (c & 1), vv[c]);
          }
#pragma unroll
          for (int c = 0; c < 4; ++c) Gf[c] -= vv[c];
        }
      }
    }
    T2_STAMP(5);     // volume-face pairs
    if (VISC) {   // stress jump .5*((sxP-sxf)*nxJ + (syP-syf)*nyJ) + J * penalty (the penalty is lifted WITHOUT 1/J, quirk Q3).
      // Its lift uses the weights of the inviscid lift times the face weight (LW = PW * WFAC) and enters the rhs with the
      // opposite sign, so it rides in the face totals: one lift serves both, no second set of face planes in LDS.
      const double Jf = gf[4], ws = viscous ? wfac : 0.0;
#pragma unroll
      for (int c = 0; c < 3; ++c) Gf[c + 1] = __builtin_fma(-ws, __builtin_fma(Jf, pnr[c], .5 * (-bPn[c] - bOwn[c])), Gf[c + 1]);
      if (WALLS) {
#pragma unroll
        for (int c = 0; c < 3; ++c) gpen[c] = ws * Jf * pnr[c];
      }
    }
    sGf[tf] = make_double2(Gf[0], Gf[1]);
    sGf[NF + tf] = make_double2(Gf[2], Gf[3]);
    // Meshes with walls: the reference divides the nodal coefficients of everything but the penalty by J[i,e] NODE BY NODE
    // (rhs_inviscid! :518, dg_div! :609); the kernel divides by the record's J at the Gauss nodes.  In the elements with a boundary
    // node the lifted stress jump of the energy row is large and that difference shows (like the gradient's, see kt2_sigma); there
    // the result is corrected after Pq: out_i (1 + g_i) - g_i (Pq X)_i, g_i = J/J[i,e] - 1, X = the part that must not be rescaled
    // (penalty, lifted without 1/J; volume divergence, which kt2_sigma already divided by J[i,e]).
    bool gb = false;
    if (WALLS && VISC && MODAL && M.wgeo) gb = __syncthreads_or(bcf != 0) != 0;
    else __syncthreads();
    prio_exit();
    double Jn = 1.0;
    if (WALLS && gb) Jn = M.wgeo[((e0 + (ev < (unsigned)nE ? ev : 0u)) * 5 + 4) * Nq + q];

    // ---- collocated rhs: -(Ph*QF + Lf*flux)/J  (+ viscous divergence and penalty) ---------------------------------------------
    double R[4], RX[4] = {0, 0, 0, 0}, gJ = 1.0;
    {
      const double* g = sGeo + ev * GEO_STRIDE;
      const double iJ = rcp_refined(g[4]);
#pragma unroll
      for (int p = LD::REUSE ? LD::FSET : 0; p < (LD::REUSE ? LD::FSET + 2 : LD::NACC); ++p) {
        const d2 s0 = sAcc[(2 * p) * NV + tv], s1 = sAcc[(2 * p + 1) * NV + tv];
        acc[0] += s0.x; acc[1] += s0.y; acc[2] += s1.x; acc[3] += s1.y;
      }
      double r[4];
#pragma unroll
      for (int c = 0; c < 4; ++c) r[c] = pd * acc[c];
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const d2 g0 = sGf[fq[k]], g1 = sGf[NF + fq[k]];
        r[0] = __builtin_fma(pw[k], g0.x, r[0]); r[1] = __builtin_fma(pw[k], g0.y, r[1]);
        r[2] = __builtin_fma(pw[k], g1.x, r[2]); r[3] = __builtin_fma(pw[k], g1.y, r[3]);
      }
#pragma unroll
      for (int c = 0; c < 4; ++c) R[c] = -r[c] * iJ;
      if (VISC) {   // dg_div! :590-611: volume part from phase 1 (the lift of the stress jumps came with the face totals)
        const double vs = viscous ? iJ : 0.0;
#pragma unroll
        for (int c = 0; c < 3; ++c) R[c + 1] = __builtin_fma(dvs[c], vs, R[c + 1]);
        if (WALLS && gb) {   // (uniform) X = lift of the penalty + volume divergence, at the Gauss nodes
          __syncthreads();   // every lane is past its gather of the accumulator planes, whose space takes the penalty's face values
          d2* sP = sAcc + 2 * NV;   // (behind the two pair planes Pq uses below)
          sP[tf] = make_double2(gpen[0], gpen[1]);
          sP[NF + tf] = make_double2(gpen[2], 0.0);
          __syncthreads();
          double lp[3] = {0, 0, 0};
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            const d2 g0 = sP[fq[k]], g1 = sP[NF + fq[k]];
            lp[0] = __builtin_fma(pw[k], g0.x, lp[0]); lp[1] = __builtin_fma(pw[k], g0.y, lp[1]); lp[2] = __builtin_fma(pw[k], g1.x, lp[2]);
          }
#pragma unroll
          for (int c = 0; c < 3; ++c) RX[c + 1] = __builtin_fma(dvs[c], vs, lp[c] * iJ);
          gJ = g[4];
        }
      }
    }
    T2_STAMP(6);     // barrier + gather + lift
    // ---- out = Pq R (modal), store or fused low-storage RK stage ---------------------------------------------------------------
    double out[4];
    if (MODAL) {
      d2* sA = sRec;                       // records are dead (all lanes are past the barrier after the flux rounds)
      d2* sB = sAcc;                       // accumulators: read above, rewritten only after the next barrier
      lds[LD::TABP + tabo] = tabp;   // IP into the (dead) third record plane
      auto pq_apply = [&](const double* Rin, double* o) {
        sA[tv] = make_double2(Rin[0], Rin[1]);
        sA[NV + tv] = make_double2(Rin[2], Rin[3]);
        __syncthreads();
        double ipl[N1P], iph[N1P];
        {
          const d2* rl = reinterpret_cast<const d2*>(lds + LD::TABP + a * N1P);
          const d2* rh = reinterpret_cast<const d2*>(lds + LD::TABP + b * N1P);
#pragma unroll
          for (int i = 0; i < N1P / 2; ++i) { const d2 t = rl[i], u = rh[i]; ipl[2 * i] = t.x; ipl[2 * i + 1] = t.y; iph[2 * i] = u.x; iph[2 * i + 1] = u.y; }
        }
        {   // stage 1: W[a + N1 b] = sum_j IP[a,j] R[b + N1 j]   (this lane: column b of R)
          const d2* rr = sA + colq;
          d2 p = rr[0], t = rr[NV];
          double w0 = ipl[0] * p.x, w1 = ipl[0] * p.y, w2 = ipl[0] * t.x, w3 = ipl[0] * t.y;
#pragma unroll
          for (int j = 1; j < N1; ++j) {
            p = rr[N1 * j]; t = rr[NV + N1 * j];
            w0 = __builtin_fma(ipl[j], p.x, w0); w1 = __builtin_fma(ipl[j], p.y, w1);
            w2 = __builtin_fma(ipl[j], t.x, w2); w3 = __builtin_fma(ipl[j], t.y, w3);
          }
          sB[tv] = make_double2(w0, w1);
          sB[NV + tv] = make_double2(w2, w3);
        }
        __syncthreads();
        {   // stage 2: out[a + N1 b] = sum_i IP[b,i] W[a + N1 i]   (this lane: column a of W)
          const d2* rr = sB + colb;
          d2 p = rr[0], t = rr[NV];
          o[0] = iph[0] * p.x; o[1] = iph[0] * p.y; o[2] = iph[0] * t.x; o[3] = iph[0] * t.y;
#pragma unroll
          for (int i = 1; i < N1; ++i) {
            p = rr[N1 * i]; t = rr[NV + N1 * i];
            o[0] = __builtin_fma(iph[i], p.x, o[0]); o[1] = __builtin_fma(iph[i], p.y, o[1]);
            o[2] = __builtin_fma(iph[i], t.x, o[2]); o[3] = __builtin_fma(iph[i], t.y, o[3]);
          }
        }
      };
      pq_apply(R, out);
      if (WALLS && gb) {   // (uniform) second product for the part that keeps the record's J; correction per element and node
        double ox[4];
        pq_apply(RX, ox);
        if (sEbR[ev]) {
          const double gam = __builtin_fma(gJ, rcp_refined(Jn), -1.0);
#pragma unroll
          for (int f = 0; f < 4; ++f) out[f] = __builtin_fma(gam, out[f] - ox[f], out[f]);
        }
      }
    } else {
#pragma unroll
      for (int f = 0; f < 4; ++f) out[f] = R[f];
    }
    T2_STAMP(7);     // Pq
    if (vact) {
      if (lf.Qw) {   // fused low-storage RK stage (uniform)
        double ro[4], qo[4];   // (res and Qw are distinct arrays: all loads first, then the stores)
#pragma unroll
        for (int f = 0; f < 4; ++f) { const int64_t idx = f * KN + e0 * Nq + tvg; ro[f] = lf.res[idx]; qo[f] = lf.Qw[idx]; }
#pragma unroll
        for (int f = 0; f < 4; ++f) {
          const int64_t idx = f * KN + e0 * Nq + tvg;
          const double rr = __builtin_fma(lf.a, ro[f], lf.dt * out[f]);
          lf.res[idx] = rr;
          lf.Qw[idx] = __builtin_fma(lf.b, rr, qo[f]);
        }
      } else {
#pragma unroll
        for (int f = 0; f < 4; ++f) rhs[f * KN + e0 * Nq + tvg] = out[f];
      }
    }
    T2_STAMP(8);
    T2_STAMP_FLUSH;
  }
}

}  // namespace t2

#if ESDG_MAX_N1 >= 12
#define ESDG_T2_DISPATCH_HI(...) case 10: { constexpr int N1 = 10; __VA_ARGS__; } break; case 11: { constexpr int N1 = 11; __VA_ARGS__; } break; case 12: { constexpr int N1 = 12; __VA_ARGS__; } break;
#elif ESDG_MAX_N1 >= 10
#define ESDG_T2_DISPATCH_HI(...) case 10: { constexpr int N1 = 10; __VA_ARGS__; } break;
#else
#define ESDG_T2_DISPATCH_HI(...)
#endif
// (N1 = 2 ... 9: every kernel of this file -- ESDG_T2_DISPATCH9, kt2_rhs's packed rows end there; ESDG_T2_DISPATCH adds the degrees
// that only phases 0 and 1 serve.  Two stand-alone macros: a body that launches a kernel cannot pass through a second macro.)
#define ESDG_T2_CASES9(...)                          \
    case 2: { constexpr int N1 = 2; __VA_ARGS__; } break;   \
    case 3: { constexpr int N1 = 3; __VA_ARGS__; } break;   \
    case 4: { constexpr int N1 = 4; __VA_ARGS__; } break;   \
    case 5: { constexpr int N1 = 5; __VA_ARGS__; } break;   \
    case 6: { constexpr int N1 = 6; __VA_ARGS__; } break;   \
    case 7: { constexpr int N1 = 7; __VA_ARGS__; } break;   \
    case 8: { constexpr int N1 = 8; __VA_ARGS__; } break;   \
    case 9: { constexpr int N1 = 9; __VA_ARGS__; } break;
#define ESDG_T2_DISPATCH9(N1v, ...)                  \
  switch (N1v) {                                     \
    ESDG_T2_CASES9(__VA_ARGS__)                      \
    default: return (int)hipErrorInvalidValue;       \
  }
#define ESDG_T2_DISPATCH(N1v, ...)                   \
  switch (N1v) {                                     \
    ESDG_T2_CASES9(__VA_ARGS__)                      \
    ESDG_T2_DISPATCH_HI(__VA_ARGS__)                 \
    default: return (int)hipErrorInvalidValue;       \
  }

#ifdef ESDG_T2_STAMP
extern "C" int esdg_debug_stamps(unsigned long long* out16, int reset) {
  unsigned long long z[16] = {0};
  if (hipMemcpyFromSymbol(out16, HIP_SYMBOL(t2::g_stamp), sizeof z) != hipSuccess) return -1;
  if (reset && hipMemcpyToSymbol(HIP_SYMBOL(t2::g_stamp), z, sizeof z) != hipSuccess) return -1;
  return 0;
}
#endif

template <int N1, bool WALLS>
static void launch_sigma2w(const TensorTables& TT, const MeshDev& M, const Phys& ph, const double* Q, const double* A_U, double* B,
                           double* SG, double* vt, hipStream_t s) {
  using G = t2::GeoS<N1>;
  const int64_t nfull = M.e_count >= G::E ? (M.e_count + G::E - 1) / G::E : 0;   // complete groups, the last one shifted back
  if (nfull > 0) {
    int nb = t2::persistent_grid<t2::kt2_sigma<N1, true, WALLS>>(G::GT, nfull);
    // Ranged launches (sharded schedule): the interior leaves a few slots free and the boundary strips ask for no more
    // than those, so that a strip never takes the slot of an interior workgroup, which would start late and still do its
    // full static share (ESDG_T2_RESERVE: slots, default 64; 0 = off)
    // (the schedule says which launch is which -- MeshDev::launch_role; any other ranged launch, e.g. the pieces of
    // esdg_rhs_phase_range, gets the full grid)
    if (M.launch_role) {
      const int reserve = t2::g_reserve;
      if (reserve > 0) {
        const int cap = t2::persistent_grid<t2::kt2_sigma<N1, true, WALLS>>(G::GT, (int64_t)1 << 40);
        if (M.launch_role == 1) { if (nb > cap - reserve && cap - reserve > 0) nb = cap - reserve; }   // interior
        else if (nb > reserve) nb = reserve;                                                            // strip
      }
    }
    MeshDev Ml = M;
    Ml.wall_rot = (WALLS && M.wgeo) ? 1 : 0;   // spread the costly wall groups over the workgroups (MeshDev::wall_rot)
    hipLaunchKernelGGL((t2::kt2_sigma<N1, true, WALLS>), dim3(nb), dim3(G::GT), 0, s, TT, Ml, ph, Q, A_U, B, SG, vt);
  }
  else hipLaunchKernelGGL((t2::kt2_sigma<N1, false, WALLS>), dim3(1), dim3(G::GT), 0, s, TT, M, ph, Q, A_U, B, SG, vt);   // fewer than E elements
}
template <int N1>
static void launch_sigma2(const TensorTables& TT, const MeshDev& M, const Phys& ph, const double* Q, const double* A_U, double* B,
                          double* SG, double* vt, hipStream_t s) {
  if (M.bc) launch_sigma2w<N1, true>(TT, M, ph, Q, A_U, B, SG, vt, s);
  else launch_sigma2w<N1, false>(TT, M, ph, Q, A_U, B, SG, vt, s);
}

// phase 1; vt_partial != null: also the visc_test partials, one per workgroup (at most SIGMA2_MAX_PARTIALS; the caller zeroes
// the buffer and sums all of it)
int launch_sigma_tensor2(int N1v, const TensorTables& TT, const MeshDev& M, const Phys& ph, const double* Q, const double* A_U,
                         double* B, double* SG, hipStream_t s, double* vt_partial) {
  if (M.e_count <= 0) return 0;
  ESDG_T2_DISPATCH(N1v, launch_sigma2<N1>(TT, M, ph, Q, A_U, B, SG, vt_partial, s));
  return (int)hipGetLastError();
}

template <int N1, bool MODAL, bool VISC>
static void launch_rhs2(const TensorTables& TT, const MeshDev& M, const Phys& ph, const double* Q, const double* A_U, const double* SG,
                        const double* B, double* rhs, const LsrkFuse& lf, hipStream_t s) {
  using G = t2::GeoR<N1>;
  const int nb = (int)((M.e_count + G::E - 1) / G::E);
  if (M.bc) hipLaunchKernelGGL((t2::kt2_rhs<N1, MODAL, VISC, true>), dim3(nb), dim3(G::GT), 0, s, TT, M, ph, Q, A_U, SG, B, rhs, lf);
  else hipLaunchKernelGGL((t2::kt2_rhs<N1, MODAL, VISC, false>), dim3(nb), dim3(G::GT), 0, s, TT, M, ph, Q, A_U, SG, B, rhs, lf);
}

// last phase on meshes without walls; returns -1 where the v2 kernel does not cover the degree (caller falls back)
int launch_rhs_tensor2(int N1v, const TensorTables& TT, const MeshDev& M, const Phys& ph, const double* Q, const double* A_U,
                       const double* SG, const double* B, double* rhs, const LsrkFuse& lf, hipStream_t s) {
  if (M.e_count <= 0) return 0;
  if (N1v < 2 || N1v > 9) return -1;   // (RhsRows packs a node's NRND <= 8 partner ids into two ints: N1 = 10 has nine)
  const bool modal = ph.formulation != 0, visc = ph.formulation == 1;
  ESDG_T2_DISPATCH9(N1v, {
    {
      if (!modal) (launch_rhs2<N1, false, false>)(TT, M, ph, Q, A_U, SG, B, rhs, lf, s);
      else if (visc) (launch_rhs2<N1, true, true>)(TT, M, ph, Q, A_U, SG, B, rhs, lf, s);
      else (launch_rhs2<N1, true, false>)(TT, M, ph, Q, A_U, SG, B, rhs, lf, s);
    }
  });
  return (int)hipGetLastError();
}

void ab_tuning_t2(int wg_per_cu, int reserve) {
  if (wg_per_cu >= 0) t2::g_wg_per_cu = wg_per_cu;
  if (reserve >= 0) t2::g_reserve = reserve;
}

// (A/B hook ESDG_T2_PROJECT_PERSIST = P > 0: the persistent grid when the launch has more than P groups per resident workgroup)
template <int N1>
static int project_grid(bool modal, int nb) {
#if ESDG_T2_PROJECT_PERSIST
  using G = t2::Geo<N1>;
  constexpr auto km = t2::kt2_project<N1, true>;
  constexpr auto kc = t2::kt2_project<N1, false>;
  const int cap = modal ? t2::persistent_grid<km>(G::GT, (int64_t)1 << 40) : t2::persistent_grid<kc>(G::GT, (int64_t)1 << 40);
  if (nb > cap * ESDG_T2_PROJECT_PERSIST) nb = (int)((int64_t)cap * ESDG_T2_PROJECT_GRID_NUM / ESDG_T2_PROJECT_GRID_DEN);
#else
  (void)modal;
#endif
  return nb;
}

// phase 0 with the v2 kernel; returns -1 where it does not cover the degree (the caller refuses the degree)
int launch_project_tensor2(int N1v, const TensorTables& TT, const MeshDev& M, const Phys& ph, const double* Q, double* A_U, hipStream_t s) {
  if (M.e_count <= 0) return 0;
  if (!tensor2d_supported_degree(N1v)) return -1;
  const bool modal = ph.formulation != 0;
  ESDG_T2_DISPATCH(N1v, {
    using G = t2::Geo<N1>;
    int nb = (int)((M.e_count + G::E - 1) / G::E);
    nb = project_grid<N1>(modal, nb);
    if (modal) hipLaunchKernelGGL((t2::kt2_project<N1, true>), dim3(nb), dim3(G::GT), 0, s, TT, M, Q, A_U);
    else hipLaunchKernelGGL((t2::kt2_project<N1, false>), dim3(nb), dim3(G::GT), 0, s, TT, M, Q, A_U);
  });
  return (int)hipGetLastError();
}

}  // namespace esdg
