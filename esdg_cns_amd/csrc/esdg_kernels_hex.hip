// esdg_kernels_hex.hip -- gfx950 (MI355X / CDNA4) kernels for the collocated entropy-stable Euler right-hand
// side on hexahedra: `rhs` of examples/dg3D_euler_hex.jl:167-222 with sparse_hadamard_sum (:122-164), the 3D
// entropy-conservative flux (examples/EntropyStableEuler/euler_fluxes.jl:51-89, logmean.jl:14-28) and the 3D
// entropy-variable maps (euler_variables.jl:79-120).
//
// Mapping (N = 3: Nq = 64 Gauss nodes = exactly one wavefront):
//   * one wave per element, lane <-> Gauss node (i0,i1,i2); 4 independent waves per workgroup share the 1D
//     operator tables in LDS (esdg_hex_tables.hpp), each wave owns its LDS slice.  The first Nfq = 6 N1^2 lane
//     slots (1.5 rounds at N = 3) double as face-node lanes.
//   * two phases with the face-trace protocol of the 2D path:
//       phase 0  kh_project : v(u) at the Gauss nodes, extrapolation to the 96 face nodes (4-term line dot
//                             products, Ef has N+1 non-zeros per row), u(v) -> (rho,u,v,w,beta) trace record
//       phase 1  kh_rhs     : own + neighbour traces (mapP gather), surface flux (+LF), flux differencing, lift
//   * flux differencing: each unordered pair once.  Volume-volume pairs by the circulant line schedule (per
//     direction N1/2 rounds, lane = lower node, partner share pushed with ds_add_f64); volume-face pairs are
//     walked by the FACE lanes (4 per face node; the two ends of a line start at opposite offsets, so no two
//     lanes ever hit the same accumulator in one instruction).  672 EC fluxes per element at N = 3 (the
//     reference evaluates 1344), each a *directional* flux g.F with the pair's metric vector g -- affine
//     elements only, so the reference's per-pair metric average (:145-146) is the element constant.
//   * one wave owns an element, so every accumulation order is fixed by the instruction stream: results are
//     bitwise reproducible.
//   * workgroup -> element-block mapping is XCD-aware: consecutive workgroup ids rotate over the 8 XCDs, so
//     XCD x is given the x-th contiguous eighth of the element range and face neighbours meet in one L2.
#include "esdg_dev.hpp"
#include "esdg_devmath.hpp"
#include "esdg_hex_tables.hpp"
#include "esdg_t2_physics.hpp"   // SeriesK and the series polynomials shared with the 2D flux

namespace esdg {
namespace hdev {

using namespace devmath;

constexpr double GM1 = 1.4 - 1;   // gamma - 1 with the package's gamma (EntropyStableEuler.jl:9)
constexpr int HW = 64;            // wave
constexpr int HNWV = 4;           // waves (= elements) per workgroup
constexpr int NXCD = 8;
#ifndef ESDG_KH_COALESCE
#define ESDG_KH_COALESCE 1
#endif

// J2 experiment (A/B hook, off): in kh_rhs one wave is one element and at N1 = 4 a lane's direction-0
// partners sit in its own quad (lane = i0 + 4 i1 + 16 i2), so the full round of direction 0 can fetch the partner's record
// and return the partner's share through DPP quad_perm moves instead of LDS (7 ds_read_b64 + 5 ds_add_f64 -> 24 + 1
// v_mov_b32_dpp).  Measured in round 4 (profiles/experiments/README.md).
#ifndef ESDG_KH_DPP
#define ESDG_KH_DPP 0
#endif
template <int CTRL> __device__ __forceinline__ double dpp_quad(double x) {
  const int lo = __builtin_amdgcn_mov_dpp(__double2loint(x), CTRL, 0xf, 0xf, true);
  const int hi = __builtin_amdgcn_mov_dpp(__double2hiint(x), CTRL, 0xf, 0xf, true);
  return __hiloint2double(hi, lo);
}
constexpr int DPP_QUAD_NEXT = 0x39;   // quad_perm [1,2,3,0]: lane i reads lane (i + 1) % 4 of its quad
constexpr int DPP_QUAD_PREV = 0x93;   // quad_perm [3,0,1,2]: lane i reads lane (i - 1) % 4 of its quad

template <int N1> struct HCfg {
  static constexpr int Nq = N1 * N1 * N1, Nfq = 6 * N1 * N1, NIT = (Nfq + HW - 1) / HW;
};

// conservative (rho, rho u, rho v, rho w, E) -> (rho,u,v,w,beta,log rho,log beta); one reciprocal
__device__ __forceinline__ void prim_logs3(const double* U, double* q) {
  const double m2 = U[1] * U[1] + U[2] * U[2] + U[3] * U[3];
  const double rre = U[0] * U[4] - .5 * m2;            // rho * rhoe
  const double R = rcp_refined(U[0] * rre);            // 1/(rho^2 rhoe)
  const double ir = R * rre;                           // 1/rho
  q[0] = U[0];
  q[1] = U[1] * ir;
  q[2] = U[2] * ir;
  q[3] = U[3] * ir;
  q[4] = (U[0] * U[0]) * (U[0] * R) * (1.0 / (2 * GM1));   // beta = rho/(2 (gamma-1) rhoe)
  q[5] = log_pos(U[0]);
  q[6] = log_pos(q[4]);
}

// entropy variables from (rho,u,v,w,beta,logs): identities of euler_variables.jl:79-92
__device__ __forceinline__ void v_of_prim3(const double* q, double* V) {
  const double s = -GM1 * q[5] - q[6] - 0.6931471805599453;
  const double b2 = 2 * GM1 * q[4];                    // rho/rhoe
  V[0] = 1.4 - s - .5 * b2 * (q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
  V[1] = b2 * q[1];
  V[2] = b2 * q[2];
  V[3] = b2 * q[3];
  V[4] = -b2;
}

// (rho,u,v,w,beta) of entropy variables: u_vfun (euler_variables.jl:95-120) followed by the driver's
// u = rhoU/rho, beta = betafun(...) (dg3D_euler_hex.jl:179-182), which reduce to u_i = -v_i/v5, beta = -v5/(2(gamma-1))
__device__ __forceinline__ void prim_of_v3(const double* V, double* q) {
  const double iv = rcp_refined(V[4]);
  const double h = (V[1] * V[1] + V[2] * V[2] + V[3] * V[3]) * .5 * iv;
  const double s = 1.4 - V[0] + h;
  const double rhoeV = exp((log(GM1) - 1.4 * log_pos(-V[4]) - s) * (1.0 / GM1));
  q[0] = rhoeV * (-V[4]);
  q[1] = -V[1] * iv;
  q[2] = -V[2] * iv;
  q[3] = -V[3] * iv;
  q[4] = V[4] * (-1.0 / (2 * GM1));
}

// g . (Fx,Fy,Fz) of the entropy-conservative flux (euler_fluxes.jl:51-89), q = (rho,u,v,w,beta,lrho,lbeta), for the metric vector
// g = 2 (hgx, hgy, hgz): the callers pass HALF the metric vector (a loop invariant, or a half scale of their packed differences).
// One reciprocal serves the rho log-mean, 1/(beta log-mean) and pa.  Same construction as ec_flux_core of the 2D tensor kernels
// (esdg_t2_physics.hpp, round 5): sums instead of averages with every 1/2 folded into a constant by exact power-of-two scaling,
// explicit FMAs throughout -- so the three variants below give every lane the same bits whichever its wave runs (the sharded and
// ranged tests compare across wave compositions) --, and the series constants that must sit in VGPRs passed in by the kernel.
// MODE 1 = every active lane takes both series (no log differences, no selects), MODE 2 = no lane takes a series (no
// polynomials), MODE 0 = per-lane selection (logmean.jl:23-27 as written).
template <int MODE>
__device__ __forceinline__ void ec_flux_core3(const double* qL, const double* qR, double hgx, double hgy, double hgz, double* F, double dr,
                                              double sr, double db, double sb, bool ser_r, bool ser_b, const t2::SeriesK& sk) {
  double yr, yb;
  if (MODE == 1) { yr = sr; yb = sb; }
  else {
    const double A = qL[5] - qR[5];
    yr = MODE == 2 ? A : (ser_r ? sr : A);
    yb = MODE == 2 ? db : (ser_b ? sb : db);
  }
  const double ybp = yb * sb;
  const double R = rcp_refined(yr * ybp);
  const double ir = R * ybp;
  const double ryr = R * yr;
  const double ib = ryr * sb;
  const double ip = ryr * yb;
  const double fr = dr * ir;
  const double fb = db * ib;
  double rholog, ibetalog;
  if (MODE == 2) {
    rholog = -fr;
    ibetalog = -((qL[6] - qR[6]) * ib);
  } else {
    const double srs = sr * t2::logmean_series_rho_h(fr * fr, sk), sbs = ib * t2::logmean_series_ibeta_2(fb * fb, sk);
    if (MODE == 1) { rholog = srs; ibetalog = sbs; }
    else { rholog = ser_r ? srs : -fr; ibetalog = ser_b ? sbs : -((qL[6] - qR[6]) * ib); }
  }
  const double su = qL[1] + qR[1], sv = qL[2] + qR[2], sw = qL[3] + qR[3];
  const double unorm = __builtin_fma(qL[3], qR[3], __builtin_fma(qL[2], qR[2], qL[1] * qR[1]));
  const double pa2 = sr * ip;                                                                    // 2 pa
  const double Ep2 = __builtin_fma(rholog, __builtin_fma(ibetalog, 1.0 / GM1, unorm), pa2);      // 2 (rholog / (2 (g-1) betalog) + pa + rholog uL.uR / 2)
  const double un = __builtin_fma(hgz, sw, __builtin_fma(hgy, sv, hgx * su));
  F[0] = rholog * un;
  const double hF0 = .5 * F[0];
  F[1] = __builtin_fma(hF0, su, pa2 * hgx);
  F[2] = __builtin_fma(hF0, sv, pa2 * hgy);
  F[3] = __builtin_fma(hF0, sw, pa2 * hgz);
  F[4] = Ep2 * (.5 * un);
}
// per-lane selection (kh_rhs, kh_rhs_g, the interface fluxes of kh_project-side checks): full metric vector in
__device__ __forceinline__ void ec_flux_dir(const double* qL, const double* qR, double gx, double gy, double gz, double* F) {
  const double dr = qR[0] - qL[0], sr = qR[0] + qL[0];
  const double db = qR[4] - qL[4], sb = qR[4] + qL[4];
  const bool ser_r = fabs(dr) < (.5 * 1e-4) * sr, ser_b = fabs(db) < (.5 * 1e-4) * sb;
  ec_flux_core3<0>(qL, qR, .5 * gx, .5 * gy, .5 * gz, F, dr, sr, db, sb, ser_r, ser_b, t2::series_k());
}

// The branch of the two log-means decided per WAVE where the wave agrees (kh_rhs_l; as ec_flux_dir of the 2D tensor kernels): one
// ballot per comparison -- the v_cmp's own lane mask -- and scalar logic on the masks.  HALF metric vector in.
#ifndef ESDG_KHL_UNIFORM_LOGMEAN
#define ESDG_KHL_UNIFORM_LOGMEAN 1   // (A/B hook: 0 = the selecting form everywhere)
#endif
__device__ __forceinline__ void ec_flux_dir_u(const double* qL, const double* qR, double hgx, double hgy, double hgz, double* F, const t2::SeriesK& sk) {
  const double dr = qR[0] - qL[0], sr = qR[0] + qL[0];
  const double db = qR[4] - qL[4], sb = qR[4] + qL[4];
  const bool ser_r = fabs(dr) < (.5 * 1e-4) * sr, ser_b = fabs(db) < (.5 * 1e-4) * sb;
#if ESDG_KHL_UNIFORM_LOGMEAN
  const unsigned long long active = __builtin_amdgcn_ballot_w64(true), br = __builtin_amdgcn_ballot_w64(ser_r), bb = __builtin_amdgcn_ballot_w64(ser_b);
  if ((br | bb) == 0) ec_flux_core3<2>(qL, qR, hgx, hgy, hgz, F, dr, sr, db, sb, ser_r, ser_b, sk);
  else if ((br & bb) == active) ec_flux_core3<1>(qL, qR, hgx, hgy, hgz, F, dr, sr, db, sb, ser_r, ser_b, sk);
  else ec_flux_core3<0>(qL, qR, hgx, hgy, hgz, F, dr, sr, db, sb, ser_r, ser_b, sk);
#else
  ec_flux_core3<0>(qL, qR, hgx, hgy, hgz, F, dr, sr, db, sb, ser_r, ser_b, sk);
#endif
}

// |wavespeed(rho, rhoU_n, E)| of dg3D_euler_hex.jl:190-192 (euler_variables.jl:7-10, sqrt(|u_n|) quirk Q1),
// from a primitive record; also returns the conservative vector
__device__ __forceinline__ double lf_lambda3(const double* q, double nx, double ny, double nz, double isJ, double* U) {
  U[0] = q[0];
  U[1] = q[0] * q[1];
  U[2] = q[0] * q[2];
  U[3] = q[0] * q[3];
  const double p = q[0] * .5 * rcp_refined(q[4]);
  U[4] = p * (1.0 / GM1) + .5 * q[0] * (q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
  const double ir = rcp_refined(q[0]);
  const double rhoUn = (U[1] * nx + U[2] * ny + U[3] * nz) * isJ;
  const double pn = GM1 * (U[4] - .5 * (rhoUn * rhoUn) * ir);
  return fabs(sqrt(fabs(rhoUn * ir)) + sqrt(1.4 * pn * ir));
}

// LDS slot of a volume node.  The face lanes of one face read/update the 16 nodes of a plane parallel to it; in the
// natural order i0 + 4 i1 + 16 i2 the planes i0 = const and i1 = const fall on a quarter of the LDS banks (4-way
// conflicts on every ds_read_b64 / ds_add_f64).  XOR-ing both 2-bit fields of the low nibble with i2 maps the 16 nodes
// of all three plane families (and any 16 consecutive nodes) to 16 distinct double-width banks.
__device__ __forceinline__ int slot_of(int node) { return node ^ ((node >> 4) * 5); }

template <int N1>
__device__ __forceinline__ void stage_tables(const HexTables& HT, double* sTab, int* sInt) {
  constexpr HexLayout L(N1);
  for (int i = threadIdx.x; i < L.NDBL; i += HW * HNWV) sTab[i] = HT.dbl[i];
  for (int i = threadIdx.x; i < L.NINT; i += HW * HNWV) sInt[i] = HT.ints[i];
}

// The same copy in two halves, for kernels whose other loads are in flight at the same time: load() issues every table load
// into registers without waiting (all lanes, the index clamped: lanes beyond the table re-read its last entry), store() writes
// them to LDS (duplicate lanes write the same value to the same slot).  A strided `for (i = tid; i < n; ...)` copy, or any
// LDS store under `if (lane < n)`, makes hipcc put the load next to the store and wait for it with vmcnt(0) -- i.e. for
// every load of the wave in flight, state and neighbour traces included, once per round of the copy.
template <int N1>
struct TableRegs {
  static constexpr HexLayout L = HexLayout(N1);
  static constexpr int T = HW * HNWV, ND = (L.NDBL + T - 1) / T, NI = (L.NINT + T - 1) / T;
  double d[ND];
  int i[NI];
  __device__ __forceinline__ void load(const HexTables& HT) {
#pragma unroll
    for (int r = 0; r < ND; ++r) d[r] = HT.dbl[min((int)threadIdx.x + r * T, L.NDBL - 1)];
#pragma unroll
    for (int r = 0; r < NI; ++r) i[r] = HT.ints[min((int)threadIdx.x + r * T, L.NINT - 1)];
  }
  __device__ __forceinline__ void store(double* sTab, int* sInt) const {
#pragma unroll
    for (int r = 0; r < ND; ++r) sTab[min((int)threadIdx.x + r * T, L.NDBL - 1)] = d[r];
#pragma unroll
    for (int r = 0; r < NI; ++r) sInt[min((int)threadIdx.x + r * T, L.NINT - 1)] = i[r];
  }
};

// XCD-aware element-block id (see the header); returns -1 for padding workgroups
__device__ __forceinline__ int64_t block_of(int64_t nblk, bool remap) {
  const int64_t b = blockIdx.x;
  if (!remap) return b < nblk ? b : -1;
  const int64_t chunk = (nblk + NXCD - 1) / NXCD;
  const int64_t blk = (b % NXCD) * chunk + b / NXCD;
  return blk < nblk ? blk : -1;
}

// line of direction d through face node code (d | t<<2 | o<<3): base node and stride
template <int N1>
__device__ __forceinline__ void line_of(int d, int o, int& base, int& stride) {
  stride = d == 0 ? 1 : (d == 1 ? N1 : N1 * N1);
  base = d == 0 ? N1 * o : (d == 1 ? (o % N1) + N1 * N1 * (o / N1) : o);
}

// ---- phase 0 -----------------------------------------------------------------------------------------------
template <int N1>
__global__ __launch_bounds__(HW * HNWV) void kh_project(HexTables HT, MeshDev M, int remap,
                                                         const double* __restrict__ Q, double* __restrict__ A_U) {
  constexpr HexLayout L(N1);
  constexpr int Nq = HCfg<N1>::Nq, Nfq = HCfg<N1>::Nfq, NIT = HCfg<N1>::NIT;
  __shared__ double sTab[L.NDBL];
  __shared__ int sInt[L.NINT];
  __shared__ double sV[HNWV][HEX_NFLD * HW];
  // ESDG_KH_COALESCE: an element's 5 Nfq trace doubles leave through a wave-private LDS block, so that every store instruction
  // writes 512 contiguous bytes instead of 64 x 8 B at a stride of 40 B (20 cache lines per instruction)
  __shared__ double sOut[ESDG_KH_COALESCE ? HNWV : 1][ESDG_KH_COALESCE ? HEX_AU_NC * Nfq : 1];
  const int64_t nblk = (M.e_count + HNWV - 1) / HNWV;
  const int64_t blk = block_of(nblk, remap != 0);
  if (blk < 0) return;
  const int lane = threadIdx.x & (HW - 1), wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x / HW));   // wave-uniform: keep it scalar
  const int64_t e = M.e_begin + blk * HNWV + wv;
  const bool active = e < M.e_begin + M.e_count;
  const int64_t ec = active ? e : M.e_begin + M.e_count - 1;
  const bool vin = lane < Nq;
  double U[HEX_NFLD];
  prio_entry_begin<ESDG_PRIO_KH_PROJECT>();
  TableRegs<N1> tr;
  tr.load(HT);
  {
    const int lu = vin ? lane : 0;   // (lanes without a node compute on node 0's data; their results are not stored)
#pragma unroll
    for (int c = 0; c < HEX_NFLD; ++c) U[c] = Q[(int64_t)c * M.K * Nq + ec * Nq + lu];
  }
  prio_entry_end<ESDG_PRIO_KH_PROJECT>();
  tr.store(sTab, sInt);
  double q[7], V[HEX_NFLD];
  prim_logs3(U, q);
  v_of_prim3(q, V);
  if (vin) {
#pragma unroll
    for (int c = 0; c < HEX_NFLD; ++c) sV[wv][c * HW + slot_of(lane)] = V[c];
  }
  __syncthreads();
#pragma unroll
  for (int it = 0; it < NIT; ++it) {
    const int f = lane + HW * it;
    if (f < Nfq) {
      const int code = sInt[L.FINV + f];
      const int d = code & 3, t = (code >> 2) & 1, o = code >> 3;
      int base, stride;
      line_of<N1>(d, o, base, stride);
      double Vf[HEX_NFLD] = {0.0, 0.0, 0.0, 0.0, 0.0};
#pragma unroll
      for (int i = 0; i < N1; ++i) {
        const double w = sTab[L.EE + (d * 2 + t) * N1 + i];
#pragma unroll
        for (int c = 0; c < HEX_NFLD; ++c) Vf[c] += w * sV[wv][c * HW + slot_of(base + i * stride)];
      }
      double qf[HEX_NFLD];
      prim_of_v3(Vf, qf);
      if (ESDG_KH_COALESCE) {
#pragma unroll
        for (int c = 0; c < HEX_NFLD; ++c) sOut[wv][f * HEX_AU_NC + c] = qf[c];
      } else if (active) {
#pragma unroll
        for (int c = 0; c < HEX_NFLD; ++c) A_U[(e * Nfq + f) * HEX_AU_NC + c] = qf[c];
      }
    }
  }
  if (ESDG_KH_COALESCE) {
    // (one wave = one element: its LDS instructions execute in order, a later read sees every lane's earlier write)
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    if (active) {
      constexpr int NT = HEX_AU_NC * Nfq;
#pragma unroll
      for (int k = 0; k < (NT + HW - 1) / HW; ++k) {
        const int idx = k * HW + lane;
        if (idx < NT) A_U[e * NT + idx] = sOut[wv][idx];
      }
    }
  }
}

// ---- phase 1 -----------------------------------------------------------------------------------------------
// Geometry mode GM:
//   1 (curved meshes): per-node metric terms / normals / J (M.G9, M.nrm, M.Jq) and the reference's per-pair metric average
//     .5 (G_i + G_j) (dg3D_euler_hex.jl:145-151);
//   0: one affine record per element (means of what the driver passed) -- what a driver that passes one metric row per
//     element (geo_ld = 1) gets, or ESDG_HEX_GEOMETRY=element asks for;
//   2 (affine meshes whose driver passed per-node arrays, the default there): the same per-node use as mode 1 from the
//     element record plus each node's DIFFERENCE to it, 10 bits per number (M.hdv / hdf / hdn, scales in the record).  The
//     per-node arrays of an affine mesh are constants plus the round-off of the driver's set-up (1e-13 ... 1e-12 relative),
//     and the reference's per-node use turns that into 2.5 x (16^3) ... 3.8 x (24^3) ... its own rounding error in the RHS
//     (tools/hex_geometry_probe.py); mode 0 filters it out, mode 1 costs 10.4 KB more traffic per element (kh_rhs 1.17 ->
//     1.69 ms at 128x128x16), mode 2 reproduces it to 1/1022 of its largest amplitude for 1.5 KB (emulated in the oracle at
//     16^3: 8 bits move the RHS by 0.33 e_orc, 10 bits by 0.08, 12 by 0.02, 16 by 0.005).
template <int N1, int GM> struct HexGeo { static constexpr bool CURVED = GM == 1, DELTA = GM == 2; };

// (x, y, z) differences of one node packed as three signed 10-bit fields of one word (v_bfe_i32 each); g = base + scale *
// (k_i + k_j) averages a pair when scale carries the factor 1/2
__device__ __forceinline__ void unpack3(unsigned p, int& a, int& b, int& c) {
  a = __builtin_amdgcn_sbfe((int)p, 0, 10);
  b = __builtin_amdgcn_sbfe((int)p, 10, 10);
  c = __builtin_amdgcn_sbfe((int)p, 20, 10);
}

template <int N1, int GM>
__global__ __launch_bounds__(HW * HNWV) void kh_rhs(HexTables HT, MeshDev M, Phys ph, int remap,
                                                     const double* __restrict__ Q, const double* __restrict__ A_U,
                                                     double* __restrict__ rhs, LsrkFuse lf) {
  constexpr HexLayout L(N1);
  constexpr int Nq = HCfg<N1>::Nq, Nfq = HCfg<N1>::Nfq, NIT = HCfg<N1>::NIT, NN = N1 * N1;
  __shared__ double sTab[L.NDBL];
  __shared__ int sInt[L.NINT];
  __shared__ double sGeo[HNWV][HEX_GEO_STRIDE + 2];
  __shared__ double sPs[HNWV][7 * HW];
  __shared__ double sAccs[HNWV][HEX_NFLD * HW];
  __shared__ double sGs[HNWV][HEX_NFLD * Nfq];
  constexpr bool CURVED = HexGeo<N1, GM>::CURVED, DELTA = HexGeo<N1, GM>::DELTA;
  __shared__ double sMs[CURVED ? HNWV : 1][CURVED ? 9 * HW : 1];   // metric terms of the volume nodes, [c*3 + op][slot]
  __shared__ unsigned sDs[DELTA ? HNWV : 1][DELTA ? 3 * HW : 1];   // packed metric differences of the volume nodes, [op][slot]
  const int64_t nblk = (M.e_count + HNWV - 1) / HNWV;
  const int64_t blk = block_of(nblk, remap != 0);
  if (blk < 0) return;
  const int lane = threadIdx.x & (HW - 1), wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x / HW));   // wave-uniform: keep it scalar
  const int64_t e = M.e_begin + blk * HNWV + wv;
  const bool active = e < M.e_begin + M.e_count;
  const int64_t ec = active ? e : M.e_begin + M.e_count - 1;
  const bool vin = lane < Nq;
  double* sP = sPs[wv];
  double* sAcc = sAccs[wv];
  double* sG = sGs[wv];
  const double* geo = sGeo[wv];
  double* sM = sMs[CURVED ? wv : 0];
  unsigned* sD = sDs[DELTA ? wv : 0];
  constexpr int Nh = Nq + Nfq;

  // ---- issue the global loads ---------------------------------------------------------------------
  // Every load below is unconditional and issued before anything is waited for (see TableRegs): the neighbour index first
  // (the neighbour's traces depend on it), then tables, geometry and state, which the pointwise stage needs, and the
  // traces of the first face round last; the waits that follow are counted, none is vmcnt(0).
  double rm[HEX_AU_NC], rp[HEX_AU_NC];
  const int fc0 = lane < Nfq ? lane : Nfq - 1;
  const int64_t nm0 = ec * Nfq + fc0;
  prio_entry_begin();
  const int64_t np0 = M.mapP[nm0];
  TableRegs<N1> tr;
  tr.load(HT);
  const int gl = lane < HEX_GEO_STRIDE ? lane : HEX_GEO_STRIDE - 1;
  const double geo_r = M.geo[ec * HEX_GEO_STRIDE + gl];
  double U[HEX_NFLD];
  {
    const int lu = vin ? lane : 0;
#pragma unroll
    for (int c = 0; c < HEX_NFLD; ++c) U[c] = Q[(int64_t)c * M.K * Nq + ec * Nq + lu];
  }
  // raw traces of the first face-node round (own + neighbour through mapP); later rounds are prefetched one
  // round ahead inside the face loop
#pragma unroll
  for (int c = 0; c < HEX_AU_NC; ++c) rm[c] = A_U[nm0 * HEX_AU_NC + c];
#pragma unroll
  for (int c = 0; c < HEX_AU_NC; ++c) rp[c] = A_U[np0 * HEX_AU_NC + c];
  unsigned kv[3] = {0u, 0u, 0u}, kf0 = 0u, kn0 = 0u;   // DELTA: this node's packed metric differences per operator; face round 0's
  if (DELTA) {
    const int lu = vin ? lane : 0;
#pragma unroll
    for (int o3 = 0; o3 < 3; ++o3) kv[o3] = M.hdv[(ec * 3 + o3) * Nq + lu];
    kf0 = M.hdf[nm0];
    kn0 = M.hdn[nm0];
  }
  prio_entry_end();
  sGeo[wv][gl] = geo_r;
  tr.store(sTab, sInt);
#pragma unroll
  for (int c = 0; c < HEX_NFLD; ++c) sAcc[c * HW + lane] = 0.0;
  const int myslot = slot_of(lane);
  if (CURVED && vin) {
#pragma unroll
    for (int m9 = 0; m9 < 9; ++m9) sM[m9 * HW + myslot] = M.G9[(ec * 9 + m9) * Nh + lane];
  }
  if (DELTA) {
#pragma unroll
    for (int o3 = 0; o3 < 3; ++o3) sD[o3 * HW + myslot] = kv[o3];
  }

  // ---- pointwise: primitives + logs ------------------------------------------------------------------
  double hsG = 0.0, sNs = 0.0;   // DELTA: half the scale of the packed metric differences, scale of the packed normal differences
  double acc[HEX_NFLD] = {0.0, 0.0, 0.0, 0.0, 0.0};
  const int lq = vin ? lane : 0;
  const int i0 = lq % N1, i1 = (lq / N1) % N1, i2 = lq / NN;
  {
    double qv[7];
    prim_logs3(U, qv);
#pragma unroll
    for (int c = 0; c < 7; ++c) sP[c * HW + myslot] = qv[c];
    __syncthreads();
    if (DELTA) { hsG = .5 * geo[HEX_GEO_STRIDE - 2]; sNs = geo[HEX_GEO_STRIDE - 1]; }

    // ---- volume lanes: circulant line schedule, each unordered pair once -------------------------------
    // full rounds: lane at position i of a line takes the pair (i, i+m mod N1), m = 1..(N1-1)/2
    constexpr int MF = (N1 - 1) / 2;
    if (MF > 0) {
#pragma unroll 1
      for (int d = 0; d < 3; ++d) {
        const int opd = d == 0 ? HT.op[0] : (d == 1 ? HT.op[1] : HT.op[2]);
        const double gx = geo[opd], gy = geo[3 + opd], gz = geo[6 + opd];
        const int id = d == 0 ? i0 : (d == 1 ? i1 : i2);
        const int o = d == 0 ? i1 + N1 * i2 : (d == 1 ? i0 + N1 * i2 : i0 + N1 * i1);
        const int stride = d == 0 ? 1 : (d == 1 ? N1 : NN);
        const double wt = sTab[L.WT + d * NN + o];
#pragma unroll
        for (int m = 1; m <= MF; ++m) {
          int j = id + m;
          j = j >= N1 ? j - N1 : j;
          const int node = vin ? lane + (j - id) * stride : lane;
          double qn[7], F[HEX_NFLD];
          const int ns = slot_of(node);
          const bool dpp = ESDG_KH_DPP && N1 == 4 && !CURVED && d == 0 && m == 1;   // (uniform)
          if (dpp) {
#pragma unroll
            for (int c = 0; c < 7; ++c) qn[c] = dpp_quad<DPP_QUAD_NEXT>(qv[c]);
          } else {
#pragma unroll
            for (int c = 0; c < 7; ++c) qn[c] = sP[c * HW + ns];
          }
          double W = sTab[L.S + (d * N1 + id) * N1 + j] * wt;
          if (dpp) {
            if (DELTA) {
              int a0, a1, a2, b0, b1, b2;
              const unsigned kown = opd == 0 ? kv[0] : (opd == 1 ? kv[1] : kv[2]);
              unpack3(kown, a0, a1, a2);
              unpack3((unsigned)__builtin_amdgcn_mov_dpp((int)kown, DPP_QUAD_NEXT, 0xf, 0xf, true), b0, b1, b2);
              ec_flux_dir(qv, qn, __builtin_fma(hsG, (double)(a0 + b0), gx), __builtin_fma(hsG, (double)(a1 + b1), gy),
                          __builtin_fma(hsG, (double)(a2 + b2), gz), F);
            } else {
              ec_flux_dir(qv, qn, gx, gy, gz, F);
            }
#pragma unroll
            for (int c = 0; c < HEX_NFLD; ++c) {   // own share, and the share of the pair whose partner this lane is
              const double wf = W * F[c];
              acc[c] += wf;
              acc[c] -= dpp_quad<DPP_QUAD_PREV>(wf);
            }
            continue;
          }
          if (CURVED) {   // metric of the pair = average of the two nodes (the .5 goes into the weight)
            W *= .5;
            ec_flux_dir(qv, qn, sM[opd * HW + myslot] + sM[opd * HW + ns], sM[(3 + opd) * HW + myslot] + sM[(3 + opd) * HW + ns],
                        sM[(6 + opd) * HW + myslot] + sM[(6 + opd) * HW + ns], F);
          } else if (DELTA) {   // the same average: record + half the scale times the two nodes' differences
            int a0, a1, a2, b0, b1, b2;
            unpack3(opd == 0 ? kv[0] : (opd == 1 ? kv[1] : kv[2]), a0, a1, a2);
            unpack3(sD[opd * HW + ns], b0, b1, b2);
            ec_flux_dir(qv, qn, __builtin_fma(hsG, (double)(a0 + b0), gx), __builtin_fma(hsG, (double)(a1 + b1), gy),
                        __builtin_fma(hsG, (double)(a2 + b2), gz), F);
          } else {
            ec_flux_dir(qv, qn, gx, gy, gz, F);
          }
          if (vin) {
#pragma unroll
            for (int c = 0; c < HEX_NFLD; ++c) {
              const double wf = W * F[c];
              acc[c] += wf;
              lds_add(&sAcc[c * HW + ns], -wf);
            }
          }
        }
      }
    }
    // even N1: the antipodal pairs (i, i+N1/2).  Only half of the lanes of a line own such a pair per direction, so the
    // three directions are packed into two rounds instead of three: with b_d = (i_d >= N1/2) and S = {b0^b1^b2 = 1},
    // every antipodal pair has exactly one endpoint in S; round A: S-lanes take their d=0 pair, the others their d=1
    // pair; round B: S-lanes take their d=2 pair.
    if (N1 % 2 == 0) {
      constexpr int H = N1 / 2;
      const bool inS = (((i0 >= H) ? 1 : 0) ^ ((i1 >= H) ? 1 : 0) ^ ((i2 >= H) ? 1 : 0)) != 0;
#pragma unroll 1
      for (int rnd = 0; rnd < 2; ++rnd) {
        const int d = rnd == 0 ? (inS ? 0 : 1) : 2;
        const bool act = vin && (rnd == 0 || inS);
        const int opd = d == 0 ? HT.op[0] : (d == 1 ? HT.op[1] : HT.op[2]);
        const double gx = geo[opd], gy = geo[3 + opd], gz = geo[6 + opd];
        const int id = d == 0 ? i0 : (d == 1 ? i1 : i2);
        const int o = d == 0 ? i1 + N1 * i2 : (d == 1 ? i0 + N1 * i2 : i0 + N1 * i1);
        const int stride = d == 0 ? 1 : (d == 1 ? N1 : NN);
        int j = id + H;
        j = j >= N1 ? j - N1 : j;
        const int node = act ? lane + (j - id) * stride : lane;
        double qn[7], F[HEX_NFLD];
        const int ns = slot_of(node);
#pragma unroll
        for (int c = 0; c < 7; ++c) qn[c] = sP[c * HW + ns];
        double W = sTab[L.S + (d * N1 + id) * N1 + j] * sTab[L.WT + d * NN + o];
        if (CURVED) {
          W *= .5;
          ec_flux_dir(qv, qn, sM[opd * HW + myslot] + sM[opd * HW + ns], sM[(3 + opd) * HW + myslot] + sM[(3 + opd) * HW + ns],
                      sM[(6 + opd) * HW + myslot] + sM[(6 + opd) * HW + ns], F);
        } else if (DELTA) {
          int a0, a1, a2, b0, b1, b2;
          unpack3(opd == 0 ? kv[0] : (opd == 1 ? kv[1] : kv[2]), a0, a1, a2);
          unpack3(sD[opd * HW + ns], b0, b1, b2);
          ec_flux_dir(qv, qn, __builtin_fma(hsG, (double)(a0 + b0), gx), __builtin_fma(hsG, (double)(a1 + b1), gy),
                      __builtin_fma(hsG, (double)(a2 + b2), gz), F);
        } else {
          ec_flux_dir(qv, qn, gx, gy, gz, F);
        }
        if (act) {
#pragma unroll
          for (int c = 0; c < HEX_NFLD; ++c) {
            const double wf = W * F[c];
            acc[c] += wf;
            lds_add(&sAcc[c * HW + ns], -wf);
          }
        }
      }
    }
  }

  // ---- face lanes: surface flux (:185-198) and the four volume partners of every face node -----------
  // SPLIT (N = 3: 96 face nodes = 1.5 rounds): the last round holds at most half a wave of face nodes, so each of them gets
  // TWO lanes -- lane l and lane l + 32 -- which take half of the node's N1 volume partners each and add their parts of the
  // face total through a cross-lane swap: 5 + 3 flux evaluations per lane instead of 5 + 5 with half the wave idle.
  constexpr int REM = Nfq - HW * (NIT - 1);
  constexpr bool SPLIT = NIT >= 2 && REM <= HW / 2 && N1 >= 2;
  static_assert(!SPLIT || N1 % 2 == 0, "the split face round gives each of a node's two lanes N1 / 2 volume partners");
#pragma unroll 1
  for (int it = 0; it < NIT; ++it) {
    const bool split = SPLIT && it == NIT - 1;                    // uniform
    const int half = split ? (lane >> 5) : 0;
    const int f = split ? HW * it + (lane & 31) : lane + HW * it;
    const bool fin = f < Nfq;
    const int fc = fin ? f : Nfq - 1;
    double qm[7], qp[7];
#pragma unroll
    for (int c = 0; c < HEX_AU_NC; ++c) {
      qm[c] = rm[c];
      qp[c] = rp[c];
    }
    const unsigned kf = kf0, kn = kn0;
    if (it + 1 < NIT) {   // prefetch the next round (the split round: both lanes of a node fetch its records)
      const int f2 = (SPLIT && it + 1 == NIT - 1) ? HW * (it + 1) + (lane & 31) : f + HW;
      const int fc2 = f2 < Nfq ? f2 : Nfq - 1;
      const int64_t nm = ec * Nfq + fc2;
      const int64_t np = M.mapP[nm];
#pragma unroll
      for (int c = 0; c < HEX_AU_NC; ++c) {
        rm[c] = A_U[nm * HEX_AU_NC + c];
        rp[c] = A_U[np * HEX_AU_NC + c];
      }
      if (DELTA) { kf0 = M.hdf[nm]; kn0 = M.hdn[nm]; }
    }
    qm[5] = log_pos(qm[0]);
    qm[6] = log_pos(qm[4]);
    qp[5] = log_pos(qp[0]);
    qp[6] = log_pos(qp[4]);
    const int face = fc / NN;
    double nx = geo[10 + 4 * face], ny = geo[11 + 4 * face], nz = geo[12 + 4 * face], sJ = geo[13 + 4 * face];
    if (CURVED) {
      const double* nr = M.nrm + ec * 4 * Nfq + fc;
      nx = nr[0]; ny = nr[Nfq]; nz = nr[2 * Nfq]; sJ = nr[3 * Nfq];
    }
    if (DELTA) {   // this node's own normal = face mean + scale * packed difference (sJ: the mean; it only scales the LF term)
      int a0, a1, a2;
      unpack3(kn, a0, a1, a2);
      nx = __builtin_fma(sNs, (double)a0, nx); ny = __builtin_fma(sNs, (double)a1, ny); nz = __builtin_fma(sNs, (double)a2, nz);
    }
    double G[HEX_NFLD];
    ec_flux_dir(qm, qp, nx, ny, nz, G);
    if (ph.lf_scale != 0.0) {
      double UM[HEX_NFLD], UP[HEX_NFLD];
      const double isJ = rcp_refined(sJ);
      const double lM = lf_lambda3(qm, nx, ny, nz, isJ, UM);
      const double lP = lf_lambda3(qp, nx, ny, nz, isJ, UP);
      const double LFc = ph.lf_scale * fmax(lM, lP) * sJ;
#pragma unroll
      for (int c = 0; c < HEX_NFLD; ++c) G[c] -= LFc * (UP[c] - UM[c]);
    }
    const double wfac = sTab[L.WFAC + fc];
#pragma unroll
    for (int c = 0; c < HEX_NFLD; ++c) G[c] *= wfac;

    const int code = sInt[L.FINV + fc];
    const int d = code & 3, t = (code >> 2) & 1, o = code >> 3;
    int base, stride;
    line_of<N1>(d, o, base, stride);
    const int opd = d == 0 ? HT.op[0] : (d == 1 ? HT.op[1] : HT.op[2]);
    double gx = geo[opd], gy = geo[3 + opd], gz = geo[6 + opd];
    if (CURVED) {   // this face node's own metric row of direction d (hybrid node Nq + f)
      const double* gm = M.G9 + (ec * 9 + opd) * Nh + Nq + fc;
      gx = gm[0]; gy = gm[3 * Nh]; gz = gm[6 * Nh];
    }
    int f0 = 0, f1 = 0, f2d = 0;
    if (DELTA) unpack3(kf, f0, f1, f2d);   // this face node's metric differences of direction d
    const double wtf = sTab[L.WTF + (d * 2 + t) * NN + o];
    // walk of the line's nodes: position (start + k) mod N1, k = 0, 1, ... (the two ends of a line start at opposite offsets).
    // Split round: the first lane of a node takes k = 0 ... N1/2 - 1 upwards, the second k = N1 - 1 ... N1/2 downwards -- in
    // every step the four (end, lane) combinations of a line are then at four different nodes.
    if (split && half) {
#pragma unroll
      for (int c = 0; c < HEX_NFLD; ++c) G[c] = 0.0;   // the surface term belongs to the first lane's part
    }
    const int kstart = t ? (N1 + 1) / 2 : 0, nsteps = split ? N1 / 2 : N1, kstep = (split && half) ? -1 : 1;
    int k = (split && half) ? N1 - 1 : 0;
#pragma unroll 1
    for (int i = 0; i < nsteps; ++i) {
      int ii = kstart + k;
      ii = ii >= N1 ? ii - N1 : ii;
      const int node = base + ii * stride;
      double qn[7], F[HEX_NFLD];
      const int ns = slot_of(node);
#pragma unroll
      for (int c = 0; c < 7; ++c) qn[c] = sP[c * HW + ns];
      double W = sTab[L.SF + (d * 2 + t) * N1 + ii] * wtf;
      if (CURVED) {
        W *= .5;
        ec_flux_dir(qn, qm, gx + sM[opd * HW + ns], gy + sM[(3 + opd) * HW + ns], gz + sM[(6 + opd) * HW + ns], F);
      } else if (DELTA) {
        int b0, b1, b2;
        unpack3(sD[opd * HW + ns], b0, b1, b2);
        ec_flux_dir(qn, qm, __builtin_fma(hsG, (double)(f0 + b0), gx), __builtin_fma(hsG, (double)(f1 + b1), gy),
                    __builtin_fma(hsG, (double)(f2d + b2), gz), F);
      } else {
        ec_flux_dir(qn, qm, gx, gy, gz, F);
      }
#pragma unroll
      for (int c = 0; c < HEX_NFLD; ++c) {
        const double wf = W * F[c];
        G[c] -= wf;
        if (fin) lds_add(&sAcc[c * HW + ns], wf);
      }
      k += kstep;
    }
    if (split) {   // the two lanes of a node exchange their parts: both hold the face total (a + b, whichever lane adds)
#pragma unroll
      for (int c = 0; c < HEX_NFLD; ++c) G[c] += __shfl_xor(G[c], 32);
    }
    if (fin) {
#pragma unroll
      for (int c = 0; c < HEX_NFLD; ++c) sG[c * Nfq + f] = G[c];
    }
  }
  __syncthreads();
  prio_exit();

  // ---- Ph*QF + Lf*flux, -(.)/J (:198-212) -------------------------------------------------------------
  if (vin) {
    double tot[HEX_NFLD];
    const double pd = sTab[L.PD + lq];
#pragma unroll
    for (int c = 0; c < HEX_NFLD; ++c) tot[c] = pd * (acc[c] + sAcc[c * HW + myslot]);
#pragma unroll
    for (int d = 0; d < 3; ++d) {
      const int id = d == 0 ? i0 : (d == 1 ? i1 : i2);
      const int o = d == 0 ? i1 + N1 * i2 : (d == 1 ? i0 + N1 * i2 : i0 + N1 * i1);
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        const int fi = sInt[L.FN + (d * 2 + t) * NN + o];
        const double w = sTab[L.PF + (d * 2 + t) * N1 + id] * sTab[L.PTF + (d * 2 + t) * NN + o];
#pragma unroll
        for (int c = 0; c < HEX_NFLD; ++c) tot[c] += w * sG[c * Nfq + fi];
      }
    }
    const double miJ = -rcp_refined(CURVED ? M.Jq[ec * Nq + lq] : geo[9]);
    if (active) {
      if (lf.Qw) {   // fused low-storage RK stage (same rounding sequence as k_lsrk); res and Qw are distinct arrays: all
        double ro[HEX_NFLD], qo[HEX_NFLD];   // loads first, then the stores (one round trip instead of ten)
#pragma unroll
        for (int c = 0; c < HEX_NFLD; ++c) { const int64_t idx = (int64_t)c * M.K * Nq + e * Nq + lane; ro[c] = lf.res[idx]; qo[c] = lf.Qw[idx]; }
#pragma unroll
        for (int c = 0; c < HEX_NFLD; ++c) {
          const int64_t idx = (int64_t)c * M.K * Nq + e * Nq + lane;
          const double r = __builtin_fma(lf.a, ro[c], lf.dt * (tot[c] * miJ));
          lf.res[idx] = r;
          lf.Qw[idx] = __builtin_fma(lf.b, r, qo[c]);
        }
      } else {
#pragma unroll
        for (int c = 0; c < HEX_NFLD; ++c) rhs[(int64_t)c * M.K * Nq + e * Nq + lane] = tot[c] * miJ;
      }
    }
  }
}

// =====================================================================================================================
// Degrees N >= 4 (N1 = 5 ... 8: more than one wavefront of Gauss nodes): one WORKGROUP per element, thread t owns the
// volume nodes t, t + T, ... and the face nodes t, t + T, ....  Same tables, trace protocol, geometry modes and formulas as
// kh_project / kh_rhs above (`rhs` and sparse_hadamard_sum of examples/dg3D_euler_hex.jl:122-222 are degree-generic), but
// ROW-WISE like the reference: every node evaluates all pairs of its own row of the hybridised operators (2 x the fluxes
// of the pair-once schedule), so nothing is accumulated across lanes -- no LDS atomics, every sum in program order, results
// bitwise reproducible however the waves of the workgroup are scheduled.  Coverage first: these kernels are not tuned.
// =====================================================================================================================
template <int N1> struct GCfg {
  static constexpr int Nq = N1 * N1 * N1, NN = N1 * N1, Nfq = 6 * NN;
  static constexpr int T = Nq <= 128 ? 128 : 256;                       // threads per workgroup
  static constexpr int NPT = (Nq + T - 1) / T, NFT = (Nfq + T - 1) / T; // volume / face nodes per thread
};

template <int N1>
__global__ __launch_bounds__(GCfg<N1>::T) void kh_project_g(HexTables HT, MeshDev M, const double* __restrict__ Q, double* __restrict__ A_U) {
  using C = GCfg<N1>;
  constexpr HexLayout L(N1);
  constexpr int Nq = C::Nq, Nfq = C::Nfq, T = C::T;
  __shared__ double sTab[L.NDBL];
  __shared__ int sInt[L.NINT];
  __shared__ double sV[HEX_NFLD * Nq];
  const int64_t e = M.e_begin + blockIdx.x;
  if (e >= M.e_begin + M.e_count) return;
  for (int i = threadIdx.x; i < L.NDBL; i += T) sTab[i] = HT.dbl[i];
  for (int i = threadIdx.x; i < L.NINT; i += T) sInt[i] = HT.ints[i];
  for (int n = threadIdx.x; n < Nq; n += T) {
    double U[HEX_NFLD], q[7], V[HEX_NFLD];
#pragma unroll
    for (int c = 0; c < HEX_NFLD; ++c) U[c] = Q[(int64_t)c * M.K * Nq + e * Nq + n];
    prim_logs3(U, q);
    v_of_prim3(q, V);
#pragma unroll
    for (int c = 0; c < HEX_NFLD; ++c) sV[c * Nq + n] = V[c];
  }
  __syncthreads();
  for (int f = threadIdx.x; f < Nfq; f += T) {
    const int code = sInt[L.FINV + f];
    const int d = code & 3, t = (code >> 2) & 1, o = code >> 3;
    int base, stride;
    line_of<N1>(d, o, base, stride);
    double Vf[HEX_NFLD] = {0.0, 0.0, 0.0, 0.0, 0.0};
    for (int i = 0; i < N1; ++i) {
      const double w = sTab[L.EE + (d * 2 + t) * N1 + i];
#pragma unroll
      for (int c = 0; c < HEX_NFLD; ++c) Vf[c] += w * sV[c * Nq + base + i * stride];
    }
    double qf[HEX_NFLD];
    prim_of_v3(Vf, qf);
#pragma unroll
    for (int c = 0; c < HEX_NFLD; ++c) A_U[(e * Nfq + f) * HEX_AU_NC + c] = qf[c];
  }
}

template <int N1, int GM>
__global__ __launch_bounds__(GCfg<N1>::T) void kh_rhs_g(HexTables HT, MeshDev M, Phys ph, const double* __restrict__ Q,
                                                       const double* __restrict__ A_U, double* __restrict__ rhs, LsrkFuse lf) {
  using C = GCfg<N1>;
  constexpr HexLayout L(N1);
  constexpr int Nq = C::Nq, NN = C::NN, Nfq = C::Nfq, T = C::T, Nh = Nq + Nfq;
  constexpr bool CURVED = GM == 1, DELTA = GM == 2;
  __shared__ double sTab[L.NDBL];
  __shared__ int sInt[L.NINT];
  __shared__ double geo[HEX_GEO_STRIDE];
  __shared__ double sP[7 * Nq];                        // (rho,u,v,w,beta,log rho,log beta) of the volume nodes
  __shared__ double sF[7 * Nfq];                       // the same of the own face trace
  __shared__ double sG[HEX_NFLD * Nfq];                // face totals
  __shared__ double sM[CURVED ? 9 * Nh : 1];           // curved: metric terms of all hybrid nodes, [c*3 + op][node]
  __shared__ unsigned sD[DELTA ? 3 * Nq + Nfq : 1];    // packed metric differences: volume nodes [op][node], then face nodes
  const int64_t e = M.e_begin + blockIdx.x;
  if (e >= M.e_begin + M.e_count) return;
  const int tid = threadIdx.x;
  for (int i = tid; i < L.NDBL; i += T) sTab[i] = HT.dbl[i];
  for (int i = tid; i < L.NINT; i += T) sInt[i] = HT.ints[i];
  for (int i = tid; i < HEX_GEO_STRIDE; i += T) geo[i] = M.geo[e * HEX_GEO_STRIDE + i];
  for (int n = tid; n < Nq; n += T) {
    double U[HEX_NFLD], q[7];
#pragma unroll
    for (int c = 0; c < HEX_NFLD; ++c) U[c] = Q[(int64_t)c * M.K * Nq + e * Nq + n];
    prim_logs3(U, q);
#pragma unroll
    for (int c = 0; c < 7; ++c) sP[c * Nq + n] = q[c];
    if (DELTA) {
#pragma unroll
      for (int o3 = 0; o3 < 3; ++o3) sD[o3 * Nq + n] = M.hdv[(e * 3 + o3) * Nq + n];
    }
  }
  if (CURVED)
    for (int i = tid; i < 9 * Nh; i += T) sM[i] = M.G9[e * 9 * Nh + i];
  for (int f = tid; f < Nfq; f += T) {
    const int64_t nm = e * Nfq + f;
    double q[7];
#pragma unroll
    for (int c = 0; c < HEX_AU_NC; ++c) q[c] = A_U[nm * HEX_AU_NC + c];
    q[5] = log_pos(q[0]);
    q[6] = log_pos(q[4]);
#pragma unroll
    for (int c = 0; c < 7; ++c) sF[c * Nfq + f] = q[c];
    if (DELTA) sD[3 * Nq + f] = M.hdf[nm];
  }
  __syncthreads();
  const double hsG = DELTA ? .5 * geo[HEX_GEO_STRIDE - 2] : 0.0, sNs = DELTA ? geo[HEX_GEO_STRIDE - 1] : 0.0;
  // metric vector of the pair (hybrid nodes a, b; b >= Nq: face node b - Nq) for operator family op, as the reference
  // averages it (:145-151); mode 0: the element record
  auto pair_metric = [&](int op, int a, int b, double* g) {
    if (CURVED) {
#pragma unroll
      for (int c = 0; c < 3; ++c) g[c] = .5 * (sM[(3 * c + op) * Nh + a] + sM[(3 * c + op) * Nh + b]);
    } else if (DELTA) {
      int a0, a1, a2, b0, b1, b2;
      unpack3(sD[op * Nq + a], a0, a1, a2);
      unpack3(b < Nq ? sD[op * Nq + b] : sD[3 * Nq + (b - Nq)], b0, b1, b2);
      g[0] = __builtin_fma(hsG, (double)(a0 + b0), geo[op]);
      g[1] = __builtin_fma(hsG, (double)(a1 + b1), geo[3 + op]);
      g[2] = __builtin_fma(hsG, (double)(a2 + b2), geo[6 + op]);
    } else {
      g[0] = geo[op]; g[1] = geo[3 + op]; g[2] = geo[6 + op];
    }
  };

  // ---- face nodes: surface flux (:185-198) minus the volume-face pairs of the node's line ----------------------------
  for (int f = tid; f < Nfq; f += T) {
    const int64_t nm = e * Nfq + f;
    const int64_t np = M.mapP[nm];
    double qm[7], qp[7];
#pragma unroll
    for (int c = 0; c < 7; ++c) qm[c] = sF[c * Nfq + f];
#pragma unroll
    for (int c = 0; c < HEX_AU_NC; ++c) qp[c] = A_U[np * HEX_AU_NC + c];
    qp[5] = log_pos(qp[0]);
    qp[6] = log_pos(qp[4]);
    const int face = f / NN;
    double nx = geo[10 + 4 * face], ny = geo[11 + 4 * face], nz = geo[12 + 4 * face], sJ = geo[13 + 4 * face];
    if (CURVED) {
      const double* nr = M.nrm + e * 4 * Nfq + f;
      nx = nr[0]; ny = nr[Nfq]; nz = nr[2 * Nfq]; sJ = nr[3 * Nfq];
    }
    if (DELTA) {
      int a0, a1, a2;
      unpack3(M.hdn[nm], a0, a1, a2);
      nx = __builtin_fma(sNs, (double)a0, nx); ny = __builtin_fma(sNs, (double)a1, ny); nz = __builtin_fma(sNs, (double)a2, nz);
    }
    double G[HEX_NFLD];
    ec_flux_dir(qm, qp, nx, ny, nz, G);
    if (ph.lf_scale != 0.0) {
      double UM[HEX_NFLD], UP[HEX_NFLD];
      const double isJ = rcp_refined(sJ);
      const double lM = lf_lambda3(qm, nx, ny, nz, isJ, UM);
      const double lP = lf_lambda3(qp, nx, ny, nz, isJ, UP);
      const double LFc = ph.lf_scale * fmax(lM, lP) * sJ;
#pragma unroll
      for (int c = 0; c < HEX_NFLD; ++c) G[c] -= LFc * (UP[c] - UM[c]);
    }
    const double wfac = sTab[L.WFAC + f];
#pragma unroll
    for (int c = 0; c < HEX_NFLD; ++c) G[c] *= wfac;
    const int code = sInt[L.FINV + f];
    const int d = code & 3, t = (code >> 2) & 1, o = code >> 3;
    int base, stride;
    line_of<N1>(d, o, base, stride);
    const int opd = d == 0 ? HT.op[0] : (d == 1 ? HT.op[1] : HT.op[2]);
    const double wtf = sTab[L.WTF + (d * 2 + t) * NN + o];
    for (int i = 0; i < N1; ++i) {
      const int node = base + i * stride;
      double qn[7], F[HEX_NFLD], g[3];
#pragma unroll
      for (int c = 0; c < 7; ++c) qn[c] = sP[c * Nq + node];
      pair_metric(opd, node, Nq + f, g);
      ec_flux_dir(qn, qm, g[0], g[1], g[2], F);
      const double W = sTab[L.SF + (d * 2 + t) * N1 + i] * wtf;
#pragma unroll
      for (int c = 0; c < HEX_NFLD; ++c) G[c] -= W * F[c];
    }
#pragma unroll
    for (int c = 0; c < HEX_NFLD; ++c) sG[c * Nfq + f] = G[c];
  }

  // ---- volume nodes: their own row of the flux differencing (volume and face partners), kept in registers -------------
  double acc[C::NPT][HEX_NFLD];
#pragma unroll
  for (int k = 0; k < C::NPT; ++k) {
    const int n = tid + k * T;
#pragma unroll
    for (int c = 0; c < HEX_NFLD; ++c) acc[k][c] = 0.0;
    if (n >= Nq) continue;
    const int i0 = n % N1, i1 = (n / N1) % N1, i2 = n / NN;
    double qv[7];
#pragma unroll
    for (int c = 0; c < 7; ++c) qv[c] = sP[c * Nq + n];
    for (int d = 0; d < 3; ++d) {
      const int opd = d == 0 ? HT.op[0] : (d == 1 ? HT.op[1] : HT.op[2]);
      const int id = d == 0 ? i0 : (d == 1 ? i1 : i2);
      const int o = d == 0 ? i1 + N1 * i2 : (d == 1 ? i0 + N1 * i2 : i0 + N1 * i1);
      const int stride = d == 0 ? 1 : (d == 1 ? N1 : NN);
      const double wt = sTab[L.WT + d * NN + o];
      for (int j = 0; j < N1; ++j) {
        if (j == id) continue;
        const int node = n + (j - id) * stride;
        double qn[7], F[HEX_NFLD], g[3];
#pragma unroll
        for (int c = 0; c < 7; ++c) qn[c] = sP[c * Nq + node];
        pair_metric(opd, n, node, g);
        ec_flux_dir(qv, qn, g[0], g[1], g[2], F);
        const double W = sTab[L.S + (d * N1 + id) * N1 + j] * wt;
#pragma unroll
        for (int c = 0; c < HEX_NFLD; ++c) acc[k][c] += W * F[c];
      }
      for (int t = 0; t < 2; ++t) {   // the two face nodes at the ends of this line
        const int f = sInt[L.FN + (d * 2 + t) * NN + o];
        double qf[7], F[HEX_NFLD], g[3];
#pragma unroll
        for (int c = 0; c < 7; ++c) qf[c] = sF[c * Nfq + f];
        pair_metric(opd, n, Nq + f, g);
        ec_flux_dir(qv, qf, g[0], g[1], g[2], F);
        const double W = sTab[L.SF + (d * 2 + t) * N1 + id] * sTab[L.WTF + (d * 2 + t) * NN + o];
#pragma unroll
        for (int c = 0; c < HEX_NFLD; ++c) acc[k][c] += W * F[c];
      }
    }
  }
  __syncthreads();

  // ---- Ph*QF + Lf*flux, -(.)/J (:198-212) ------------------------------------------------------------------------------
#pragma unroll
  for (int k = 0; k < C::NPT; ++k) {
    const int n = tid + k * T;
    if (n >= Nq) continue;
    const int i0 = n % N1, i1 = (n / N1) % N1, i2 = n / NN;
    double tot[HEX_NFLD];
    const double pd = sTab[L.PD + n];
#pragma unroll
    for (int c = 0; c < HEX_NFLD; ++c) tot[c] = pd * acc[k][c];
    for (int d = 0; d < 3; ++d) {
      const int id = d == 0 ? i0 : (d == 1 ? i1 : i2);
      const int o = d == 0 ? i1 + N1 * i2 : (d == 1 ? i0 + N1 * i2 : i0 + N1 * i1);
      for (int t = 0; t < 2; ++t) {
        const int fi = sInt[L.FN + (d * 2 + t) * NN + o];
        const double w = sTab[L.PF + (d * 2 + t) * N1 + id] * sTab[L.PTF + (d * 2 + t) * NN + o];
#pragma unroll
        for (int c = 0; c < HEX_NFLD; ++c) tot[c] += w * sG[c * Nfq + fi];
      }
    }
    const double miJ = -rcp_refined(CURVED ? M.Jq[e * Nq + n] : geo[9]);
#pragma unroll
    for (int c = 0; c < HEX_NFLD; ++c) {
      const int64_t idx = (int64_t)c * M.K * Nq + e * Nq + n;
      const double out = tot[c] * miJ;
      if (lf.Qw) {   // fused low-storage RK stage (same rounding sequence as k_lsrk)
        const double r = __builtin_fma(lf.a, lf.res[idx], lf.dt * out);
        lf.res[idx] = r;
        lf.Qw[idx] = __builtin_fma(lf.b, r, lf.Qw[idx]);
      } else {
        rhs[idx] = out;
      }
    }
  }
}

// rhstest = sum(wJq .* v(Q) .* rhs) (dg3D_euler_hex.jl:214-219): per-block partial sums
__global__ void kh_rhstest(int64_t n, const double* __restrict__ wJq, const double* __restrict__ Q,
                           const double* __restrict__ rhs, double* __restrict__ partial) {
  __shared__ double red[256];
  double s = 0.0;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    double U[HEX_NFLD], q[7], V[HEX_NFLD];
#pragma unroll
    for (int c = 0; c < HEX_NFLD; ++c) U[c] = Q[(int64_t)c * n + i];
    prim_logs3(U, q);
    v_of_prim3(q, V);
    double t = 0.0;
#pragma unroll
    for (int c = 0; c < HEX_NFLD; ++c) t += V[c] * rhs[(int64_t)c * n + i];
    s += wJq[i] * t;
  }
  red[threadIdx.x] = s;
  __syncthreads();
  for (int w = 128; w > 0; w >>= 1) {
    if ((int)threadIdx.x < w) red[threadIdx.x] += red[threadIdx.x + w];
    __syncthreads();
  }
  if (threadIdx.x == 0) partial[blockIdx.x] = red[0];
}


// =====================================================================================================================
// Line per lane (round 4; every degree, all three geometry modes).
// kh_rhs_g above is row-wise like the reference: twice the fluxes of a pair-once schedule.  Here, as in kt3_rhs
// (esdg_kernels_tensor3.hip), a lane owns ONE LINE of an element -- its N1 Gauss nodes and the two face nodes at its ends --
// and evaluates each of the line's pairs once (C(N1,2) volume-volume, 2 N1 volume-face, 2 interface fluxes) with the
// N1 + 2 accumulators in registers: no LDS atomics, no row-wise duplicates.  A workgroup holds E elements (three at N = 4, two
// at N = 5, one above: small workgroups overlap their load and compute phases better than full lanes pay -- at N = 4,
// 32x32x16 elements, five elements on 384 threads at 98 % lane use took 0.226 ms, two on 192 threads at 78 % 0.153 ms,
// three on 256 threads at 88 % 0.140 ms); lane l is line l mod (E N1^2) of direction l / (E N1^2).  A node's result is the sum of what its three lines hold for it: direction 0 writes its share to LDS,
// directions 1 and 2 add theirs in turn (three barriers, every sum in program order: bitwise reproducible, and an
// element's result does not depend on the group it sits in).  Node-wise work (primitives + logs, the final scaling and
// store) runs in rounds over all threads.
// =====================================================================================================================
#ifndef ESDG_KHL_E4
#define ESDG_KHL_E4 4
#endif
#ifndef ESDG_KHL_E5
#define ESDG_KHL_E5 3
#endif
#ifndef ESDG_KHL_E6
#define ESDG_KHL_E6 2
#endif
template <int N1> struct LCfg {   // elements per workgroup: its 3 E N1^2 lines fill T = 64 ceil(3 E N1^2 / 64) lanes
  static constexpr int E = N1 == 2 ? 16 : (N1 == 3 ? 7 : (N1 == 4 ? ESDG_KHL_E4 : (N1 == 5 ? ESDG_KHL_E5 : (N1 == 6 ? ESDG_KHL_E6 : 1))));
  static constexpr int NN = N1 * N1, Nq = NN * N1, Nfq = 6 * NN, LLD = E * NN;
  // LDS slot of node i0 + N1 i1 + N1^2 i2: i0 + P (i1 + N1 i2) with an odd pitch P, so that the lanes of every direction --
  // consecutive lines -- read their i-th nodes from distinct banks (measured at N1 = 4 without the padding: 60 % of the LDS
  // cycles of the kernel were bank conflicts, the direction-0 lanes four to a bank)
  static constexpr int P = N1 % 2 == 0 ? N1 + 1 : N1, NQP = P * NN, NV = E * NQP, NVN = E * Nq;
  static constexpr int T = ((3 * LLD + HW - 1) / HW) * HW;
  static constexpr int NRN = (NVN + T - 1) / T;
};
// slot base and stride of the line o of direction d (nodes: line_of above)
template <int N1>
__device__ __forceinline__ void line_slots(int d, int o, int& base, int& stride) {
  constexpr int P = LCfg<N1>::P;
  stride = d == 0 ? 1 : (d == 1 ? P : P * N1);
  base = d == 0 ? P * o : (d == 1 ? (o % N1) + P * N1 * (o / N1) : (o % N1) + P * (o / N1));
}

// (compiler fences of the line stage, see T3_FENCE / T3_PIN4 in esdg_kernels_tensor3.hip)
#define KHL_FENCE() do { __builtin_amdgcn_sched_barrier(0); asm volatile("" ::: "memory"); } while (0)
#define KHL_PIN5(a) asm volatile("" : "+v"((a)[0]), "+v"((a)[1]), "+v"((a)[2]), "+v"((a)[3]), "+v"((a)[4]))

// STG: the DOPRI45 stage fusion of kt3_rhs (StageFuse, esdg_dev.hpp) for the hexahedral path: the node rounds also form the next
// stage's state / the error norm from the value they hold.  Q and rhs are not __restrict__: the fused updates write the state
// the launch read (lf.Qw, sf.y) -- every read of Q lies before the barrier that precedes the node rounds.
template <int N1, int GM, bool STG = false>
__global__ __launch_bounds__(LCfg<N1>::T) void kh_rhs_l(HexTables HT, MeshDev M, Phys ph, const double* Q,
                                                       const double* __restrict__ A_U, double* rhs, LsrkFuse lf, StageFuse sf) {
  using C = LCfg<N1>;
  constexpr HexLayout L(N1);
  constexpr int E = C::E, NN = C::NN, Nq = C::Nq, Nfq = C::Nfq, T = C::T, NV = C::NV, NVN = C::NVN, NQP = C::NQP, LLD = C::LLD, NRN = C::NRN;
  constexpr bool DELTA = GM == 2, CURVED = GM == 1;   // (geometry modes: see kh_rhs)
  constexpr int ND = (L.NDBL + T - 1) / T, NI = (L.NINT + T - 1) / T, NG = (E * HEX_GEO_STRIDE + T - 1) / T;
  typedef double2 d2;
  __shared__ __align__(16) double sTab[ND * T];
  __shared__ int sInt[NI * T];
  __shared__ double sGeo[NG * T];
  __shared__ __align__(16) double arena[8 * NV];                 // records: 4 pair planes [NV]; later the results [5][NV]
  __shared__ unsigned sD[DELTA ? 3 * NV : 1];                    // packed metric differences of the volume nodes, [op][slot]
  d2* sP = reinterpret_cast<d2*>(arena);                         // (rho,u) (v,w) (beta,log rho) (log beta, -)
  double* sR = arena;

  const int tid = threadIdx.x;
  const int64_t e0 = M.e_begin + (int64_t)blockIdx.x * E;
  const int nE = (int)min((int64_t)E, M.e_begin + M.e_count - e0);
  const int64_t KN = M.K * Nq;

  // ---- this lane's line ----------------------------------------------------------------------------------------------
  const bool lact = tid < 3 * LLD;                               // (lanes beyond the lines redo line 0: nothing of theirs is written)
  const int gl = lact ? tid : 0;
  const int d = gl / LLD, llc = gl - d * LLD;                    // direction, line within the direction
  const int el = llc / NN, o = llc - el * NN;
  const int elc = el < nE ? el : 0;                              // (elements beyond the range: the data of the first one)
  const int64_t ec = e0 + elc;
  int base, stride, sbase, sstride;
  line_of<N1>(d, o, base, stride);          // nodes base + i stride (for the per-node table PD)
  line_slots<N1>(d, o, sbase, sstride);     // their LDS slots n0 + i sstride
  const int n0 = el * NQP + sbase;
  const int opd = d == 0 ? HT.op[0] : (d == 1 ? HT.op[1] : HT.op[2]);
  const int fA = HT.ints[L.FN + (d * 2) * NN + o], fB = HT.ints[L.FN + (d * 2 + 1) * NN + o];
  const int64_t nmA = ec * Nfq + fA, nmB = ec * Nfq + fB;
  const int64_t npA = M.mapP[nmA], npB = M.mapP[nmB];

  // ---- global loads, unconditionally --------------------------------------------------------------------------------
  double tabd[ND], geo_r[NG], U[NRN][HEX_NFLD];
  int tabi[NI];
  unsigned kvn[NRN][3];
#pragma unroll
  for (int r = 0; r < ND; ++r) tabd[r] = HT.dbl[min(tid + r * T, L.NDBL - 1)];
#pragma unroll
  for (int r = 0; r < NI; ++r) tabi[r] = HT.ints[min(tid + r * T, L.NINT - 1)];
#pragma unroll
  for (int r = 0; r < NG; ++r) geo_r[r] = M.geo[e0 * HEX_GEO_STRIDE + min(tid + r * T, nE * HEX_GEO_STRIDE - 1)];
#pragma unroll
  for (int r = 0; r < NRN; ++r) {
    const int n = tid + r * T, nc = n < nE * Nq ? n : 0;
#pragma unroll
    for (int c = 0; c < HEX_NFLD; ++c) U[r][c] = Q[(int64_t)c * KN + e0 * Nq + nc];
    if (DELTA) {
      const int en = nc / Nq, q = nc - en * Nq;
#pragma unroll
      for (int o3 = 0; o3 < 3; ++o3) kvn[r][o3] = M.hdv[((e0 + en) * 3 + o3) * Nq + q];
    }
  }
  double rmA[HEX_AU_NC], rpA[HEX_AU_NC], rmB[HEX_AU_NC], rpB[HEX_AU_NC];
#pragma unroll
  for (int c = 0; c < HEX_AU_NC; ++c) {
    rmA[c] = A_U[nmA * HEX_AU_NC + c]; rpA[c] = A_U[npA * HEX_AU_NC + c];
    rmB[c] = A_U[nmB * HEX_AU_NC + c]; rpB[c] = A_U[npB * HEX_AU_NC + c];
  }
  unsigned kfA = 0u, knA = 0u, kfB = 0u, knB = 0u;
  if (DELTA) { kfA = M.hdf[nmA]; knA = M.hdn[nmA]; kfB = M.hdf[nmB]; knB = M.hdn[nmB]; }
  // curved meshes: the metric row of this line's operator at its N1 nodes and two face nodes (x, y, z components), the face nodes'
  // own normals and sJ.  A node's row of operator op(d) is read by its direction-d line only, so nothing is shared between lanes.
  constexpr int Nh = Nq + Nfq;
  double Gl[CURVED ? N1 : 1][3], gfA[3] = {0, 0, 0}, gfB[3] = {0, 0, 0}, nA[4] = {0, 0, 0, 0}, nB[4] = {0, 0, 0, 0};
  if (CURVED) {
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const double* row = M.G9 + (ec * 9 + c * 3 + opd) * Nh;
#pragma unroll
      for (int i = 0; i < N1; ++i) Gl[i][c] = row[base + i * stride];
      gfA[c] = row[Nq + fA]; gfB[c] = row[Nq + fB];
    }
#pragma unroll
    for (int c = 0; c < 4; ++c) { nA[c] = M.nrm[ec * 4 * Nfq + c * Nfq + fA]; nB[c] = M.nrm[ec * 4 * Nfq + c * Nfq + fB]; }
  }

  // ---- staging; node rounds: primitives + logs -> records ---------------------------------------------------------------
#pragma unroll
  for (int r = 0; r < ND; ++r) sTab[tid + r * T] = tabd[r];
#pragma unroll
  for (int r = 0; r < NI; ++r) sInt[tid + r * T] = tabi[r];
#pragma unroll
  for (int r = 0; r < NG; ++r) sGeo[min(tid + r * T, E * HEX_GEO_STRIDE - 1)] = geo_r[r];
#pragma unroll
  for (int r = 0; r < NRN; ++r) {
    const int n = tid + r * T;
    double qv[7];
    prim_logs3(U[r], qv);
    if (n < NVN) {
      const int en = n / Nq, q = n - en * Nq, sl = en * NQP + q + (C::P != N1 ? q / N1 : 0);
      sP[sl] = make_double2(qv[0], qv[1]); sP[NV + sl] = make_double2(qv[2], qv[3]);
      sP[2 * NV + sl] = make_double2(qv[4], qv[5]); sP[3 * NV + sl] = make_double2(qv[6], 0.0);
      if (DELTA) {
#pragma unroll
        for (int o3 = 0; o3 < 3; ++o3) sD[o3 * NV + sl] = kvn[r][o3];
      }
    }
  }
  __syncthreads();

  // ---- line stage ------------------------------------------------------------------------------------------------------------
  const double* geo = sGeo + elc * HEX_GEO_STRIDE;
  // (the flux takes HALF the metric vector, ec_flux_core3: halves of the record's row and of the difference scale)
  const double hgx = .5 * geo[opd], hgy = .5 * geo[3 + opd], hgz = .5 * geo[6 + opd];
  const double qsG = DELTA ? .25 * geo[HEX_GEO_STRIDE - 2] : 0.0, sNs = DELTA ? geo[HEX_GEO_STRIDE - 1] : 0.0;
  const t2::SeriesK sk = t2::series_k_pinned();
  double acc[N1][HEX_NFLD], GA[HEX_NFLD], GB[HEX_NFLD];
#pragma unroll
  for (int i = 0; i < N1; ++i)
#pragma unroll
    for (int c = 0; c < HEX_NFLD; ++c) acc[i][c] = 0.0;
  auto record = [&](int slot, double* q) {
    const d2 p0 = sP[slot], p1 = sP[NV + slot], p2 = sP[2 * NV + slot], p3 = sP[3 * NV + slot];
    q[0] = p0.x; q[1] = p0.y; q[2] = p1.x; q[3] = p1.y; q[4] = p2.x; q[5] = p2.y; q[6] = p3.x;
  };
  // one face turn: surface flux of face node f (end t of the line), then its N1 volume-face pairs
  auto face_turn = [&](int t, int f, const double* rm, const double* rp, unsigned kf, unsigned kn, const double* gfc, const double* nc, double* G) {
    double qm[7], qp[7];
#pragma unroll
    for (int c = 0; c < HEX_AU_NC; ++c) { qm[c] = rm[c]; qp[c] = rp[c]; }
    qm[5] = log_pos(qm[0]); qm[6] = log_pos(qm[4]);
    qp[5] = log_pos(qp[0]); qp[6] = log_pos(qp[4]);
    const int face = f / NN;
    double nx = geo[10 + 4 * face], ny = geo[11 + 4 * face], nz = geo[12 + 4 * face];
    double sJ = geo[13 + 4 * face];
    if (CURVED) { nx = nc[0]; ny = nc[1]; nz = nc[2]; sJ = nc[3]; }
    if (DELTA) {   // this node's own normal = face mean + scale * packed difference
      int a0, a1, a2;
      unpack3(kn, a0, a1, a2);
      nx = __builtin_fma(sNs, (double)a0, nx); ny = __builtin_fma(sNs, (double)a1, ny); nz = __builtin_fma(sNs, (double)a2, nz);
    }
    ec_flux_dir_u(qm, qp, .5 * nx, .5 * ny, .5 * nz, G, sk);
    if (ph.lf_scale != 0.0) {   // (uniform)
      double UM[HEX_NFLD], UP[HEX_NFLD];
      const double isJ = rcp_refined(sJ);
      const double lM = lf_lambda3(qm, nx, ny, nz, isJ, UM);
      const double lP = lf_lambda3(qp, nx, ny, nz, isJ, UP);
      const double LFc = ph.lf_scale * fmax(lM, lP) * sJ;
#pragma unroll
      for (int c = 0; c < HEX_NFLD; ++c) G[c] -= LFc * (UP[c] - UM[c]);
    }
    const double wfac = sTab[L.WFAC + f];
#pragma unroll
    for (int c = 0; c < HEX_NFLD; ++c) G[c] *= wfac;
    int f0 = 0, f1 = 0, f2 = 0;
    if (DELTA) unpack3(kf, f0, f1, f2);   // this face node's metric differences of direction d
    const double wtf = sTab[L.WTF + (d * 2 + t) * NN + o];
#pragma unroll
    for (int i = 0; i < N1; ++i) {
      const int slot = n0 + i * sstride;
      KHL_FENCE();
      double qn[7], F[HEX_NFLD];
      record(slot, qn);
      double W = sTab[L.SF + (d * 2 + t) * N1 + i] * wtf;
      if (CURVED) {   // metric of the pair = average of the two nodes (dg3D_euler_hex.jl:145-151); the sum goes in as the HALF metric,
        W *= .25;     // i.e. four times the half average: the factor rides in the weight
        ec_flux_dir_u(qn, qm, gfc[0] + Gl[i][0], gfc[1] + Gl[i][1], gfc[2] + Gl[i][2], F, sk);
      } else if (DELTA) {
        int b0, b1, b2;
        unpack3(sD[opd * NV + slot], b0, b1, b2);
        ec_flux_dir_u(qn, qm, __builtin_fma(qsG, (double)(f0 + b0), hgx), __builtin_fma(qsG, (double)(f1 + b1), hgy),
                    __builtin_fma(qsG, (double)(f2 + b2), hgz), F, sk);
      } else {
        ec_flux_dir_u(qn, qm, hgx, hgy, hgz, F, sk);
      }
#pragma unroll
      for (int c = 0; c < HEX_NFLD; ++c) { acc[i][c] = __builtin_fma(W, F[c], acc[i][c]); G[c] = __builtin_fma(-W, F[c], G[c]); }
      KHL_PIN5(acc[i]); KHL_PIN5(G);
    }
  };
  face_turn(0, fA, rmA, rpA, kfA, knA, gfA, nA, GA);
  face_turn(1, fB, rmB, rpB, kfB, knB, gfB, nB, GB);
  {   // volume-volume pairs of the line, each once
    const double wt = sTab[L.WT + d * NN + o];
#pragma unroll
    for (int i = 0; i < N1 - 1; ++i) {
      const int si = n0 + i * sstride;
      KHL_FENCE();
      double qi[7];
      record(si, qi);
      int a0 = 0, a1 = 0, a2 = 0;
      if (DELTA) unpack3(sD[opd * NV + si], a0, a1, a2);
#pragma unroll
      for (int j = i + 1; j < N1; ++j) {
        const int sj = n0 + j * sstride;
        KHL_FENCE();
        double qj[7], F[HEX_NFLD];
        record(sj, qj);
        double W = sTab[L.S + (d * N1 + i) * N1 + j] * wt;
        if (CURVED) {
          W *= .25;
          ec_flux_dir_u(qi, qj, Gl[i][0] + Gl[j][0], Gl[i][1] + Gl[j][1], Gl[i][2] + Gl[j][2], F, sk);
        } else if (DELTA) {   // metric of the pair = average of the two nodes: record + half the scale times the two differences
          int b0, b1, b2;
          unpack3(sD[opd * NV + sj], b0, b1, b2);
          ec_flux_dir_u(qi, qj, __builtin_fma(qsG, (double)(a0 + b0), hgx), __builtin_fma(qsG, (double)(a1 + b1), hgy),
                      __builtin_fma(qsG, (double)(a2 + b2), hgz), F, sk);
        } else {
          ec_flux_dir_u(qi, qj, hgx, hgy, hgz, F, sk);
        }
#pragma unroll
        for (int c = 0; c < HEX_NFLD; ++c) { acc[i][c] = __builtin_fma(W, F[c], acc[i][c]); acc[j][c] = __builtin_fma(-W, F[c], acc[j][c]); }
        KHL_PIN5(acc[i]); KHL_PIN5(acc[j]);
      }
    }
  }
  // ---- the line's share of Ph*QF + Lf*flux at its nodes; the three directions' shares added in turn ----------------------
  {
    const double ptA = sTab[L.PTF + (d * 2) * NN + o], ptB = sTab[L.PTF + (d * 2 + 1) * NN + o];
#pragma unroll
    for (int i = 0; i < N1; ++i) {
      const double pd = sTab[L.PD + base + i * stride];
      const double pa = sTab[L.PF + (d * 2) * N1 + i] * ptA, pb = sTab[L.PF + (d * 2 + 1) * N1 + i] * ptB;
#pragma unroll
      for (int c = 0; c < HEX_NFLD; ++c) acc[i][c] = __builtin_fma(pb, GB[c], __builtin_fma(pa, GA[c], pd * acc[i][c]));
    }
  }
  __syncthreads();   // every lane is past its reads of the records, whose space takes the results
#pragma unroll 1
  for (int step = 0; step < 3; ++step) {
    if (d == step && lact) {
#pragma unroll
      for (int i = 0; i < N1; ++i) {
        const int slot = n0 + i * sstride;
#pragma unroll
        for (int c = 0; c < HEX_NFLD; ++c) sR[c * NV + slot] = step == 0 ? acc[i][c] : sR[c * NV + slot] + acc[i][c];
      }
    }
    __syncthreads();
  }
  // ---- node rounds: -(.)/J, store or fused low-storage RK stage (:198-212) ------------------------------------------------
  asm volatile("" ::: "memory");   // (no load of the state sinks below this point: the fused updates may overwrite it in place)
#pragma unroll
  for (int r = 0; r < NRN; ++r) {
    const int n = tid + r * T;
    if (n < nE * Nq) {
      const int en = n / Nq, q = n - en * Nq, sl = en * NQP + q + (C::P != N1 ? q / N1 : 0);
      const double miJ = -rcp_refined(CURVED ? M.Jq[e0 * Nq + n] : sGeo[en * HEX_GEO_STRIDE + 9]);
      if (STG) {   // DOPRI45: the store of k_s plus the next stage's state / the error norm -- the fma chains of kt3_rhs's STG epilogue
        const int64_t i0 = e0 * Nq + n;
        double out[HEX_NFLD];
#pragma unroll
        for (int c = 0; c < HEX_NFLD; ++c) out[c] = sR[c * NV + sl] * miJ;
        if (sf.y) {   // (uniform)
          double xo[HEX_NFLD], kk[6][HEX_NFLD];
#pragma unroll
          for (int c = 0; c < HEX_NFLD; ++c) xo[c] = sf.x0[c * KN + i0];
#pragma unroll
          for (int j = 0; j < 6; ++j)
            if (j < sf.ns) {   // (uniform)
#pragma unroll
              for (int c = 0; c < HEX_NFLD; ++c) kk[j][c] = sf.k[j][c * KN + i0];
            }
#pragma unroll
          for (int c = 0; c < HEX_NFLD; ++c) rhs[c * KN + i0] = out[c];
#pragma unroll
          for (int c = 0; c < HEX_NFLD; ++c) {
            double a = 0.0, ee = 0.0;
#pragma unroll
            for (int j = 0; j < 6; ++j)
              if (j < sf.ns) { a = __builtin_fma(sf.c[j], kk[j][c], a); ee = __builtin_fma(sf.ce[j], kk[j][c], ee); }
            a = __builtin_fma(sf.c_last, out[c], a);
            sf.y[c * KN + i0] = __builtin_fma(sf.dt, a, xo[c]);
            if (sf.e_out) sf.e_out[c * KN + i0] = __builtin_fma(sf.ce_last, out[c], ee);
          }
        } else {
          double xo[HEX_NFLD], ei[HEX_NFLD];
#pragma unroll
          for (int c = 0; c < HEX_NFLD; ++c) { xo[c] = sf.x0[c * KN + i0]; ei[c] = sf.err ? rhs[c * KN + i0] : 0.0; }
          double t = 0.0;   // this node's term (k_dopri_err's chain) at the node's own index: the host adds them in ONE order (k_chunk_sum)
#pragma unroll
          for (int c = 0; c < HEX_NFLD; ++c) {
            rhs[c * KN + i0] = out[c];
            const double e = __builtin_fma(sf.ce_last, out[c], ei[c]);
            const double sc = fabs(e) / (sf.tol * (1 + fabs(xo[c])));
            t = __builtin_fma(sc, sc, t);
          }
          if (sf.err) sf.partial[i0] = t;   // (uniform)
        }
        continue;
      }
      if (lf.Qw) {   // (uniform) fused low-storage RK stage, same rounding sequence as k_lsrk; res and Qw are distinct arrays: all
        double ro[HEX_NFLD], qo[HEX_NFLD];   // loads first, then the stores (one round trip instead of ten)
#pragma unroll
        for (int c = 0; c < HEX_NFLD; ++c) { const int64_t idx = (int64_t)c * KN + e0 * Nq + n; ro[c] = lf.res[idx]; qo[c] = lf.Qw[idx]; }
#pragma unroll
        for (int c = 0; c < HEX_NFLD; ++c) {
          const int64_t idx = (int64_t)c * KN + e0 * Nq + n;
          const double rr = __builtin_fma(lf.a, ro[c], lf.dt * (sR[c * NV + sl] * miJ));
          lf.res[idx] = rr;
          lf.Qw[idx] = __builtin_fma(lf.b, rr, qo[c]);
        }
      } else {
#pragma unroll
        for (int c = 0; c < HEX_NFLD; ++c) rhs[(int64_t)c * KN + e0 * Nq + n] = sR[c * NV + sl] * miJ;
      }
    }
  }
}

}  // namespace hdev

// N1 = 2 ... 4: one wavefront per element (kh_project / kh_rhs); N1 = 5 ... 8: one workgroup per element (kh_*_g)
bool hex_supported_degree(int N1) { return N1 >= 2 && N1 <= 11; }

#define ESDG_HEX_DISPATCH(N1v, STMT)   \
  switch (N1v) {                        \
    case 2: { constexpr int N1 = 2; STMT; } break; \
    case 3: { constexpr int N1 = 3; STMT; } break; \
    case 4: { constexpr int N1 = 4; STMT; } break; \
    default: return (int)hipErrorInvalidValue;     \
  }
#define ESDG_HEXG_DISPATCH(N1v, STMT)  \
  switch (N1v) {                        \
    case 5: { constexpr int N1 = 5; STMT; } break; \
    case 6: { constexpr int N1 = 6; STMT; } break; \
    case 7: { constexpr int N1 = 7; STMT; } break; \
    case 8: { constexpr int N1 = 8; STMT; } break; \
    default: return (int)hipErrorInvalidValue;     \
  }
// (phase 0 goes two degrees further than the row-wise kh_rhs_g, whose curved instantiation would need 200 KB of LDS at N1 = 9: the
// last phase of N1 = 9, 10 is kh_rhs_l only)
#define ESDG_HEXP_DISPATCH(N1v, STMT)  \
  switch (N1v) {                        \
    case 5: { constexpr int N1 = 5; STMT; } break; \
    case 6: { constexpr int N1 = 6; STMT; } break; \
    case 7: { constexpr int N1 = 7; STMT; } break; \
    case 8: { constexpr int N1 = 8; STMT; } break; \
    case 9: { constexpr int N1 = 9; STMT; } break; \
    case 10: { constexpr int N1 = 10; STMT; } break; \
    case 11: { constexpr int N1 = 11; STMT; } break; \
    default: return (int)hipErrorInvalidValue;     \
  }

static inline unsigned hex_grid(int64_t K, bool remap) {
  const int64_t nblk = (K + hdev::HNWV - 1) / hdev::HNWV;
  if (!remap) return (unsigned)nblk;
  return (unsigned)(((nblk + hdev::NXCD - 1) / hdev::NXCD) * hdev::NXCD);
}

int launch_project_hex(int N1v, const HexTables& HT, const MeshDev& M, const Phys& ph, const double* Q, double* A_U,
                       hipStream_t s) {
  if (M.e_count <= 0) return 0;
  const int remap = (ph.dbg & 16) ? 0 : 1;
  if (N1v > 4) {
    ESDG_HEXP_DISPATCH(N1v, hipLaunchKernelGGL((hdev::kh_project_g<N1>), dim3((unsigned)M.e_count), dim3(hdev::GCfg<N1>::T), 0, s, HT, M, Q, A_U));
    return (int)hipGetLastError();
  }
  ESDG_HEX_DISPATCH(N1v, hipLaunchKernelGGL((hdev::kh_project<N1>), dim3(hex_grid(M.e_count, remap)), dim3(hdev::HW * hdev::HNWV), 0, s, HT, M, remap, Q, A_U));
  return (int)hipGetLastError();
}

static int g_hex_line = 1;   // kh_rhs_l; 0 = kh_rhs / kh_rhs_g (A/B builds: esdg_api.hip under -DESDG_AB_HOOKS, ESDG_HEX_LINE=0)
void ab_tuning_hex(int line) { g_hex_line = line; }

// workgroups of a last-phase launch over e_count elements; -1: the line kernel is off
int rhs_hex_blocks(int N1v, int64_t e_count) {
  if (g_hex_line == 0) return -1;
  switch (N1v) {
#define ESDG_HEXL_BLOCKS(N1c) case N1c: return (int)((e_count + hdev::LCfg<N1c>::E - 1) / hdev::LCfg<N1c>::E);
    ESDG_HEXL_BLOCKS(2) ESDG_HEXL_BLOCKS(3) ESDG_HEXL_BLOCKS(4) ESDG_HEXL_BLOCKS(5) ESDG_HEXL_BLOCKS(6) ESDG_HEXL_BLOCKS(7) ESDG_HEXL_BLOCKS(8) ESDG_HEXL_BLOCKS(9) ESDG_HEXL_BLOCKS(10) ESDG_HEXL_BLOCKS(11)
#undef ESDG_HEXL_BLOCKS
    default: return -1;
  }
}

int launch_rhs_hex(int N1v, const HexTables& HT, const MeshDev& M, const Phys& ph, const double* Q, const double* A_U,
                   double* rhs, const LsrkFuse& lf, hipStream_t s, const StageFuse* sf) {
  if (M.e_count <= 0) return 0;
  if (sf && g_hex_line == 0) return -1;   // (DOPRI45 stage fusion: the line kernel only)
  const StageFuse sf0{};
  const int remap = (ph.dbg & 16) ? 0 : 1;
  // The line-per-lane kernel kh_rhs_l at every degree and in every geometry mode (round 4: N = 3 128x128x16 1.03 vs
  // 1.23 ms for kh_rhs; N = 1, 2: 0.65, 0.83 x kh_rhs; N = 4 ... 7: 0.47 ... 0.74 x the row-wise kh_rhs_g);
  // ESDG_HEX_LINE=0: kh_rhs / kh_rhs_g (A/B).
  {
    if (g_hex_line != 0) {   // (0 only in A/B builds: ab_tuning_hex)
#define ESDG_HEXL_LAUNCH(N1c)                                                                                                 \
  case N1c: {                                                                                                                \
    const dim3 grid((unsigned)((M.e_count + hdev::LCfg<N1c>::E - 1) / hdev::LCfg<N1c>::E)), blk(hdev::LCfg<N1c>::T);         \
    if (sf) {                                                                                                                \
      if (M.G9) hipLaunchKernelGGL((hdev::kh_rhs_l<N1c, 1, true>), grid, blk, 0, s, HT, M, ph, Q, A_U, rhs, lf, *sf);         \
      else if (M.hdv) hipLaunchKernelGGL((hdev::kh_rhs_l<N1c, 2, true>), grid, blk, 0, s, HT, M, ph, Q, A_U, rhs, lf, *sf);   \
      else hipLaunchKernelGGL((hdev::kh_rhs_l<N1c, 0, true>), grid, blk, 0, s, HT, M, ph, Q, A_U, rhs, lf, *sf);               \
    } else if (M.G9) hipLaunchKernelGGL((hdev::kh_rhs_l<N1c, 1>), grid, blk, 0, s, HT, M, ph, Q, A_U, rhs, lf, sf0);         \
    else if (M.hdv) hipLaunchKernelGGL((hdev::kh_rhs_l<N1c, 2>), grid, blk, 0, s, HT, M, ph, Q, A_U, rhs, lf, sf0);          \
    else hipLaunchKernelGGL((hdev::kh_rhs_l<N1c, 0>), grid, blk, 0, s, HT, M, ph, Q, A_U, rhs, lf, sf0);                      \
  } break;
      switch (N1v) {
        ESDG_HEXL_LAUNCH(2) ESDG_HEXL_LAUNCH(3) ESDG_HEXL_LAUNCH(4) ESDG_HEXL_LAUNCH(5) ESDG_HEXL_LAUNCH(6) ESDG_HEXL_LAUNCH(7) ESDG_HEXL_LAUNCH(8) ESDG_HEXL_LAUNCH(9) ESDG_HEXL_LAUNCH(10) ESDG_HEXL_LAUNCH(11)
        default: return (int)hipErrorInvalidValue;
      }
#undef ESDG_HEXL_LAUNCH
      return (int)hipGetLastError();
    }
  }
  if (N1v > 4) {
    const dim3 grid((unsigned)M.e_count);
    if (M.G9) { ESDG_HEXG_DISPATCH(N1v, hipLaunchKernelGGL((hdev::kh_rhs_g<N1, 1>), grid, dim3(hdev::GCfg<N1>::T), 0, s, HT, M, ph, Q, A_U, rhs, lf)); }
    else if (M.hdv) { ESDG_HEXG_DISPATCH(N1v, hipLaunchKernelGGL((hdev::kh_rhs_g<N1, 2>), grid, dim3(hdev::GCfg<N1>::T), 0, s, HT, M, ph, Q, A_U, rhs, lf)); }
    else { ESDG_HEXG_DISPATCH(N1v, hipLaunchKernelGGL((hdev::kh_rhs_g<N1, 0>), grid, dim3(hdev::GCfg<N1>::T), 0, s, HT, M, ph, Q, A_U, rhs, lf)); }
    return (int)hipGetLastError();
  }
  if (M.G9) {
    ESDG_HEX_DISPATCH(N1v, hipLaunchKernelGGL((hdev::kh_rhs<N1, 1>), dim3(hex_grid(M.e_count, remap)), dim3(hdev::HW * hdev::HNWV), 0, s, HT, M, ph, remap, Q, A_U, rhs, lf));
  } else if (M.hdv) {
    ESDG_HEX_DISPATCH(N1v, hipLaunchKernelGGL((hdev::kh_rhs<N1, 2>), dim3(hex_grid(M.e_count, remap)), dim3(hdev::HW * hdev::HNWV), 0, s, HT, M, ph, remap, Q, A_U, rhs, lf));
  } else {
    ESDG_HEX_DISPATCH(N1v, hipLaunchKernelGGL((hdev::kh_rhs<N1, 0>), dim3(hex_grid(M.e_count, remap)), dim3(hdev::HW * hdev::HNWV), 0, s, HT, M, ph, remap, Q, A_U, rhs, lf));
  }
  return (int)hipGetLastError();
}

int launch_rhstest_hex(int64_t n, const double* wJq, const double* Q, const double* rhs, double* partial, int nblocks,
                       hipStream_t s) {
  hipLaunchKernelGGL(hdev::kh_rhstest, dim3(nblocks), dim3(256), 0, s, n, wJq, Q, rhs, partial);
  return (int)hipGetLastError();
}

}  // namespace esdg
