// esdg_dev.hpp -- device-side data structures shared by the host API (esdg_api.hip) and the
// gfx950 kernels (esdg_kernels.hip).  MI355X-only code: no CUDA paths, wave = 64.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace esdg {

// Per-element affine geometry record (doubles):
//   [0..3] rxJ sxJ ryJ syJ   (row 1 of the Vh-interpolated metric arrays, dg2D_euler_quad.jl:175)
//   [4]    J
//   [5+3f .. 7+3f] nxJ nyJ sJ of face f (first node of the face; constant on affine faces)
constexpr int GEO_STRIDE = 17;

constexpr int AU_NC = 5;  // trace buffer A_U: (rho, rho*u, rho*v, E, lam) per face node
constexpr int AV_NC = 3;  // trace buffer A_v: projected entropy variables (v2, v3, v4)
constexpr int B_NC = 3;   // trace buffer B: normal viscous stress (sigma_x*nxJ + sigma_y*nyJ), rows 2..4

// Sparse (ELL) collocated operators and the flux-differencing pair list, derived once on the
// host from the dense matrices the driver passes (esdg_ops_t).  All pointers are device memory.
struct Tables {
  int N1, Np, Nq, Nfq, Nh;
  int P;                   // number of unordered flux pairs with a non-zero weight
  const uint8_t* pair_ij;  // [P][2]  (i, j) node ids in the hybridized numbering (vol, then face)
  const double* pair_c;    // [P][2]  (Qrhskew[i,j], Qshskew[i,j])
  const uint16_t* inc_ptr; // [Nh+1]  CSR over rows: incident pairs
  const uint16_t* inc;     // [2P]    pair id | 0x8000 if this row is the pair's j (subtract)
  const uint8_t* Ef_idx;   // [Nfq][wEf]
  const double* Ef_val;
  const uint8_t* Ph_idx;   // [Nq][wPh]  columns in 0..Nh-1
  const double* Ph_val;
  const uint8_t* Lf_idx;   // [Nq][wLf]  columns in 0..Nfq-1
  const double* Lf_val;
  const uint8_t* Dr_idx;   // [Nq][wD]
  const double* Dr_val;
  const uint8_t* Ds_idx;
  const double* Ds_val;
  int wEf, wPh, wLf, wD;
  const double* Vq;        // [Nq][Np] row-major (modal)
  const double* Pq;        // [Np][Nq] row-major (modal)
};

// Trace of the tensor kernels: (rho, u, v, beta) of the entropy-projected face state, one 32-B record per face node.  Its logs,
// energy and LF wavespeed -- a second 32-B record until round 3 -- are rebuilt by the consumer (devmath::trace_rest): phase 0
// writes and exchanges half the bytes, the last phase reads 1280 B per element less and pays ~240 VALU instructions.
constexpr int FAU_NC = 4;

// Largest N1 = N + 1 of the 2D tensor kernels (kt2_project, kt2_sigma, kt3_rhs; kt2_rhs stops at 9).  The generic pair-list
// kernels stop at N1 = 8, the hexahedral kernels at 10 (the row-wise kh_rhs_g at 8).
#ifndef ESDG_MAX_N1
#define ESDG_MAX_N1 12
#endif

struct TensorTables;
struct MeshDev;
struct Phys;
// v2 tensor kernels (esdg_kernels_tensor2.hip)
struct LsrkFuse;
int launch_rhs_tensor2(int N1, const TensorTables& TT, const MeshDev& M, const Phys& ph, const double* Q, const double* A_U,
                       const double* SG, const double* B, double* rhs, const LsrkFuse& lf, hipStream_t s);
constexpr int SIGMA2_MAX_PARTIALS = 8192;   // >= the persistent grid of kt2_sigma (CUs x resident workgroups per CU)
int launch_sigma_tensor2(int N1, const TensorTables& TT, const MeshDev& M, const Phys& ph, const double* Q, const double* A_U,
                         double* B, double* SG, hipStream_t s, double* vt_partial = nullptr);
int launch_project_tensor2(int N1, const TensorTables& TT, const MeshDev& M, const Phys& ph, const double* Q, double* A_U, hipStream_t s);
// knobs of the A/B builds (-DESDG_AB_HOOKS): set by esdg_api.hip from the environment, never by the shipped library
void ab_tuning_t2(int wg_per_cu, int reserve);   // (-1 = leave)
void ab_tuning_hex(int line);
// v3 last-phase kernel (esdg_kernels_tensor3.hip); -1 where it does not apply
struct StageFuse;
int launch_rhs_tensor3(int N1, const TensorTables& TT, const MeshDev& M, const Phys& ph, const double* Q, const double* A_U,
                       const double* SG, const double* B, double* rhs, const LsrkFuse& lf, hipStream_t s, const StageFuse* sf = nullptr);
int rhs_tensor3_blocks(int N1, int64_t e_count);
inline bool tensor2d_supported_degree(int N1) { return N1 >= 2 && N1 <= ESDG_MAX_N1; }   // workgroups of that launch (StageFuse::partial has one entry each)

struct MeshDev {
  int64_t K;               // local elements
  int64_t trace_nodes;       // K*Nfq + ghost slots (records of the A_U / B trace buffers)
  int64_t e_begin, e_count;  // element range a launch covers (tensor / hex kernels; the host sets 0, K for full launches)
  const double* geo;       // [K][GEO_STRIDE]
  const int32_t* mapP;     // [K][Nfq]  local face-node index, or ghost slot >= K*Nfq
  const uint8_t* bc;       // [K][Nfq]  0 interior/periodic, 1 wall, 2 lid, 3 Dirichlet inflow, 4 copy; may be null
  const double* vlid;      // [K][Nfq]  lid velocity where bc == 2; null = 1 everywhere
  const double* fnrm;      // quads: [K][Nfq][3] = nxJ, nyJ, sJ of every face node AS THE DRIVER HOLDS THEM.  On an affine face they
                           // differ from node to node by the round-off of the driver's set-up (~1e-13 relative), and the reference
                           // uses them per node: replacing them by the face mean moves the RHS by 1e-10 ... 4e-10 relative on
                           // 128^2 ... 256^2 meshes, several times the Float64 reference's own rounding error (round 3,
                           // tools/parity_scaling.py).  The geometry record keeps the means for code that needs a face constant.
  // The same per-node normals for the v2 tensor kernels, as single-precision DIFFERENCES to the face means of the geometry
  // record: fnd[K][Nfq][2] = (nxJ, nyJ) - mean, fsd[K][Nfq] = sJ - mean (read by the wall instantiations only).  On an affine
  // face a node's value differs from the mean by some thousand ulps at most, so mean + (double)(float)(v - mean) == v bit for
  // bit for every component that carries the face (checked at esdg_create; a component below sJ / 4 may lose 2^-24 of its
  // difference, <= 6e-18 sJ) -- a third of the bytes of fnrm.  Where only a quantity that is itself a small jump is
  // scaled by the normal (LF wavespeed and penalty: sJ, lambda) the kernels use the mean.
  const float* fnd;
  const float* fsd;
  // Meshes with walls, CNS, v2 tensor kernels; null otherwise.  dg_grad! / dg_div! (cavity_optimized.jl:549-611) scale NODAL
  // coefficients by rows 1:Np of the metric arrays and divide them by J[i,e], node by node; on an affine element those arrays
  // are a constant plus the round-off of the driver's set-up.  In the elements that touch a wall the lifted wall jump dominates
  // the momentum rows and the reference evaluates it almost exactly, so there that round-off shows (round 3: rhs_viscous! alone
  // 2.6 ... 8.9 x e_orc with one record per element).  kt2_sigma therefore scales gradient and volume divergence of THOSE
  // elements in the nodal basis with these arrays: wgeo[K][5][Np] = rxJ, sxJ, ryJ, syJ (rows 1:Np as passed), J.
  const double* wgeo;
  // ... those elements cost a multiple of the others and sit at regular distances in the element numbering (every Kx-th
  // element on the side walls), which resonates with the persistent kernel's stride of one grid per round: the same few
  // workgroups would take all of them.  With wgeo set, workgroup w takes group k G + ((w + k wall_rot) mod G) in round k
  // (G = grid size): neighbouring workgroups still work on neighbouring groups, and the costly ones move on by wall_rot
  // workgroups per round.  0: the plain stride.
  int32_t wall_rot;
  const double* wJq;       // [K][Nq] (diagnostics) may be null
  // curved (non-affine) hexahedra only, null otherwise: per-node metric terms [K][9][Nh] (row m9 = comp*3 + operator:
  // rxJ sxJ txJ ryJ syJ tyJ rzJ szJ tzJ), J at the quadrature nodes [K][Nq], normals [K][4][Nfq] = nxJ nyJ nzJ sJ
  const double* G9;
  const double* Jq;
  const double* nrm;
  // affine hexahedra whose driver passed per-node arrays (geometry mode 2 of kh_rhs), null otherwise: each node's difference to
  // the element record, three signed 10-bit numbers (x, y, z) per word, in units of the record's scales [34] / [35]:
  // hdv[K][3 operators][Nq] metric rows at the volume nodes, hdf[K][Nfq] the row of its own direction at every face node
  // (hybrid node Nq + f), hdn[K][Nfq] normals (nxJ, nyJ, nzJ) minus the face means
  const uint32_t* hdv;
  const uint32_t* hdf;
  const uint32_t* hdn;
  // Role of a ranged launch in the overlapped sharded schedule (set by rhs_sharded_impl only; 0 everywhere else, incl.
  // esdg_rhs_phase_range): 1 = the interior beside which boundary strips run, 2 = a boundary strip.  The persistent
  // kt2_sigma leaves a few workgroup slots free in role 1 and asks for no more than those in role 2 (ESDG_T2_RESERVE).
  int32_t launch_role;
  // Trace buffers laid out BY MESH FACE (v2 / v3 tensor kernels, round 4): record of local face node fn = f N1 + k of element e
  // at slot f (K N1) + e N1 + k instead of e Nfq + fn -- four planes, one per face of the reference element, so that the faces a
  // workgroup's consecutive elements read from their neighbours across face f are ONE contiguous run of the opposite plane
  // (1 KB for six elements at N = 4) instead of one 160-byte run per element inside that element's 640-byte block.  Measured
  // with known byte counts (profiles/r04_fetch_calibration.txt): the per-element runs fetch 1.6 x their bytes (two 128-byte lines
  // per 160-byte run) and stream at 4.7 instead of 6.4 TB/s.  With bf set, mapP holds SLOTS (remapped on the host, ghost
  // slots >= K Nfq unchanged).  0: the linear layout (round-1 and generic kernels, hexahedra).
  int32_t bf;
};

// slot of the trace record (A_U, B) of local face node fn of element e
template <int N1>
__device__ __forceinline__ int64_t trace_slot(const MeshDev& M, int64_t e, unsigned fn) {
  return M.bf ? (int64_t)(fn / N1) * (M.K * N1) + e * N1 + (int64_t)(fn % N1) : e * (4 * N1) + (int64_t)fn;
}

// Attribution builds (-DESDG_EXP_WINDOW=mask, mask = 2^k - 1; tools/session_r03b.sh): every global address a kernel of the
// tensor path forms -- state, geometry, neighbour index, traces, outputs -- is folded into the first mask+1 elements, which
// stay in L2.  Results are wrong, the instruction stream is the same: the time left is what the kernel costs with the memory
// system taken out (VALU + LDS + barriers), and the difference to the normal build is what HBM adds.
#ifdef ESDG_EXP_WINDOW
#define ESDG_EW(e) ((e) & (int64_t)(ESDG_EXP_WINDOW))
#define ESDG_EWN(n, nfq) ((n) % (unsigned)(((ESDG_EXP_WINDOW) + 1) * (nfq)))
#else
#define ESDG_EW(e) (e)
#define ESDG_EWN(n, nfq) (n)
#endif

// optional fusion of the low-storage RK update into the last phase (esdg_rhs_lsrk):
//   res = a*res + dt*rhs ; Q += b*res   instead of storing rhs   (dg2D_euler_quad.jl:204-205)
struct LsrkFuse {
  double* Qw;   // the state, updated in place (null = plain rhs store)
  double* res;
  double a, b, dt;
};

// optional fusion of the DOPRI45 stage combination and error norm into the last phase (esdg_dopri45_attempt; the STG
// instantiations of kt3_rhs and kh_rhs_l).  The launch stores its right-hand side k_s as always and, from the value in registers,
//   y != null : y = x0 + dt (sum_{j<ns} c[j] k[j] + c_last k_s), the state of the next stage (y may be the state the launch read:
//               a workgroup reads its elements' state at entry only), and with e_out != null also
//               e_out = sum_{j<ns} ce[j] k[j] + ce_last k_s, the error combination so far;
//   err != 0  : e = (what the rhs array held at the node: the error combination so far) + ce_last k_s, and
//               partial[node] = sum over the node's fields (field order, one fma chain) of (|e| / (tol (1 + |x0|)))^2 at the
//               node's own index (dg2D_CNS_cavity_optimized.jl:1014-1021); the host adds them with k_chunk_sum + k_sum.
// The chains are the fma chains of k_axpy_stages / k_dopri_err (esdg_kernels.hip) in the same order, and the norm's terms meet in
// k_dopri_err's order: same bits as the unfused attempt per node AND in the estimate, however the last phase is cut into launches.
struct StageFuse {
  double* y;
  const double* x0;
  const double* k[6];
  double c[6], c_last, dt;
  int ns, err;
  double* e_out;
  double ce[6], ce_last, tol;
  double* partial;
};

struct Phys {
  int formulation;
  double lf_scale;
  int inviscid_dissp, viscous_dissp, BCTYPE;
  double Re, mu, lambda, Pr;
  double kappa, inv_Re;   // 1.4*mu/Pr and 1/Re, computed once on the host (a per-lane fp64 division costs ~15 instructions)
  int dbg;  // timing-ablation mask from the ESDG_DBG environment variable (1: skip the volume flux differencing, 2: skip the
            // viscous stage, 16: hex workgroup remap off, 32: kt3_rhs computes every logarithm -- its smooth-wave short cut off,
            // same results bit for bit); 0 in normal use
  double inflow_q[6];   // BCTYPE 4: Dirichlet state as a trace record (rho,u,v,beta,log rho,log beta)
  double inflow_vv[3];  //           and its entropy variables (v2,v3,v4) = v_ufun(...)[2:4]
  int parts;  // bit 0: inviscid terms (rhs_inviscid!), bit 1: viscous terms (rhs_viscous!); 3 = rhsRK!
};

// kernel launchers (esdg_kernels.hip); return hipError_t as int
int launch_project(const Tables& T, const MeshDev& M, const Phys& ph, const double* Q, double* A_U, double* A_v,
                   hipStream_t s);
int launch_sigma(const Tables& T, const MeshDev& M, const Phys& ph, const double* Q, const double* A_v, double* B,
                 hipStream_t s);
int launch_rhs(const Tables& T, const MeshDev& M, const Phys& ph, const double* Q, const double* A_U,
               const double* A_v, const double* B, double* rhs, hipStream_t s);
int launch_pack(const double* src, int ncomp, const int32_t* list, int64_t n, double* dst, hipStream_t s);
int launch_rhstest(const Tables& T, const MeshDev& M, const Phys& ph, const double* Q, const double* rhs,
                   double* partial, int nblocks, hipStream_t s);
int launch_lsrk(double* Q, double* resQ, const double* rhs, double a, double b, double dt, int64_t n, hipStream_t s);
int launch_axpy_stages(double* y, const double* x0, const double* const* k, const double* coef, int ns, double dt,
                       int64_t n, hipStream_t s);
int launch_sum(const double* x, int64_t n, double* out, hipStream_t s);
// Hairer norm (k_dopri_err / k_chunk_sum, esdg_kernels.hip): one term per node (its nfld fields in one fma chain), runs of
// ESDG_ERR_CHUNK consecutive nodes, one sum each into chunk[0 .. err_chunks(nodes)); launch_sum over those gives the numerator in
// an order that depends on (nodes, nfld) alone
#define ESDG_ERR_CHUNK 4096
static inline int64_t err_chunks(int64_t n) { return (n + ESDG_ERR_CHUNK - 1) / ESDG_ERR_CHUNK; }
int launch_dopri_err(const double* Q, const double* const* k, const double* coefE, int ns, double tol, int64_t nodes, int nfld,
                     double* chunk, hipStream_t s);
int launch_chunk_sum(const double* x, int64_t n, double* chunk, hipStream_t s);
// error functionals (esdg_kernels_err.hip); all arrays on the device
struct ErrDev {
  int64_t K;
  int Np, Nq2, Nfq;
  const double* Vq2;          // [Nq2][Np] row-major: state nodes -> error quadrature
  const double* wq2;          // [Nq2]
  const double *x, *y, *J;    // [K][Np] at the state's nodes
  const double* Vf;           // [Nfq][Np] row-major (boundary-velocity functional), may be null
  const double* wf;           // [Nfq]
};
int launch_err_l2(const ErrDev& E, const double* Q, int kind, const double* par, double t, double* partial, int nblocks,
                  hipStream_t s);
int launch_err_nodal(const ErrDev& E, const double* Q, int kind, const double* par, double t, double* partial, int nblocks,
                     hipStream_t s);
int launch_err_boundary(const ErrDev& E, const uint8_t* bc, const double* vlid, const double* Q, double Jf, double* partial,
                        int nblocks, hipStream_t s);
int launch_min_rho_p(const double* Q, int nfld, int64_t n, double* partial, int nblocks, hipStream_t s);
bool supported_degree(int N1);

// hexahedral path (esdg_kernels_hex.hip)
struct HexTables;
bool hex_supported_degree(int N1);
int launch_project_hex(int N1, const HexTables& HT, const MeshDev& M, const Phys& ph, const double* Q, double* A_U,
                       hipStream_t s);
int launch_rhs_hex(int N1, const HexTables& HT, const MeshDev& M, const Phys& ph, const double* Q, const double* A_U,
                   double* rhs, const LsrkFuse& lf, hipStream_t s, const StageFuse* sf = nullptr);
int rhs_hex_blocks(int N1, int64_t e_count);   // workgroups of that launch (StageFuse::partial); -1 where the fused stage does not apply
int launch_rhstest_hex(int64_t n, const double* wJq, const double* Q, const double* rhs, double* partial, int nblocks,
                       hipStream_t s);
int launch_log_test(const double* x, double* y, int64_t n, hipStream_t s);

}  // namespace esdg
