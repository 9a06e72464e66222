/* esdg_hip.h -- C ABI of libesdg_hip.so: MI355X (gfx950) right-hand-side engine for the
 * entropy-stable DG compressible Euler / Navier-Stokes solvers of yiminllin/ESDG-CNS.
 *
 * This is the drop-in boundary of the hot path.  The reference has no
 * FFI: its `rhs` functions are script-local Julia (examples/dg2D_euler_quad.jl:141,
 * examples/CompressibleNS/dg2D_CNS_cavity_optimized.jl:447,749,955).  Each entry point below
 * names the reference function it replaces; INTEGRATION.md shows the Julia `ccall` stubs.
 *
 * Conventions (chosen so a Julia driver passes `pointer(A)` with no copies):
 *   - all matrices are dense **column-major** float64, exactly as Julia stores them;
 *   - index arrays are int64 and **1-based** (mapP, mapB), linear into (Nfq x K);
 *   - a state / rhs is `nfld` (=4 in 2D) matrices of shape (Np x K); on the device they are
 *     stacked field-major in ONE buffer: Q[f*K*Np + e*Np + i]  (== Julia's per-field layout);
 *   - every function returns 0 on success, a negative esdg_status otherwise;
 *     esdg_last_error() gives the message (the reference throws Julia exceptions instead);
 *   - `stream` arguments are hipStream_t passed as void* (NULL = default stream);
 *   - gamma = 1.4 is fixed, as in the reference (EntropyStableEuler.jl:9 and the literals
 *     0.4/1.4/2.4 in dg2D_CNS_cavity_optimized.jl:463-474).
 * No CPU fallback exists: without a HIP device every compute entry point fails with
 * ESDG_ERR_NO_DEVICE.  The library reads no environment variable; the switches that select partner kernels, geometry modes and
 * schedule variants for A/B measurements exist only in the separate build libesdg_hip_ab.so (esdg_cns_amd/build.py).
 */
#ifndef ESDG_HIP_H
#define ESDG_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum {
  ESDG_OK = 0,
  ESDG_ERR_ARG = -1,        /* bad argument / unsupported size */
  ESDG_ERR_STRUCTURE = -2,  /* operators are not tensor-product-sparse / mesh not affine */
  ESDG_ERR_NO_DEVICE = -3,  /* no HIP device or HIP runtime error */
  ESDG_ERR_ALLOC = -4,
  ESDG_ERR_STATE = -5,      /* call order (e.g. workspace not bound) */
  ESDG_ERR_COMM = -6        /* RCCL error, or halo plans of two ranks disagree */
} esdg_status;

typedef struct esdg_ctx esdg_ctx;

/* Formulation of the hot path. */
typedef enum {
  /* `rhs` of examples/dg2D_euler_quad.jl:141-194: state lives at the Gauss quadrature nodes
   * (Np == Nq), operators in the quadrature basis (Ph = W^-1 Vh', Lf), LF factor .5 (:165). */
  ESDG_EULER_COLLOCATED = 0,
  /* `rhsRK!` of examples/CompressibleNS/dg2D_CNS_cavity_optimized.jl:955-972 = rhs_inviscid!
   * (:447-528) + rhs_viscous! (:749-849): state = nodal (LGL) coefficients, LF factor .25 (:508). */
  ESDG_CNS_MODAL = 1,
  /* rhs_inviscid! alone (same file), i.e. modal Euler. */
  ESDG_EULER_MODAL = 2,
  /* `rhs` of examples/dg3D_euler_hex.jl:167-222 (hexahedra, 5 fields, state at the Gauss nodes, Ph includes
   * the factor 2 (:96), LF factor `0*.25` (:193) -> esdg_phys_t.lf_scale).  Created with esdg_create_hex. */
  ESDG_EULER_HEX_COLLOCATED = 3
} esdg_formulation;

/* Reference-element operators: fields of `rd::RefElemData` (src/SetupDG.jl:38-75) and of the
 * driver's `ops` tuple.  Column-major.  Pointers not needed by the formulation may be NULL. */
typedef struct {
  int32_t N;    /* polynomial degree.  Quads: 1..11, periodic meshes and meshes with walls alike; the generic pair-list kernels (operators
                 * that do not factor into 1D tables; same boundary closures) 1..7.  esdg_create refuses anything beyond with the reason. */
  int32_t Np;   /* rows of a state matrix: (N+1)^2 */
  int32_t Nq;   /* volume quadrature nodes: length(rd.wq) */
  int32_t Nfq;  /* face quadrature nodes: length(rd.wf) */
  /* both formulations */
  const double* Qrhskew; /* (Nh x Nh), Nh = Nq+Nfq; dg2D_euler_quad.jl:61 / cavity_optimized.jl:82 */
  const double* Qshskew; /* (Nh x Nh) */
  const double* Ph;      /* collocated: (Nq x Nh) W^-1 Vh' (euler_quad.jl:77); modal: (Np x Nh) M\Vh' (:76) */
  const double* wq;      /* (Nq) rd.wq */
  const double* wf;      /* (Nfq) rd.wf */
  /* collocated only */
  const double* Ef;      /* (Nfq x Nq) Vf*Pq, euler_quad.jl:49 */
  const double* Lf;      /* (Nq x Nfq) W^-1 Ef' Wf, euler_quad.jl:78 */
  /* modal only */
  const double* Vq;      /* (Nq x Np) rd.Vq */
  const double* Pq;      /* (Np x Nq) rd.Pq */
  const double* VhP;     /* (Nh x Nq) Vh*Pq, cavity_optimized.jl:77 */
  const double* LIFT;    /* (Np x Nfq) rd.LIFT */
  const double* Vf;      /* (Nfq x Np) rd.Vf     (viscous) */
  const double* Dr;      /* (Np x Np) rd.Dr      (viscous) */
  const double* Ds;      /* (Np x Np) rd.Ds      (viscous) */
} esdg_ops_t;

/* Mesh data: fields of `md::MeshData` (src/SetupDG.jl:77-115) for the LOCAL elements of this
 * process.  rxJ..syJ are the Vh-interpolated (Nh x K) arrays the drivers store back into md
 * (euler_quad.jl:86-88); only affine elements are supported (the reference's flux differencing
 * assumes them too: "assumes affine elements for now", euler_quad.jl:175) and this is checked. */
typedef struct {
  int64_t K;             /* local element count */
  int32_t geo_ld;        /* leading dimension (rows) of rxJ..syJ: Nh (drivers) or Np */
  const double *rxJ, *sxJ, *ryJ, *syJ;
  const double* J;       /* (Np x K) */
  const double* wJq;     /* (Nq x K) -- diagnostics (rhstest) only, may be NULL */
  const double *nxJ, *nyJ, *sJ; /* (Nfq x K) */
  const int64_t* mapP;   /* (Nfq x K) 1-based GLOBAL linear index into (Nfq x Kglobal) */
  const int64_t* mapB;   /* wall-boundary face nodes (md.mapB), 1-based GLOBAL linear index, may be NULL (periodic);
                          * walls: mirror state for the inviscid flux, BCTYPE-dependent entropy-variable and
                          * stress traces, boundary penalty (init_BC_funs, cavity_optimized.jl:135-265, 827-837) */
  int64_t NmapB;
  const uint8_t* bkind;  /* per mapB entry: 0 = wall, 1 = lid (init_BC_funs :139-148); NULL = all wall.
                          * BCTYPE 4: 0 = copy ("rightwall"), 1 = Dirichlet inflow ("leftwall"); the listed nodes may
                          * carry a periodic partner in mapP (the driver patches mapP first, modalESDG.jl:72-78) */
  /* element-index sharding.  Single process: elem_offset=0, Kglobal=K, nranks=1. */
  int64_t elem_offset;   /* global index (0-based) of the first local element */
  int64_t Kglobal;
  int32_t nranks, rank;
  const int64_t* rank_offsets; /* (nranks+1) element offsets of every rank; NULL if nranks==1 */
  const double* vlid;    /* per mapB entry: lid velocity, read where bkind == 1 (BCTYPE 1).  NULL = 1 everywhere
                          * (cavity_optimized.jl:147); dg2D_CNS_convergence_test.jl:76 passes (1+cos(pi*xlid))/2 */
} esdg_mesh_t;

typedef struct {
  int32_t formulation;     /* esdg_formulation */
  double lf_scale;         /* .5 (euler_quad.jl:165) or .25 (cavity_optimized.jl:508); 0 disables LF */
  int32_t inviscid_dissp;  /* cavity_optimized.jl:29 */
  int32_t viscous_dissp;   /* cavity_optimized.jl:30 */
  int32_t BCTYPE;          /* 1 adiabatic no-slip, 2 isothermal, 3 slip (cavity_optimized.jl:26);
                            * 4 = inflow/outflow closures of the shock-tube driver (dg2D_CNS_modalESDG.jl:161-217):
                            *     bkind 1 = Dirichlet state `inflow_*`, bkind 0 = copy of the interior trace, lam = 0 and
                            *     sigma+ = sigma- on both; that driver has no penalty, so viscous_dissp must be 0 */
  double Re, mu, lambda, Pr; /* cavity_optimized.jl:33-36; lambda as passed to init_visc_fxn (:646) */
  double inflow_rho, inflow_u, inflow_v, inflow_p; /* BCTYPE 4: (rhoL, uL, vL, pL), dg2D_CNS_modalESDG.jl:50-57 */
} esdg_phys_t;

/* Hexahedral path: operators of examples/dg3D_euler_hex.jl:34-98 (quadrature basis) and the 3D MeshData
 * fields the driver holds when it calls `rhs` (:167).  Same conventions as above. */
typedef struct {
  int32_t N;    /* polynomial degree: hexahedra 1..10 (one element per workgroup from N = 7 on; affine and curved meshes) */
  int32_t Nq;   /* (N+1)^3 */
  int32_t Nfq;  /* 6 (N+1)^2 */
  const double *Qrhskew, *Qshskew, *Qthskew; /* (Nh x Nh), dg3D_euler_hex.jl:49-51 */
  const double* Ph;  /* (Nq x Nh) 2 W^-1 Vh', :96 */
  const double* Lf;  /* (Nq x Nfq) W^-1 Ef' Wf, :97 */
  const double* Ef;  /* (Nfq x Nq) Vf*Pq, :39 */
  const double *wq, *wf; /* may be NULL (not needed by the kernels) */
} esdg_hex_ops_t;

typedef struct {
  int64_t K;
  int32_t geo_ld;  /* rows of the metric arrays.  Nh as stored by the driver (:88-90): every node's own metric terms and
                    * normals are used, averaged per pair as sparse_hadamard_sum does (:145-151) -- on affine meshes from an
                    * element record plus 10-bit per-node differences (the arrays then are constants plus the set-up's
                    * round-off, which the reference's per-node use passes on to the RHS), on curved meshes (the `a != 0`
                    * mapping, :67-73, detected at create) from the full arrays, which need all Nh rows.  Any other value
                    * >= 1 on an affine mesh says the geometry is element-constant: the mean of the passed rows is used.
                    * Environment: ESDG_HEX_GEOMETRY=element / ESDG_HEX_PER_NODE=1 force the plain record / the full arrays */
  const double *rxJ, *sxJ, *txJ, *ryJ, *syJ, *tyJ, *rzJ, *szJ, *tzJ; /* (geo_ld x K) */
  const double* J;    /* (Nq x K) = Vq*J, :94 */
  const double* wJq;  /* (Nq x K), diagnostics only, may be NULL */
  const double *nxJ, *nyJ, *nzJ, *sJ; /* (Nfq x K), :81-86 */
  const int64_t* mapP; /* (Nfq x K) 1-based GLOBAL linear index (periodic patch applied, :59-65) */
  int64_t elem_offset, Kglobal;
  int32_t nranks, rank;
  const int64_t* rank_offsets;
} esdg_hex_mesh_t;

/* ---- life cycle ------------------------------------------------------------------------ */
/* Copies operators/mesh to the device, derives the sparse collocated operators, converts
 * mapP to 0-based int32 with ghost slots for off-rank neighbours. */
int esdg_create(const esdg_ops_t* ops, const esdg_mesh_t* mesh, const esdg_phys_t* phys, esdg_ctx** out);
/* Hexahedral engine (replaces `rhs`, dg3D_euler_hex.jl:167-222).  The returned context is driven through the
 * same entry points as the 2D ones (esdg_workspace_bytes .. esdg_rhs_lsrk, esdg_rhstest, esdg_halo_*); states are
 * [5][K][Nq], two phases, one face-trace exchange (rho,u,v,w,beta). */
int esdg_create_hex(const esdg_hex_ops_t* ops, const esdg_hex_mesh_t* mesh, const esdg_phys_t* phys, esdg_ctx** out);
int esdg_num_fields(const esdg_ctx* ctx);   /* 4 (2D) or 5 (hex) */
int esdg_destroy(esdg_ctx* ctx);
const char* esdg_last_error(void);
const char* esdg_version(void);
/* sizeof of a public struct by name ("esdg_ops_t", "esdg_mesh_t", "esdg_phys_t", "esdg_hex_ops_t", "esdg_hex_mesh_t",
 * "esdg_err_ops_t"), -1 if unknown: lets an FFI binding (Julia struct, ctypes.Structure) check its mirror of the
 * layout at load time instead of corrupting memory */
int64_t esdg_abi_sizeof(const char* struct_name);

/* Scratch (face-trace buffers A/B, halo send buffers) is caller-owned device memory so the host
 * framework (torch / Julia) controls allocation; bind it once. */
size_t esdg_workspace_bytes(const esdg_ctx* ctx);
int esdg_bind_workspace(esdg_ctx* ctx, void* dev_ptr, size_t bytes);

/* ---- the hot path ------------------------------------------------------------------------
 * One RHS evaluation = phases 0..esdg_num_phases()-1; between phase p and p+1 the face traces
 * of off-rank neighbours must be exchanged (esdg_halo_segment).  With nranks==1 esdg_rhs()
 * runs all phases back to back.  Q_dev / rhs_dev: device buffers [nfld][K][Np] (nfld = esdg_num_fields).
 * Replaces: rhs (euler_quad.jl:141), rhs_inviscid! / rhs_viscous! / rhsRK!
 * (cavity_optimized.jl:447, 749, 955).  All launches are asynchronous on `stream`. */
int esdg_num_phases(const esdg_ctx* ctx);
/* 1 if the tensor-line kernels (esdg_kernels_tensor2.hip, esdg_kernels_tensor3.hip / esdg_kernels_hex.hip) are in use, 0 for the generic pair-list kernels
 * (operators without tensor-product Gauss structure, or ESDG_FORCE_GENERIC=1 in the environment). */
int esdg_uses_tensor_kernels(const esdg_ctx* ctx);
int esdg_rhs_phase(esdg_ctx* ctx, int phase, const double* Q_dev, double* rhs_dev, void* stream);
int esdg_rhs(esdg_ctx* ctx, const double* Q_dev, double* rhs_dev, void* stream);

/* RHS evaluation fused with the low-storage RK stage that consumes it (dg2D_euler_quad.jl:200-206):
 *   rhs = RHS(Q);  resQ = a*resQ + dt*rhs;  Q += b*resQ
 * The last phase updates Q and resQ in place and never writes rhs to memory (saves five full-state sweeps per
 * stage).  Tensor kernels only (ESDG_ERR_STATE otherwise); esdg_rhs_phase_lsrk is the per-phase form for
 * sharded meshes (non-final phases behave exactly like esdg_rhs_phase). */
int esdg_rhs_lsrk(esdg_ctx* ctx, double* Q_dev, double* resQ_dev, double a, double b, double dt, void* stream);
int esdg_rhs_phase_lsrk(esdg_ctx* ctx, int phase, double* Q_dev, double* resQ_dev, double a, double b, double dt,
                        void* stream);

/* Entropy-production diagnostics returned by the reference beside rhsQ:
 * diag[0] = rhstest = sum(wJq .* v(u) .* rhs)   (euler_quad.jl:186-191, cavity_optimized.jl:958-966)
 * Device-side reduction, synchronises `stream`.  Local elements only (all-reduce across ranks
 * is the caller's job). */
int esdg_rhstest(esdg_ctx* ctx, const double* Q_dev, const double* rhs_dev, double* diag, void* stream);

/* Admissibility of a state (the reference throws Julia DomainErrors from log / sqrt / ^ on negative density or
 * pressure; the kernels would silently produce NaNs): min_rho_p[0] = min rho, [1] = min p over the local nodal values
 * of Q_dev (NaNs count as -1e300).  Device reduction, synchronises `stream`. */
int esdg_check_state(esdg_ctx* ctx, const double* Q_dev, double* min_rho_p, void* stream);

/* rhsRK! = rhs_inviscid! + rhs_viscous! (cavity_optimized.jl:955-957).  esdg_set_parts selects which of the two a
 * CNS context evaluates in the following esdg_rhs* calls: 1 = rhs_inviscid! (:447), 2 = rhs_viscous! (:749), 3 = both
 * (default).  esdg_viscous_entropy_test returns the second value of rhs_viscous!,
 *   visc_test = sum(wJq .* (VUx .* sigma_x + VUy .* sigma_y))   (:802-806),
 * so that rhsRK!'s third return is  rhstest_visc = esdg_rhstest(Q, rhs_viscous) + visc_test  (:962-969).
 * (Every degree the context serves: the reduction rides in the phase-1 kernel.  Runs phases 0 and 1 itself; synchronises
 * `stream`.  On a sharded context -- communicator attached -- it exchanges
 * the traces of phase 0 and returns this rank's share of the sum, like esdg_rhstest: add the shares with
 * esdg_comm_allreduce.) */
int esdg_set_parts(esdg_ctx* ctx, int parts);
int esdg_viscous_entropy_test(esdg_ctx* ctx, const double* Q_dev, double* visc_test, void* stream);

/* Literal drop-in with host arrays (Julia Matrix{Float64} per field): H2D, rhs, D2H.
 * PCIe-bound -- for validation, not for time stepping. nranks must be 1. */
int esdg_rhs_host(esdg_ctx* ctx, const double* const* Q, double* const* rhs);

/* ---- error functionals of the drivers, on the device (2D contexts) ------------------------ */
/* What the scripts print after a run, without moving the state to the host.  Exact solutions: */
enum {
  ESDG_EXACT_VORTEX = 0, /* vortex(x,y,t), EntropyStableEuler.jl:21-35 (x0=5, y0=0, beta=5); par ignored */
  ESDG_EXACT_BECKER = 1  /* exact_sol_viscous_shocktube, dg2D_CNS_modalESDG.jl:545-579 (bisection, max_iter 100,
                          * tol 1e-14); par = (v_0, v_1, v_01, m_0, L_k = kappa/m_0/cv, v_inf) */
};
typedef struct {
  int32_t Nq2;         /* nodes of the error quadrature per element ((N+3)^2 for the N+2 Gauss rule); 0 = none */
  const double* Vq2;   /* (Nq2 x Np) column-major, state nodes -> error quadrature: vandermonde_2D(N,rq2,sq2)/VDM
                        * (dg2D_euler_quad.jl:220).  Np = nodes of the state per element; the collocated Euler driver,
                        * whose state lives at the Gauss nodes, folds its projection in: Vq2*Pq (:215) */
  const double* wq2;   /* (Nq2) */
  const double *x, *y; /* (Np x K) coordinates at the state's nodes (md.x, md.y; md.xq, md.yq when collocated) */
  const double* J;     /* (Np x K) */
  const double* Vf;    /* (Nfq x Np) column-major rd.Vf and (Nfq) rd.wf: boundary-velocity functional only, */
  const double* wf;    /*   may be NULL */
} esdg_err_ops_t;
/* copies the arrays to the device (once per mesh); the esdg_error_* calls below need it */
int esdg_error_setup(esdg_ctx* ctx, const esdg_err_ops_t* e);
/* out[1+f] = sum_{elements, nodes} wq2*(Vq2*J) * (Vq2*Q_f - Qexact_f(Vq2*x, Vq2*y, t))^2 over the LOCAL elements,
 * out[0] = sqrt(sum_f out[1+f])  = "L2err" of dg2D_euler_quad.jl:224-231 (sharded runs add out[1..4] over the ranks).
 * Synchronises the stream. */
int esdg_error_l2(esdg_ctx* ctx, const double* Q, int32_t exact, const double* par, double t, double* out5, void* stream);
/* Nodal errors of dg2D_CNS_modalESDG.jl:745-771 over rho, rho*u, E at the state's nodes:
 * out[0] = L1err  = sum_f sum|Qex_f-Q_f| / sum|Q_f|   (the script's uniform J cancels),
 * out[1] = Linferr = sum_f max|Qex_f-Q_f| / max|Q_f|,
 * out[2+4c..5+4c] = (sum|d|, sum|q|, max|d|, max|q|) of field c in (rho, rho*u, E) for reductions over ranks. */
int esdg_error_nodal(esdg_ctx* ctx, const double* Q, int32_t exact, const double* par, double t, double* out14, void* stream);
/* Boundary-velocity error of the lid-driven cavity, dg2D_CNS_convergence_test.jl:1055-1080, with
 * u = Vf*(Q[2]./Q[1], Q[3]./Q[1]) and Jf = 2/K1D in the script:
 *   out[2] = sum_{wall+lid nodes} Jf*wf*u_2^2, out[3] = sum_{wall} Jf*wf*u_1^2, out[4] = sum_{lid} Jf*wf*(u_1-vlid)^2
 *   (vlid from esdg_mesh_t.vlid), out[1] = sqrt(out[2]+out[3]+out[4]) = the error as the script reads,
 *   out[0] = sqrt(out[2]) = the error as Julia executes it: the statement `err = sum(...)` (:1075) is complete at its
 *   line end, so the two continuation lines starting with `+sum(...)` (:1076-1077) are separate unary-plus expressions
 *   and never reach `err` (quirk, reproduced for parity of the printed numbers). */
int esdg_error_boundary_velocity(esdg_ctx* ctx, const double* Q, double Jf, double* out5, void* stream);  /* esdg_num_fields() pointers each */

/* ---- halo exchange plan (element-index sharding) ---------------------------------------- */
/* The reference's three x[mapP] gathers (QM/Uf+lam :496-511, VUf :776, sigma_f :813-814) become face-trace
 * exchanges: A_U (entropy-projected face state + lam) and B (normal viscous stress); the tensor kernels rebuild the
 * neighbour's projected entropy variables VUf[mapP] from its A_U record, the generic kernels exchange them as a
 * third buffer A_v.  The tensor kernels' A_U record is (rho,u,v,beta); its logs, energy and wavespeed are rebuilt by
 * the consumer.
 * esdg_num_exchanges / esdg_exchange_info enumerate them: exchange x is produced (and packed) by
 * phase `after_phase` and must have landed before phase `before_phase` starts. */
int esdg_halo_num_neighbors(const esdg_ctx* ctx);
int esdg_num_exchanges(const esdg_ctx* ctx);
int esdg_exchange_info(const esdg_ctx* ctx, int xch, int32_t* after_phase, int32_t* before_phase, int32_t* ncomp);
/* Exchange `xch`, neighbour slot `nbr`: peer rank and the byte ranges inside the bound workspace
 * to send from (packed, contiguous) / receive into (ghost slots of the trace buffer, contiguous).
 * send_bytes of rank a towards b equals recv_bytes of b from a on a conforming mesh. */
int esdg_halo_segment(const esdg_ctx* ctx, int xch, int nbr, int32_t* peer, size_t* send_off,
                      size_t* send_bytes, size_t* recv_off, size_t* recv_bytes);

/* Overlap of the exchanges with computation.  [e_begin, e_end) is the longest run of local elements that touch no
 * ghost slot ("interior"; the rest is "boundary").  esdg_rhs_phase_range runs one phase on an element range WITHOUT
 * packing; esdg_halo_pack packs exchange `xch` once its producing phase has run on all boundary elements.  Schedule
 * (RhsEngine.rhs_into): phase 0 on the boundary ranges, pack + start exchange, phase 0 on the interior; every later
 * phase p: interior first (overlaps the incoming exchange), wait, boundary ranges, pack + start the exchange produced
 * by p.  Tensor / hex kernels only. */
int esdg_interior_range(const esdg_ctx* ctx, int64_t* e_begin, int64_t* e_end);
int esdg_rhs_phase_range(esdg_ctx* ctx, int phase, int64_t e_begin, int64_t e_count, const double* Q_dev, double* rhs_dev,
                         void* stream);
int esdg_rhs_phase_range_lsrk(esdg_ctx* ctx, int phase, int64_t e_begin, int64_t e_count, double* Q_dev, double* resQ_dev,
                              double a, double b, double dt, void* stream);
int esdg_halo_pack(esdg_ctx* ctx, int xch, void* stream);

/* ---- RCCL transport inside the library --------------------------------------------------------------------------
 * No reference counterpart: the reference is single-process; this is the "RCCL halo exchange over xGMI for
 * mapP face neighbours once per RK stage" of the north star, reachable from a Julia or C host without torch.
 * One process per GPU.  Bootstrap: ONE rank calls esdg_comm_unique_id and hands the ESDG_COMM_ID_BYTES bytes to the
 * others by whatever the host has (MPI_Bcast, a file, torch.distributed); every rank then calls esdg_comm_init with its
 * shard's rank / nranks (must equal esdg_mesh_t.rank / .nranks; collective: ncclCommInitRank).  esdg_comm_init also
 * cross-checks the halo plan with every neighbour (send count of one side == receive count of the other).
 * With a communicator attached, esdg_rhs / esdg_rhs_lsrk / esdg_lsrk45_step / esdg_dopri45_attempt work on a sharded
 * context: they run the overlapped schedule described at esdg_interior_range, the exchanges as grouped
 * ncclSend/ncclRecv (one group per producing phase) on an internal comm stream, ordered against `stream` with events.
 * esdg_halo_exchange / esdg_halo_wait are that schedule's two transport steps for hosts that drive the phases
 * themselves: post everything phase `phase` produced (its packed buffers must be complete on `stream`); make
 * `stream` wait for everything that must have landed before `phase`.
 * esdg_comm_allreduce: sum (op 0) / max (op 1) / min (op 2) of n <= 64 host doubles over the ranks, in place
 * (rhstest, DOPRI error norm, dt; synchronises `stream`).
 * esdg_comm_set_loopback(ctx, 1) before esdg_comm_init(ctx, id, 0, 1): one-GPU rehearsal of the transport -- every
 * neighbour is this rank itself, a segment sent towards neighbour n is received as the ghost data of neighbour n+1
 * (cyclic), which for a strip that is periodic by itself is exactly what the real neighbours would send. */
#define ESDG_COMM_ID_BYTES 128
int esdg_comm_unique_id(void* id_out);
int esdg_comm_init(esdg_ctx* ctx, const void* id, int rank, int nranks);
int esdg_comm_set_loopback(esdg_ctx* ctx, int on);
int esdg_comm_size(const esdg_ctx* ctx);   /* ranks in the communicator as RCCL reports them (ncclCommCount); 0 = none */
int esdg_comm_destroy(esdg_ctx* ctx);
int esdg_halo_exchange(esdg_ctx* ctx, int phase, void* stream);
int esdg_halo_wait(esdg_ctx* ctx, int phase, void* stream);
int esdg_comm_allreduce(esdg_ctx* ctx, double* host_vals, int n, int op, void* stream);

/* Host-only construction of the same plan (no GPU needed; used by the gloo CPU tests and by
 * hosts that want to inspect the partition).  mapP: (Nfq x K) 1-based GLOBAL indices of the local
 * elements.  Offsets/counts are in face nodes; ghost slot g lives at local index K*Nfq + g. */
typedef struct esdg_halo_plan esdg_halo_plan;
int esdg_halo_plan_create(const int64_t* mapP, int64_t K, int32_t Nfq, int64_t elem_offset, int64_t Kglobal,
                          int32_t nranks, const int64_t* rank_offsets, esdg_halo_plan** out);
int esdg_halo_plan_destroy(esdg_halo_plan* plan);
int esdg_halo_plan_num_neighbors(const esdg_halo_plan* plan);
int64_t esdg_halo_plan_num_ghosts(const esdg_halo_plan* plan);
int64_t esdg_halo_plan_num_sends(const esdg_halo_plan* plan);
int esdg_halo_plan_neighbor(const esdg_halo_plan* plan, int nbr, int32_t* peer, int64_t* send_off,
                            int64_t* send_cnt, int64_t* recv_off, int64_t* recv_cnt);
const int32_t* esdg_halo_plan_mapP(const esdg_halo_plan* plan);      /* (Nfq x K) local/ghost indices */
const int32_t* esdg_halo_plan_sendlist(const esdg_halo_plan* plan);  /* num_sends local face nodes */

/* ---- the steps either side of the path ----------------------
 * Low-storage RK stage of src/CommonUtils.jl:29-49 as used in euler_quad.jl:204-205:
 *   resQ = a*resQ + dt*rhs ;  Q += b*resQ          (n = 4*K*Np doubles) */
int esdg_lsrk_update(double* Q_dev, double* resQ_dev, const double* rhs_dev, double a, double b, double dt,
                     int64_t n, void* stream);
/* y = x0 + dt * sum_s coef[s]*k[s]  (DOPRI stage combination, cavity_optimized.jl:1004-1010) */
int esdg_axpy_stages(double* y_dev, const double* x0_dev, const double* const* k_dev, const double* coef,
                     int nstages, double dt, int64_t n, void* stream);
/* Hairer error norm numerator sum((|sum_s E[s] k[s]| / (tol*(1+|Q|)))^2) (cavity_optimized.jl:1014-1021) */
int esdg_dopri_error(const double* Q_dev, const double* const* k_dev, const double* coefE, int nstages,
                     double tol, int64_t n, double* result_host, void* stream);
/* The same sum for a state of nfld fields of `nodes` entries each (n = nfld * nodes), in the order esdg_dopri45_attempt uses:
 * a node's nfld terms first (field order, one fma chain), then the nodes' sums in an order that depends on `nodes` alone (runs of
 * 4096 consecutive nodes, a fixed tree inside a run, the runs in a fixed order).  esdg_dopri_error(n) is the nfld = 1 case.  A host
 * that drives the stages itself gets the bits of esdg_dopri45_attempt's estimate from this entry point. */
int esdg_dopri_error_fields(const double* Q_dev, const double* const* k_dev, const double* coefE, int nstages,
                            double tol, int64_t nodes, int nfld, double* result_host, void* stream);

/* Whole steps (unsharded meshes).  esdg_lsrk45_step = the five stages of dg2D_euler_quad.jl:200-206 on the fused
 * RHS+stage kernels.  esdg_dopri45_attempt = stages 2..7 and the Hairer error estimate of one DOPRI45 attempt
 * (cavity_optimized.jl:1002-1021): k is an array of 7 device state buffers, k[0] = rhs(Q) on entry (FSAL); on return
 * Qtmp is the candidate state, *err_est the estimate; the caller accepts (Q <- Qtmp, swap k[0], k[6]) if it is < 1 and
 * takes the next step size from esdg_dopri45_next_dt (the P / PI controller of :1027-1033).
 * On contexts whose last phase is the line-per-lane kernel -- every 2D formulation on the tensor kernels (CNS wall meshes up to
 * N = 4, inviscid ones up to N = 6) and hexahedra; unsharded, or sharded with a communicator attached -- the stage combinations
 * and the error norm are computed inside the last-phase launch of each stage from the k_s it holds in registers (no separate
 * passes over the state; esdg_axpy_stages / esdg_dopri_error above are then not used): the same bits per node and in the estimate,
 * 4.3 instead of 5.1 ms per attempt for CNS at N=4 on 512x512.
 * Reproducibility of the estimate: the norm's terms are added in ONE order that depends on the state's shape alone (a node's
 * fields first, then runs of 4096 consecutive nodes, a fixed tree inside a run, the runs in a fixed order) -- by
 * esdg_dopri_error_fields, by the fused attempt, and whatever launches a sharded schedule cuts the last phase into.  On one rank an
 * adaptive run is therefore bitwise reproducible across the fused / separate-pass forms and across stand-alone / sharded
 * contexts; across DIFFERENT numbers of ranks the per-rank sums meet in the all-reduce, so the estimate (and with it the step
 * sizes) agrees to rounding, not bit for bit.
 * (The accept copy Q <- Qtmp may be a pointer swap on the caller's side: the library keeps no reference to either array.) */
int esdg_lsrk45_step(esdg_ctx* ctx, double* Q_dev, double* resQ_dev, double dt, void* stream);
int esdg_dopri45_attempt(esdg_ctx* ctx, const double* Q_dev, double* Qtmp_dev, double* const* k_dev, double dt, double err_tol,
                         double* err_est, void* stream);
double esdg_dopri45_next_dt(double dt, double dt0, double err_est, double prev_err_est, int64_t attempts);

/* ---- set-up for hosts without a SetupDG of their own (host-only code, no GPU needed) ------------------------------
 * esdg_setup_quad builds, for elements [e_begin, e_end) of a quad mesh (e_end <= 0: all), everything a reference
 * driver holds when it enters its time loop: RefElemData of init_reference_quad(N) with the default Gauss rule
 * (src/SetupDG.jl:205-268), MeshData of init_mesh (:271-318; connectivity is always global, mapP holds global 1-based
 * indices), optionally the periodic patch of examples/dg2D_euler_quad.jl:38-44, the driver-level operators
 * (formulation 0: dg2D_euler_quad.jl:47-91; 1/2: CompressibleNS/dg2D_CNS_cavity_optimized.jl:62-90) and the metrics
 * interpolated to the hybrid nodes.  EToV is (K x 4) column-major, 1-based, vertex order of
 * uniform_quad_mesh (src/UniformQuadMesh.jl:25-50), which esdg_setup_uniform_quad_mesh reproduces on [-1,1]^2
 * (VX, VY: (Kx+1)(Ky+1) doubles, EToV: 4*Kx*Ky int64, caller-allocated).
 * Arrays by name (column-major; rows/cols returned): r s V1 Dr Ds rf sf wf nrJ nsJ rq sq wq Vq M Pq Vf LIFT
 *   Qrhskew Qshskew Ef Vh Ph Lf|VhP   x y xf yf xq yq rxJ sxJ ryJ syJ (Nh x K) J wJq nxJ nyJ sJ;
 * maps by name (1-based): FToF mapM mapP mapB.  esdg_setup_fill points an esdg_ops_t / esdg_mesh_t pair into the
 * set-up object (self-mapped boundary nodes become walls, bkind = 1 on the y = ymax side), ready for esdg_create. */
typedef struct esdg_setup esdg_setup;
int esdg_setup_uniform_quad_mesh(int Kx, int Ky, double* VX, double* VY, int64_t* EToV);
int esdg_setup_quad(int N, int formulation, const double* VX, const double* VY, int64_t Nv, const int64_t* EToV, int64_t K,
                    int periodic, int64_t e_begin, int64_t e_end, esdg_setup** out);
const double* esdg_setup_array(const esdg_setup* s, const char* name, int64_t* rows, int64_t* cols);
const int64_t* esdg_setup_map(const esdg_setup* s, const char* name, int64_t* n);
int esdg_setup_fill(const esdg_setup* s, esdg_ops_t* ops, esdg_mesh_t* mesh);
/* Hexahedra: init_reference_hex (src/SetupDG.jl:323-387), init_mesh 3D (:389-434) with the intended hex_face_vertices
 * (DESIGN.md section 9), uniform_hex_mesh (src/UniformHexMesh.jl:25-80), the periodic patch and the operators /
 * geometry post-processing of examples/dg3D_euler_hex.jl:34-98.  EToV is (K x 8) column-major, 1-based.  Arrays as for
 * quads plus t tf tq Dt ntJ Qthskew z zf zq txJ tyJ rzJ szJ tzJ nzJ (metrics at the hybrid nodes, J at the quadrature
 * nodes). */
int esdg_setup_uniform_hex_mesh(int Kx, int Ky, int Kz, double* VX, double* VY, double* VZ, int64_t* EToV);
int esdg_setup_hex(int N, const double* VX, const double* VY, const double* VZ, int64_t Nv, const int64_t* EToV, int64_t K,
                   int periodic, int64_t e_begin, int64_t e_end, esdg_setup** out);
int esdg_setup_fill_hex(const esdg_setup* s, esdg_hex_ops_t* ops, esdg_hex_mesh_t* mesh);
int esdg_setup_destroy(esdg_setup* s);
const char* esdg_setup_last_error(void);

/* ---- plain device-memory helpers for hosts without a GPU array package (Julia ccall) ---- */
void* esdg_dmalloc(size_t bytes);
int esdg_dfree(void* p);
int esdg_memcpy_h2d(void* dst_dev, const void* src_host, size_t bytes);
int esdg_memcpy_d2h(void* dst_host, const void* src_dev, size_t bytes);
int esdg_device_synchronize(void);
int esdg_device_count(void);
/* hipSetDevice for hosts that run one process per GPU without another HIP-aware runtime (call before esdg_create) */
int esdg_set_device(int device);

#ifdef __cplusplus
}
#endif
#endif /* ESDG_HIP_H */
