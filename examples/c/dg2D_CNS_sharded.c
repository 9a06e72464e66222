/* Sharded CNS right-hand side on the C ABI alone (no Python, no torch, no Julia): element-index strips, one process per
 * GPU, the halo exchange over the library's own RCCL transport (esdg_comm_init / esdg_rhs on a sharded context).
 * The reference has no counterpart (single process); the per-rank work is rhsRK! of
 * examples/CompressibleNS/dg2D_CNS_cavity_optimized.jl:955-972 on the reference quad element.
 *
 *   gcc -O2 -I include examples/c/dg2D_CNS_sharded.c -o /tmp/cns_sharded_c -L esdg_cns_amd -lesdg_hip -lm \
 *       -Wl,-rpath,$PWD/esdg_cns_amd
 *   /tmp/cns_sharded_c 1 [N] [Kx] [Ky_per_rank]    one GPU: rank 0's strip of an 8-rank mesh, communicator in loopback,
 *                                                  compared bit for bit with the same strip as a stand-alone periodic mesh
 *   /tmp/cns_sharded_c R [N] [Kx] [Ky_per_rank]    R >= 2 GPUs: forks R ranks (rank r on device r), bootstraps the
 *                                                  ncclUniqueId through a file, prints all-reduced checksums and timing
 */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/time.h>
#include <sys/wait.h>
#include <unistd.h>

#include "esdg_hip.h"

#define CHECK(call)                                                                                        \
  do {                                                                                                     \
    int rc_ = (call);                                                                                      \
    if (rc_ != 0) {                                                                                        \
      fprintf(stderr, "%s failed (%d): %s %s\n", #call, rc_, esdg_last_error(), esdg_setup_last_error()); \
      return 1;                                                                                            \
    }                                                                                                      \
  } while (0)

static const double GAMMA = 1.4;

/* smooth state, period 15 in x and LY in y: every strip of the box is periodic on its own */
static void state(double x, double y, double y0, double LY, double* U) {
  const double cx = 2 * M_PI * x / 15.0, cy = 2 * M_PI * (y - y0) / LY;
  const double rho = 1 + .2 * sin(cx + .3) * cos(cy + .1), u = .4 + .1 * cos(cx - .2) * sin(cy + .4);
  const double v = -.3 + .1 * sin(cx + .5) * sin(cy - .3), p = 1 + .15 * cos(cx + .7) * cos(cy + .2);
  U[0] = rho; U[1] = rho * u; U[2] = rho * v; U[3] = p / (GAMMA - 1) + .5 * rho * (u * u + v * v);
}

typedef struct {
  esdg_setup* S;
  esdg_ctx* ctx;
  void* ws;
  double *Qd, *rhsd, *Q;
  const double* wJq;
  int64_t K;
  int Np;
} shard_t;

/* elements [e0, e1) of the Kx x Ky_total periodic box [0,15] x [-5 s, 5 s] (square elements), rank `rank` of `nranks` */
static int build_shard(int N, int Kx, int Ky_total, int Kyr, int rank, int nranks, int64_t e0, int64_t e1, int64_t* offsets,
                       shard_t* sh) {
  const int64_t Kg = (int64_t)Kx * Ky_total, Nv = (int64_t)(Kx + 1) * (Ky_total + 1);
  double *VX = malloc(Nv * sizeof(double)), *VY = malloc(Nv * sizeof(double));
  int64_t* EToV = malloc(4 * Kg * sizeof(int64_t));
  CHECK(esdg_setup_uniform_quad_mesh(Kx, Ky_total, VX, VY, EToV));
  const double sy = 5.0 * Ky_total / Kx;
  for (int64_t i = 0; i < Nv; ++i) { VX[i] = 15 * (1 + VX[i]) / 2; VY[i] = sy * VY[i]; }
  CHECK(esdg_setup_quad(N, ESDG_CNS_MODAL, VX, VY, Nv, EToV, Kg, 1, e0, e1, &sh->S));
  esdg_ops_t ops; esdg_mesh_t mesh;
  CHECK(esdg_setup_fill(sh->S, &ops, &mesh));
  mesh.NmapB = 0; mesh.mapB = NULL; mesh.bkind = NULL;           /* fully periodic: no walls */
  mesh.rank = rank; mesh.nranks = nranks; mesh.rank_offsets = nranks > 1 ? offsets : NULL;
  esdg_phys_t ph;
  memset(&ph, 0, sizeof ph);
  ph.formulation = ESDG_CNS_MODAL; ph.lf_scale = 0.25; ph.inviscid_dissp = 1; ph.viscous_dissp = 1; ph.BCTYPE = 1;
  ph.Re = 1000.0; ph.mu = 1e-3; ph.lambda = -2e-3 / 3; ph.Pr = 0.71;      /* cavity_optimized.jl:33-36 */
  CHECK(esdg_create(&ops, &mesh, &ph, &sh->ctx));
  const size_t wsb = esdg_workspace_bytes(sh->ctx);
  sh->ws = esdg_dmalloc(wsb);
  CHECK(esdg_bind_workspace(sh->ctx, sh->ws, wsb));
  sh->K = mesh.K; sh->Np = ops.Np;
  const size_t n = (size_t)sh->K * sh->Np, bytes = 4 * n * sizeof(double);
  int64_t r, c;
  const double *x = esdg_setup_array(sh->S, "x", &r, &c), *y = esdg_setup_array(sh->S, "y", &r, &c);
  sh->wJq = esdg_setup_array(sh->S, "wJq", &r, &c);
  sh->Q = malloc(bytes);
  const double LY = 10.0 * Kyr / Kx;
  for (size_t i = 0; i < n; ++i) {
    double U[4];
    state(x[i], y[i], -sy, LY, U);
    for (int f = 0; f < 4; ++f) sh->Q[f * n + i] = U[f];
  }
  sh->Qd = esdg_dmalloc(bytes); sh->rhsd = esdg_dmalloc(bytes);
  CHECK(esdg_memcpy_h2d(sh->Qd, sh->Q, bytes));
  free(VX); free(VY); free(EToV);
  return 0;
}

static double now(void) {
  struct timeval tv;
  gettimeofday(&tv, NULL);
  return tv.tv_sec + 1e-6 * tv.tv_usec;
}

static int run_rank(int rank, int nranks, int N, int Kx, int Kyr, const char* idfile) {
  CHECK(esdg_set_device(rank % esdg_device_count()));
  int64_t* offsets = malloc((nranks + 1) * sizeof(int64_t));
  for (int r = 0; r <= nranks; ++r) offsets[r] = (int64_t)Kx * Kyr * r;
  shard_t sh;
  memset(&sh, 0, sizeof sh);
  if (build_shard(N, Kx, Kyr * nranks, Kyr, rank, nranks, offsets[rank], offsets[rank + 1], offsets, &sh)) return 1;
  unsigned char id[ESDG_COMM_ID_BYTES];
  if (rank == 0) {                                     /* bootstrap: rank 0 writes the id, the others wait for the file */
    CHECK(esdg_comm_unique_id(id));
    char tmp[512];
    snprintf(tmp, sizeof tmp, "%s.tmp", idfile);
    FILE* f = fopen(tmp, "wb");
    if (!f || fwrite(id, 1, sizeof id, f) != sizeof id) return 1;
    fclose(f);
    rename(tmp, idfile);
  } else {
    FILE* f = NULL;
    for (int t = 0; t < 6000 && !(f = fopen(idfile, "rb")); ++t) usleep(10000);
    if (!f || fread(id, 1, sizeof id, f) != sizeof id) { fprintf(stderr, "rank %d: no id file\n", rank); return 1; }
    fclose(f);
  }
  CHECK(esdg_comm_init(sh.ctx, id, rank, nranks));
  const int steps = 50;
  for (int i = 0; i < 10; ++i) CHECK(esdg_rhs(sh.ctx, sh.Qd, sh.rhsd, NULL));
  CHECK(esdg_device_synchronize());
  double one = 1.0;
  CHECK(esdg_comm_allreduce(sh.ctx, &one, 1, 0, NULL));                     /* barrier */
  const double t0 = now();
  for (int i = 0; i < steps; ++i) CHECK(esdg_rhs(sh.ctx, sh.Qd, sh.rhsd, NULL));
  CHECK(esdg_device_synchronize());
  double dt = now() - t0;
  CHECK(esdg_comm_allreduce(sh.ctx, &dt, 1, 1, NULL));                      /* max over ranks */
  const size_t n = (size_t)sh.K * sh.Np;
  double* rhs = malloc(4 * n * sizeof(double));
  CHECK(esdg_memcpy_d2h(rhs, sh.rhsd, 4 * n * sizeof(double)));
  double sums[5] = {0, 0, 0, 0, 0};
  for (int f = 0; f < 4; ++f)
    for (size_t i = 0; i < n; ++i) { sums[f] += rhs[f * n + i]; sums[4] += fabs(rhs[f * n + i]); }
  CHECK(esdg_comm_allreduce(sh.ctx, sums, 5, 0, NULL));
  if (rank == 0)
    printf("ranks=%d (RCCL comm size %d) N=%d mesh=%dx%d  %.4f ms/RHS  %.4e DOF updates/s  sum|rhs|=%.15e\n", nranks,
           esdg_comm_size(sh.ctx), N, Kx, Kyr * nranks, dt / steps * 1e3, (double)Kx * Kyr * nranks * sh.Np * steps / dt, sums[4]);
  esdg_comm_destroy(sh.ctx);
  esdg_destroy(sh.ctx);
  return 0;
}

/* The CNS drivers' time loop (dg2D_CNS_cavity_optimized.jl:997-1037) on the C ABI: adaptive DOPRI45 with FSAL; the accept "copy"
 * is a swap of the two state pointers, the k[0] <-> k[6] exchange a swap of two entries.  On return sh->Qd is the state at *t. */
static int dopri_loop(shard_t* sh, int attempts, double dt0, double tol, double* t, double* last_err) {
  const size_t bytes = 4 * (size_t)sh->K * sh->Np * sizeof(double);
  double* k[7];
  for (int i = 0; i < 7; ++i) k[i] = esdg_dmalloc(bytes);
  double *Q = sh->Qd, *Qtmp = esdg_dmalloc(bytes);
  CHECK(esdg_rhs(sh->ctx, Q, k[0], NULL));
  double dt = dt0, prev = 0.0;
  *t = 0.0;
  for (int i = 0; i < attempts; ++i) {
    double err = 0.0;
    CHECK(esdg_dopri45_attempt(sh->ctx, Q, Qtmp, k, dt, tol, &err, NULL));
    if (err < 1.0) {
      double* s = Q; Q = Qtmp; Qtmp = s;
      s = k[0]; k[0] = k[6]; k[6] = s;
      *t += dt;
    }
    dt = esdg_dopri45_next_dt(dt, dt0, err, prev, i);
    prev = err;
    *last_err = err;
  }
  CHECK(esdg_device_synchronize());
  sh->Qd = Q;
  return 0;
}

static int run_loopback(int N, int Kx, int Kyr) {
  const int nr = 8;
  int64_t offsets[9];
  for (int r = 0; r <= nr; ++r) offsets[r] = (int64_t)Kx * Kyr * r;
  shard_t sh, one;
  memset(&sh, 0, sizeof sh); memset(&one, 0, sizeof one);
  if (build_shard(N, Kx, Kyr * nr, Kyr, 0, nr, 0, offsets[1], offsets, &sh)) return 1;     /* rank 0 of 8 */
  if (build_shard(N, Kx, Kyr, Kyr, 0, 1, 0, offsets[1], NULL, &one)) return 1;             /* the strip on its own */
  unsigned char id[ESDG_COMM_ID_BYTES];
  CHECK(esdg_comm_set_loopback(sh.ctx, 1));
  CHECK(esdg_comm_unique_id(id));
  CHECK(esdg_comm_init(sh.ctx, id, 0, 1));
  const size_t n = (size_t)sh.K * sh.Np, bytes = 4 * n * sizeof(double);
  /* the two set-ups see different global meshes: feed both engines the same state */
  CHECK(esdg_memcpy_h2d(one.Qd, sh.Q, bytes));
  double *a = malloc(bytes), *b = malloc(bytes), maxd = 0, maxv = 0;
  for (int it = 0; it < 3; ++it) {
    CHECK(esdg_rhs(sh.ctx, sh.Qd, sh.rhsd, NULL));
    CHECK(esdg_rhs(one.ctx, one.Qd, one.rhsd, NULL));
  }
  CHECK(esdg_device_synchronize());
  CHECK(esdg_memcpy_d2h(a, sh.rhsd, bytes));
  CHECK(esdg_memcpy_d2h(b, one.rhsd, bytes));
  for (size_t i = 0; i < 4 * n; ++i) { maxd = fmax(maxd, fabs(a[i] - b[i])); maxv = fmax(maxv, fabs(b[i])); }
  printf("loopback rank 0 of %d (RCCL comm size %d) N=%d strip=%dx%d: max|rhs_sharded - rhs_standalone| = %.3e (max|rhs| %.3e) %s\n",
         nr, esdg_comm_size(sh.ctx), N, Kx, Kyr, maxd, maxv, maxd <= 1e-11 * maxv ? "OK" : "MISMATCH");
  /* ... and six attempted DOPRI45 steps on both: stage combinations and error norm ride in the last phase of every stage
   * (on the sharded context: in each piece of its overlapped schedule) */
  double ta = 0, tb = 0, ea = 0, eb = 0, maxq = 0;
  const double dt0 = 0.5 * (15.0 / Kx) / ((N + 1) * (N + 2) / 2.0);
  if (dopri_loop(&sh, 6, dt0, 1e-5, &ta, &ea) || dopri_loop(&one, 6, dt0, 1e-5, &tb, &eb)) return 1;
  CHECK(esdg_memcpy_d2h(a, sh.Qd, bytes));
  CHECK(esdg_memcpy_d2h(b, one.Qd, bytes));
  double maxa = 0;
  for (size_t i = 0; i < 4 * n; ++i) { maxq = fmax(maxq, fabs(a[i] - b[i])); maxa = fmax(maxa, fabs(b[i])); }
  /* (the two set-ups' geometry differs in the last bits, see above: agreement to that level, not bitwise) */
  const int ok2 = maxq <= 1e-9 * maxa && fabs(ta - tb) <= 1e-9 * tb && ta > 0 && fabs(ea - eb) <= 1e-6 * eb;
  printf("DOPRI45, 6 attempts on both: t = %.6e / %.6e, last errEst %.6e / %.6e, max|Q_sharded - Q_standalone| = %.3e %s\n", ta, tb, ea, eb,
         maxq, ok2 ? "OK" : "MISMATCH");
  esdg_comm_destroy(sh.ctx);
  esdg_destroy(sh.ctx); esdg_destroy(one.ctx);
  return maxd <= 1e-11 * maxv && ok2 ? 0 : 2;
}

int main(int argc, char** argv) {
  const int R = argc > 1 ? atoi(argv[1]) : 1, N = argc > 2 ? atoi(argv[2]) : 4;
  const int Kx = argc > 3 ? atoi(argv[3]) : 64, Kyr = argc > 4 ? atoi(argv[4]) : 8;
  if (R <= 1) return run_loopback(N, Kx, Kyr);
  char idfile[256];
  snprintf(idfile, sizeof idfile, "/tmp/esdg_nccl_id_%d", (int)getpid());
  unlink(idfile);
  /* one process per GPU; fork BEFORE anything touches the GPU */
  pid_t* pids = malloc(R * sizeof(pid_t));
  for (int r = 0; r < R; ++r) {
    pids[r] = fork();
    if (pids[r] == 0) _exit(run_rank(r, R, N, Kx, Kyr, idfile));
  }
  int bad = 0;
  for (int r = 0; r < R; ++r) {
    int st = 0;
    waitpid(pids[r], &st, 0);
    bad |= !(WIFEXITED(st) && WEXITSTATUS(st) == 0);
  }
  unlink(idfile);
  return bad;
}
