/* examples/dg2D_euler_quad.jl on the C ABI alone (no Python, no Julia): set-up, isentropic vortex, LSRK45 time loop
 * with the fused RHS+stage entry point, L2 error at the Gauss nodes.
 *
 *   gcc -O2 -I include examples/c/dg2D_euler_quad.c -o /tmp/euler_quad_c -L esdg_cns_amd -lesdg_hip -lm \
 *       -Wl,-rpath,$PWD/esdg_cns_amd
 *   /tmp/euler_quad_c [N] [K1D] [T]
 */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "esdg_hip.h"

#define CHECK(call)                                                                     \
  do {                                                                                  \
    int rc_ = (call);                                                                   \
    if (rc_ != 0) {                                                                     \
      fprintf(stderr, "%s failed (%d): %s %s\n", #call, rc_, esdg_last_error(), esdg_setup_last_error()); \
      return 1;                                                                         \
    }                                                                                   \
  } while (0)

static const double GAMMA = 1.4;

/* EntropyStableEuler.jl:21-35 */
static void vortex(double x, double y, double t, double* rho, double* u, double* v, double* p) {
  const double x0 = 5, y0 = 0, beta = 5;
  const double r2 = (x - x0 - t) * (x - x0 - t) + (y - y0) * (y - y0);
  *u = 1 - beta * exp(1 - r2) * (y - y0) / (2 * M_PI);
  *v = beta * exp(1 - r2) * (x - x0 - t) / (2 * M_PI);
  const double b = beta * exp(1 - r2);
  *rho = pow(1 - (1 / (8 * GAMMA * M_PI * M_PI)) * (GAMMA - 1) / 2 * b * b, 1 / (GAMMA - 1));
  *p = pow(*rho, GAMMA);
}

int main(int argc, char** argv) {
  const int N = argc > 1 ? atoi(argv[1]) : 2, K1D = argc > 2 ? atoi(argv[2]) : 12;
  const double T = argc > 3 ? atof(argv[3]) : 1.0, CFL = 2.0;
  const int Kx = 4 * K1D / 3, Ky = K1D;
  const int64_t K = (int64_t)Kx * Ky, Nv = (int64_t)(Kx + 1) * (Ky + 1);
  double *VX = malloc(Nv * sizeof(double)), *VY = malloc(Nv * sizeof(double));
  int64_t* EToV = malloc(4 * K * sizeof(int64_t));
  CHECK(esdg_setup_uniform_quad_mesh(Kx, Ky, VX, VY, EToV));               /* dg2D_euler_quad.jl:26-31 */
  for (int64_t i = 0; i < Nv; ++i) { VX[i] = 15 * (1 + VX[i]) / 2; VY[i] = 5 * VY[i]; }
  esdg_setup* S = NULL;
  CHECK(esdg_setup_quad(N, ESDG_EULER_COLLOCATED, VX, VY, Nv, EToV, K, 1, 0, 0, &S));   /* :35-91 */
  esdg_ops_t ops; esdg_mesh_t mesh;
  CHECK(esdg_setup_fill(S, &ops, &mesh));
  esdg_phys_t ph;
  memset(&ph, 0, sizeof ph);
  ph.formulation = ESDG_EULER_COLLOCATED; ph.lf_scale = 0.5; ph.inviscid_dissp = 1;
  esdg_ctx* ctx = NULL;
  CHECK(esdg_create(&ops, &mesh, &ph, &ctx));
  const size_t wsb = esdg_workspace_bytes(ctx);
  void* ws = esdg_dmalloc(wsb);
  CHECK(esdg_bind_workspace(ctx, ws, wsb));

  const int Nq = ops.Nq;
  const size_t n = (size_t)K * Nq, bytes = 4 * n * sizeof(double);
  int64_t r, c;
  const double *xq = esdg_setup_array(S, "xq", &r, &c), *yq = esdg_setup_array(S, "yq", &r, &c);
  const double* wJq = esdg_setup_array(S, "wJq", &r, &c);
  double* Q = malloc(bytes);
  for (size_t i = 0; i < n; ++i) {                                           /* :81-83 */
    double rho, u, v, p;
    vortex(xq[i], yq[i], 0.0, &rho, &u, &v, &p);
    Q[i] = rho; Q[n + i] = rho * u; Q[2 * n + i] = rho * v; Q[3 * n + i] = p / (GAMMA - 1) + .5 * rho * (u * u + v * v);
  }
  double *Qd = esdg_dmalloc(bytes), *resd = esdg_dmalloc(bytes), *rhsd = esdg_dmalloc(bytes);
  double* zero = calloc(4 * n, sizeof(double));
  CHECK(esdg_memcpy_h2d(Qd, Q, bytes));
  CHECK(esdg_memcpy_h2d(resd, zero, bytes));

  /* rk45_coeffs, src/CommonUtils.jl:29-49 */
  const double rk4a[5] = {0.0, -567301805773.0 / 1357537059087.0, -2404267990393.0 / 2016746695238.0,
                          -3550918686646.0 / 2091501179385.0, -1275806237668.0 / 842570457699.0};
  const double rk4b[5] = {1432997174477.0 / 9575080441755.0, 5161836677717.0 / 13612068292357.0,
                          1720146321549.0 / 2090206949498.0, 3134564353537.0 / 4481467310338.0,
                          2277821191437.0 / 14882151754819.0};
  const double CN = (N + 1) * (N + 2) / 2.0, h = 2.0 / K1D;                   /* :93-99 */
  double dt = CFL * h / CN;
  const int Nsteps = (int)ceil(T / dt);
  dt = T / Nsteps;
  for (int i = 0; i < Nsteps; ++i)                                            /* :196-212 */
    for (int k = 0; k < 5; ++k) CHECK(esdg_rhs_lsrk(ctx, Qd, resd, rk4a[k], rk4b[k], dt, NULL));
  double diag[2];
  CHECK(esdg_rhs(ctx, Qd, rhsd, NULL));
  CHECK(esdg_rhstest(ctx, Qd, rhsd, diag, NULL));
  CHECK(esdg_memcpy_d2h(Q, Qd, bytes));

  double err2 = 0, sum = 0;
  for (size_t i = 0; i < n; ++i) {
    double rho, u, v, p;
    vortex(xq[i], yq[i], T, &rho, &u, &v, &p);
    const double ex[4] = {rho, rho * u, rho * v, p / (GAMMA - 1) + .5 * rho * (u * u + v * v)};
    for (int f = 0; f < 4; ++f) {
      const double d = Q[f * n + i] - ex[f];
      err2 += wJq[i] * d * d;
      sum += wJq[i] * Q[f * n + i];
    }
  }
  printf("N=%d K=%dx%d steps=%d rhstest=%.6e L2err_gauss=%.12e integral=%.15e\n", N, Kx, Ky, Nsteps, diag[0], sqrt(err2), sum);
  esdg_dfree(Qd); esdg_dfree(resd); esdg_dfree(rhsd); esdg_dfree(ws);
  esdg_destroy(ctx); esdg_setup_destroy(S);
  free(VX); free(VY); free(EToV); free(Q); free(zero);
  return 0;
}
