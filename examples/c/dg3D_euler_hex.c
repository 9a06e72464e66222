/* examples/dg3D_euler_hex.jl on the C ABI alone: set-up (with the intended hex_face_vertices), the script's random
 * initial condition, one `rhs` evaluation and its entropy-conservation diagnostic (`@show rhstest`, :224-226).
 *
 *   gcc -O2 -I include examples/c/dg3D_euler_hex.c -o /tmp/euler_hex_c -L esdg_cns_amd -lesdg_hip -lm -Wl,-rpath,$PWD/esdg_cns_amd
 *   /tmp/euler_hex_c [N] [K1D]
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

static double urand(unsigned long long* s) {   /* splitmix64 -> [0,1) */
  unsigned long long z = (*s += 0x9e3779b97f4a7c15ULL);
  z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ULL;
  z = (z ^ (z >> 27)) * 0x94d049bb133111ebULL;
  z ^= z >> 31;
  return (double)(z >> 11) * (1.0 / 9007199254740992.0);
}

int main(int argc, char** argv) {
  const int N = argc > 1 ? atoi(argv[1]) : 2, K1D = argc > 2 ? atoi(argv[2]) : 8;
  const int64_t K = (int64_t)K1D * K1D * K1D, Nv = (int64_t)(K1D + 1) * (K1D + 1) * (K1D + 1);
  double *VX = malloc(Nv * sizeof(double)), *VY = malloc(Nv * sizeof(double)), *VZ = malloc(Nv * sizeof(double));
  int64_t* EToV = malloc(8 * K * sizeof(int64_t));
  CHECK(esdg_setup_uniform_hex_mesh(K1D, K1D, K1D, VX, VY, VZ, EToV));      /* dg3D_euler_hex.jl:26 */
  esdg_setup* S = NULL;
  CHECK(esdg_setup_hex(N, VX, VY, VZ, Nv, EToV, K, 1, 0, 0, &S));            /* :31-98 */
  esdg_hex_ops_t ops; esdg_hex_mesh_t mesh;
  CHECK(esdg_setup_fill_hex(S, &ops, &mesh));
  esdg_phys_t ph;
  memset(&ph, 0, sizeof ph);
  ph.formulation = ESDG_EULER_HEX_COLLOCATED; ph.lf_scale = 0.0;              /* LFc = 0*.25*..., :193 */
  esdg_ctx* ctx = NULL;
  CHECK(esdg_create_hex(&ops, &mesh, &ph, &ctx));
  const size_t wsb = esdg_workspace_bytes(ctx);
  void* ws = esdg_dmalloc(wsb);
  CHECK(esdg_bind_workspace(ctx, ws, wsb));
  const size_t n = (size_t)K * ops.Nq, bytes = 5 * n * sizeof(double);
  double* Q = malloc(bytes);
  unsigned long long seed = 20250117ULL;
  for (size_t i = 0; i < n; ++i) {                                           /* :101-110 */
    const double rho = 2 + .1 * urand(&seed), p = 1 + .1 * urand(&seed), u = 0, v = 1, w = 0;
    Q[i] = rho; Q[n + i] = rho * u; Q[2 * n + i] = rho * v; Q[3 * n + i] = rho * w;
    Q[4 * n + i] = p / 0.4 + .5 * rho * (u * u + v * v + w * w);
  }
  double *Qd = esdg_dmalloc(bytes), *rhsd = esdg_dmalloc(bytes);
  CHECK(esdg_memcpy_h2d(Qd, Q, bytes));
  double diag[2];
  CHECK(esdg_rhs(ctx, Qd, rhsd, NULL));                                      /* :224 */
  CHECK(esdg_rhstest(ctx, Qd, rhsd, diag, NULL));
  CHECK(esdg_memcpy_d2h(Q, rhsd, bytes));
  double amax = 0;
  for (size_t i = 0; i < 5 * n; ++i) amax = fmax(amax, fabs(Q[i]));
  printf("N=%d K=%d^3 fields=%d rhstest=%.6e max|rhs|=%.6e\n", N, K1D, esdg_num_fields(ctx), diag[0], amax);
  esdg_dfree(Qd); esdg_dfree(rhsd); esdg_dfree(ws);
  esdg_destroy(ctx); esdg_setup_destroy(S);
  free(VX); free(VY); free(VZ); free(EToV); free(Q);
  return 0;
}
