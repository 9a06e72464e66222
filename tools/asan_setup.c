/* Sanitizer run of the host-only set-up code (csrc/esdg_setup.cpp), CPU build only (GPU ASan is not available on the
 * pool):   g++ -fsanitize=address,undefined -g -Iinclude tools/asan_setup.c esdg_cns_amd/csrc/esdg_setup.cpp -o /tmp/asan_setup
 * Exercises quads (all formulations, periodic and walls, element ranges) and hexes, then frees everything. */
#include <stdio.h>
#include <stdlib.h>

#include "esdg_hip.h"

static int quad(int N, int form, int Kx, int Ky, int periodic, long e0, long e1) {
  const long Nv = (long)(Kx + 1) * (Ky + 1), K = (long)Kx * Ky;
  double* VX = (double*)malloc(sizeof(double) * Nv);
  double* VY = (double*)malloc(sizeof(double) * Nv);
  int64_t* E = (int64_t*)malloc(sizeof(int64_t) * 4 * K);
  int rc = esdg_setup_uniform_quad_mesh(Kx, Ky, VX, VY, E);
  esdg_setup* s = NULL;
  if (!rc) rc = esdg_setup_quad(N, form, VX, VY, Nv, E, K, periodic, e0, e1, &s);
  if (!rc) {
    esdg_ops_t ops; esdg_mesh_t mesh;
    rc = esdg_setup_fill(s, &ops, &mesh);
    int64_t r, c, n;
    const double* J = esdg_setup_array(s, "J", &r, &c);
    const int64_t* mp = esdg_setup_map(s, "mapP", &n);
    double sum = 0; long long ms = 0;
    for (int64_t i = 0; J && i < r * c; ++i) sum += J[i];
    for (int64_t i = 0; mp && i < n; ++i) ms += mp[i];
    printf("quad N=%d form=%d %dx%d per=%d [%ld,%ld): K=%lld sumJ=%.6f sum(mapP)=%lld\n", N, form, Kx, Ky, periodic, e0, e1,
           (long long)mesh.K, sum, ms);
  } else {
    printf("quad N=%d: %s\n", N, esdg_setup_last_error());
  }
  esdg_setup_destroy(s);
  free(VX); free(VY); free(E);
  return rc;
}

static int hex(int N, int Kx, int Ky, int Kz, int periodic, long e0, long e1) {
  const long Nv = (long)(Kx + 1) * (Ky + 1) * (Kz + 1), K = (long)Kx * Ky * Kz;
  double* V[3];
  for (int i = 0; i < 3; ++i) V[i] = (double*)malloc(sizeof(double) * Nv);
  int64_t* E = (int64_t*)malloc(sizeof(int64_t) * 8 * K);
  int rc = esdg_setup_uniform_hex_mesh(Kx, Ky, Kz, V[0], V[1], V[2], E);
  esdg_setup* s = NULL;
  if (!rc) rc = esdg_setup_hex(N, V[0], V[1], V[2], Nv, E, K, periodic, e0, e1, &s);
  if (!rc) {
    esdg_hex_ops_t ops; esdg_hex_mesh_t mesh;
    rc = esdg_setup_fill_hex(s, &ops, &mesh);
    printf("hex N=%d %dx%dx%d per=%d [%ld,%ld): K=%lld\n", N, Kx, Ky, Kz, periodic, e0, e1, (long long)mesh.K);
  } else {
    printf("hex N=%d: %s\n", N, esdg_setup_last_error());
  }
  esdg_setup_destroy(s);
  for (int i = 0; i < 3; ++i) free(V[i]);
  free(E);
  return rc;
}

int main(void) {
  int bad = 0;
  for (int N = 1; N <= 6; ++N)
    for (int form = 0; form <= 2; ++form) bad |= quad(N, form, 4 + N % 3, 3 + N % 2, 1, 0, 0);
  bad |= quad(3, 1, 5, 5, 0, 0, 0);      /* walls */
  bad |= quad(4, 1, 6, 8, 1, 12, 30);    /* element range of a sharded mesh */
  bad |= quad(2, 0, 16, 16, 1, 0, 0);
  for (int N = 1; N <= 3; ++N) bad |= hex(N, 3, 2 + N % 2, 4, 1, 0, 0);
  bad |= hex(2, 4, 4, 4, 1, 16, 48);
  /* argument errors must be reported, not crash */
  esdg_setup* s = NULL;
  if (esdg_setup_quad(0, 0, NULL, NULL, 0, NULL, 0, 1, 0, 0, &s) == 0) bad = 1;
  printf("error path: %s\n", esdg_setup_last_error());
  printf(bad ? "FAILED\n" : "OK\n");
  return bad;
}
