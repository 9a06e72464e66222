// LDS read-rate micro-benchmark for the layout decisions of esdg_kernels_tensor2.hip:
//   mode 0: ds_read_b64 singles (runtime stride blocks the compiler's read2 merging)
//   mode 1: ds_read2_b64 (compile-time offsets, merged pairs)
//   mode 2: ds_read_b128 (16-byte aligned pairs)
//   mode 3: ds_read2_b64 with near offsets (the plain, non-st64 form)
// Every lane reads consecutive 8-byte slots (conflict-free).  Prints LDS bytes per clock per CU (2.4 GHz assumed).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
template <int MODE>
__global__ __launch_bounds__(256) void k(double* out, int iters, int rs) {
  __shared__ __align__(16) double s[8192];
  unsigned t = threadIdx.x;
  for (int i = t; i < 8192; i += 256) s[i] = i * 0.5;
  __syncthreads();
  double a0 = 0, a1 = 0, a2 = 0, a3 = 0;
  unsigned base = (t & 63) * (MODE == 2 ? 2 : 1) + (t >> 6) * 1024;
  for (int it = 0; it < iters; ++it) {
    if (MODE == 0) {
      const double* p = s + base + (it & 7);
      a0 += p[0 * rs]; a1 += p[128 * rs]; a2 += p[256 * rs]; a3 += p[384 * rs];
      a0 += p[512 * rs]; a1 += p[640 * rs]; a2 += p[768 * rs]; a3 += p[896 * rs];
    } else if (MODE == 1) {
      const double* p = s + base + (it & 7);
      a0 += p[0]; a1 += p[128]; a2 += p[256]; a3 += p[384];
      a0 += p[512]; a1 += p[640]; a2 += p[768]; a3 += p[896];
    } else if (MODE == 3) {
      const double* p = s + base + (it & 7);
      a0 += p[0]; a1 += p[64]; a2 += p[128]; a3 += p[192];
      a0 += p[256]; a1 += p[320]; a2 += p[384]; a3 += p[448];
    } else {
      const double2* p = reinterpret_cast<const double2*>(s + base + 2 * (it & 31));
      double2 v0 = p[0], v1 = p[64], v2 = p[128], v3 = p[192];
      a0 += v0.x; a1 += v0.y; a2 += v1.x; a3 += v1.y; a0 += v2.x; a1 += v2.y; a2 += v3.x; a3 += v3.y;
    }
  }
  out[blockIdx.x * 256 + t] = a0 + a1 + a2 + a3;
}
int main() {
  double* out;
  hipMalloc(&out, 256 * 4096 * sizeof(double));
  const int iters = 4000, nb = 256 * 8;
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  const char* names[4] = {"ds_read_b64 singles", "ds_read2st64_b64", "ds_read_b128", "ds_read2_b64"};
  for (int m = 0; m < 4; ++m) {
    for (int rep = 0; rep < 2; ++rep) {
      hipEventRecord(e0);
      if (m == 0) hipLaunchKernelGGL(k<0>, dim3(nb), dim3(256), 0, 0, out, iters, 1);
      if (m == 1) hipLaunchKernelGGL(k<1>, dim3(nb), dim3(256), 0, 0, out, iters, 1);
      if (m == 2) hipLaunchKernelGGL(k<2>, dim3(nb), dim3(256), 0, 0, out, iters, 1);
      if (m == 3) hipLaunchKernelGGL(k<3>, dim3(nb), dim3(256), 0, 0, out, iters, 1);
      hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      double bytes = (double)nb * 256 * iters * 64.0;
      if (rep) printf("%-22s %.3f ms  %.1f TB/s  %.1f B/clk/CU @2.4GHz\n", names[m], ms, bytes / ms / 1e9, bytes / (ms * 1e-3) / 256 / 2.4e9);
    }
  }
  return 0;
}
