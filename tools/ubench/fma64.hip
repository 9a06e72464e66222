// fp64 VALU issue / latency micro-benchmark behind the occupancy and MFMA decisions of DESIGN.md section 4:
//   * C independent v_fma_f64 chains per wave (C = 1, 2, 4, 8), W waves per SIMD (W = 1..4): cycles per FMA per SIMD
//   * the same flops as v_mfma_f64_16x16x4_f64 (one chain / four chains)
// One workgroup of 64 * 4 * W lanes per CU (W waves on each of the 4 SIMDs), 256 workgroups.  2.4 GHz assumed.
#include <hip/hip_runtime.h>
#include <cstdio>

template <int C>
__global__ void k_fma(double* out, int iters, double a, double b) {
  double x[C];
#pragma unroll
  for (int c = 0; c < C; ++c) x[c] = threadIdx.x * 1e-9 + c;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 8; ++u) {
#pragma unroll
      for (int c = 0; c < C; ++c) x[c] = __builtin_fma(x[c], a, b);
    }
  }
  double s = 0;
#pragma unroll
  for (int c = 0; c < C; ++c) s += x[c];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

typedef double d4 __attribute__((ext_vector_type(4)));
template <int C>
__global__ void k_mfma(double* out, int iters, double a, double b) {
  d4 acc[C];
#pragma unroll
  for (int c = 0; c < C; ++c) acc[c] = d4{0, 0, 0, 0};
  const double av = a + threadIdx.x * 1e-9, bv = b;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 8; ++u) {
#pragma unroll
      for (int c = 0; c < C; ++c) acc[c] = __builtin_amdgcn_mfma_f64_16x16x4f64(av, bv, acc[c], 0, 0, 0);
    }
  }
  double s = 0;
#pragma unroll
  for (int c = 0; c < C; ++c) s += acc[c][0] + acc[c][1] + acc[c][2] + acc[c][3];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <typename K>
static void run(const char* name, K kern, int chains, int waves_per_simd, double flops_per_inst, double* out) {
  const int iters = 2000, nb = 256, threads = 64 * 4 * waves_per_simd;
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  float ms = 0;
  for (int rep = 0; rep < 2; ++rep) {
    hipEventRecord(e0);
    hipLaunchKernelGGL(kern, dim3(nb), dim3(threads), 0, 0, out, iters, 1.0000001, 1e-9);
    hipEventRecord(e1); hipEventSynchronize(e1);
    hipEventElapsedTime(&ms, e0, e1);
  }
  const double inst_per_simd = (double)iters * 8 * chains * waves_per_simd;
  const double cyc = ms * 1e-3 * 2.4e9 / inst_per_simd;
  const double tflops = inst_per_simd * 1024 * flops_per_inst / (ms * 1e-3) / 1e12;
  printf("%-10s chains %d  waves/SIMD %d : %.3f ms  %.2f cycles per wave-instruction per SIMD  %.1f TFLOP/s\n", name, chains, waves_per_simd, ms,
         cyc, tflops);
}

int main() {
  double* out;
  hipMalloc(&out, 256 * 1024 * sizeof(double));
  for (int w = 1; w <= 4; ++w) {
    run("v_fma_f64", k_fma<1>, 1, w, 128.0, out);
    run("v_fma_f64", k_fma<2>, 2, w, 128.0, out);
    run("v_fma_f64", k_fma<4>, 4, w, 128.0, out);
    run("v_fma_f64", k_fma<8>, 8, w, 128.0, out);
  }
  for (int w = 1; w <= 2; ++w) {
    run("mfma_f64", k_mfma<1>, 1, w, 2.0 * 16 * 16 * 4, out);
    run("mfma_f64", k_mfma<4>, 4, w, 2.0 * 16 * 16 * 4, out);
  }
  return 0;
}
