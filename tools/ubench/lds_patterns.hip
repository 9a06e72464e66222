// LDS cost of the access patterns of kt2_sigma (N1 = 5, 128 lanes = 5 elements): cycles of LDS time per wave-instruction for
//   0 b128 contiguous (lane -> slot lane)                       4 b64  contiguous
//   1 b128 row walk    slot = e*25 + 5*b + j  (5 lanes share)    5 b64  row walk
//   2 b128 column walk slot = e*25 + a + 5*j  (lanes of a column share, 5 distinct per element)
//   3 b128 lift gather slot = e*20 + face node of the lane's line end (5 lanes share)
//   6 b64  element broadcast slot = e*17 + k (25 lanes share)
// Each workgroup = 128 threads as in the kernel; 8 waves per CU resident; the loop issues 10 reads per iteration (j = 0..4 twice).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
template <int W>   // W = 16: b128, 8: b64
__global__ __launch_bounds__(128) void k(const int* __restrict__ base, const int* __restrict__ step, double* out, int iters) {
  __shared__ __align__(16) double s[8192];
  const unsigned t = threadIdx.x;
  for (int i = t; i < 8192; i += 128) s[i] = i * 0.5;
  __syncthreads();
  const int b0 = base[t], st = step[t];
  double a0 = 0, a1 = 0;
#pragma unroll 1
  for (int it = 0; it < iters; ++it) {
    asm volatile("" ::: "memory");   // (the loads stay in the loop)
    const int sh = (it * 3) & 7;
#pragma unroll
    for (int j = 0; j < 10; ++j) {
      const int idx = b0 + (j % 5) * st + (j / 5) * 1024 + sh * 2;
      if (W == 16) { const double2 v = reinterpret_cast<const double2*>(s)[idx]; a0 += v.x; a1 += v.y; }
      else a0 += s[idx];
    }
  }
  out[blockIdx.x * 128 + t] = a0 + a1;
}
int main() {
  const int NP = 7;
  std::vector<int> hb(NP * 128), hs(NP * 128);
  for (int t = 0; t < 128; ++t) {
    const int tv = t < 125 ? t : t - 125, e = tv / 25, q = tv % 25, a = q % 5, b = q / 5;
    hb[0 * 128 + t] = t;               hs[0 * 128 + t] = 128;
    hb[1 * 128 + t] = e * 25 + 5 * b;  hs[1 * 128 + t] = 1;
    hb[2 * 128 + t] = e * 25 + a;      hs[2 * 128 + t] = 5;
    hb[3 * 128 + t] = e * 20 + b;      hs[3 * 128 + t] = 5;   // (face 2d+t of the row line: node b of the face; next face +5)
    hb[4 * 128 + t] = t;               hs[4 * 128 + t] = 128;
    hb[5 * 128 + t] = e * 25 + 5 * b;  hs[5 * 128 + t] = 1;
    hb[6 * 128 + t] = e * 17;          hs[6 * 128 + t] = 1;
  }
  int *db, *ds; double* out;
  hipMalloc(&db, hb.size() * 4); hipMalloc(&ds, hs.size() * 4); hipMalloc(&out, 128 * 8192 * sizeof(double));
  hipMemcpy(db, hb.data(), hb.size() * 4, hipMemcpyHostToDevice); hipMemcpy(ds, hs.data(), hs.size() * 4, hipMemcpyHostToDevice);
  const int iters = 2000, nb = 256 * 16;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const char* names[NP] = {"b128 contiguous", "b128 row walk", "b128 column walk", "b128 lift gather", "b64 contiguous", "b64 row walk", "b64 element broadcast"};
  for (int m = 0; m < NP; ++m)
    for (int rep = 0; rep < 2; ++rep) {
      hipEventRecord(e0);
      if (m < 4) hipLaunchKernelGGL(k<16>, dim3(nb), dim3(128), 0, 0, db + m * 128, ds + m * 128, out, iters);
      else hipLaunchKernelGGL(k<8>, dim3(nb), dim3(128), 0, 0, db + m * 128, ds + m * 128, out, iters);
      hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      // wave-instructions per CU = nb * 2 waves * iters * 10 / 256 CUs; LDS cycles each = time * clock / that
      const double winst = (double)nb * 2 * iters * 10 / 256.0;
      if (rep) printf("%-24s %.3f ms   %.2f cycles of a CU's LDS per wave-instruction @2.0 GHz\n", names[m], ms, ms * 1e-3 * 2.0e9 / winst);
    }
  return 0;
}
