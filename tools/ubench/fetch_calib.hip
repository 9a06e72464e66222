// fetch_calib.hip -- kernels with KNOWN byte counts for calibrating rocprofv3's FETCH_SIZE / WRITE_SIZE on gfx950 in the access
// shapes of the ESDG kernels, and for the practical HBM rates those shapes reach.
//   hipcc --offload-arch=gfx950 -O3 -o fetch_calib fetch_calib.hip && ./fetch_calib
// Every kernel prints: bytes read / written by construction, time per launch (HIP events, 20 launches), GB/s.
// Under `rocprofv3 --pmc FETCH_SIZE` (and, separately, WRITE_SIZE) the per-kernel counter divided by the printed byte count is the
// calibration factor of that shape (tools/fetch_calibration.sh).  Buffers are 1 GiB (>> the 256 MiB Infinity Cache).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

// 8 B per lane, coalesced (the state loads Q[f][e][node] of every kernel)
__global__ void k_stream8(const double* __restrict__ a, double* __restrict__ out, size_t n) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  double s = 0;
  for (; i < n; i += (size_t)gridDim.x * blockDim.x) s += a[i];
  if (s == 1.2345e-300) out[0] = s;
}
// 16 B per lane, coalesced
__global__ void k_stream16(const double2* __restrict__ a, double* __restrict__ out, size_t n) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  double s = 0;
  for (; i < n; i += (size_t)gridDim.x * blockDim.x) { const double2 v = a[i]; s += v.x + v.y; }
  if (s == 1.2345e-300) out[0] = s;
}
// float4 copy (the guide's 6.29 TB/s shape)
__global__ void k_copy16(const float4* __restrict__ a, float4* __restrict__ b, size_t n) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  for (; i < n; i += (size_t)gridDim.x * blockDim.x) b[i] = a[i];
}
// 8-B copy
__global__ void k_copy8(const double* __restrict__ a, double* __restrict__ b, size_t n) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  for (; i < n; i += (size_t)gridDim.x * blockDim.x) b[i] = a[i];
}
// Trace gather of the last phase: 32-B records, a face = 5 consecutive records (160 B), an element = 4 faces (640 B).  Lane t of a
// 128-thread group reads record t of its group's 5 elements' NEIGHBOURS: element e, face f -> element nbr(e, f), face (f + 2) % 4
// on a Kx x Ky periodic grid (x fastest) -- the address stream of kt2_rhs / kt2_sigma / kt3_rhs for the neighbour traces.
// own = 1: the own records instead (contiguous 3200 B per group).
__global__ void k_trace(const double2* __restrict__ A, double* __restrict__ out, int Kx, int Ky, int own) {
  const int t = threadIdx.x;
  const long g = blockIdx.x;
  double s = 0;
  if (t < 100) {
    const long e = g * 5 + t / 20;
    const int f = (t % 20) / 5, k = t % 5;
    long src = e * 20 + f * 5 + k;
    if (!own) {
      const long ex = e % Kx, ey = e / Kx;
      long nx = ex, ny = ey;
      if (f == 0) ny = (ey + Ky - 1) % Ky; else if (f == 1) nx = (ex + 1) % Kx; else if (f == 2) ny = (ey + 1) % Ky; else nx = (ex + Kx - 1) % Kx;
      src = (ny * Kx + nx) * 20 + ((f + 2) % 4) * 5 + (4 - k);
    }
    const double2 a = A[2 * src], b = A[2 * src + 1];
    s = a.x + a.y + b.x + b.y;
  }
  if (s == 1.2345e-300) out[0] = s;
}
// stores: 24-B records written as three 8-B stores per lane (kt2_sigma's B), 32-B records as two 16-B stores (kt2_project's A_U)
__global__ void k_store24(double* __restrict__ B, size_t nrec) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < nrec) { B[3 * i] = 1.0; B[3 * i + 1] = 2.0; B[3 * i + 2] = 3.0; }
}
__global__ void k_store32(double2* __restrict__ A, size_t nrec) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < nrec) { A[2 * i] = make_double2(1.0, 2.0); A[2 * i + 1] = make_double2(3.0, 4.0); }
}
__global__ void k_store8(double* __restrict__ B, size_t n) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) B[i] = 1.0;
}

template <class F> static void timeit(const char* name, double rbytes, double wbytes, F launch) {
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int i = 0; i < 3; ++i) launch();
  CK(hipEventRecord(e0));
  const int n = 20;
  for (int i = 0; i < n; ++i) launch();
  CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= n;
  printf("%-22s read %10.1f MB  write %10.1f MB  %8.4f ms  %7.2f TB/s\n", name, rbytes / 1e6, wbytes / 1e6, ms, (rbytes + wbytes) / ms / 1e9);
}

int main() {
  const size_t BYTES = (size_t)1 << 30;
  void *a, *b; double* out;
  CK(hipMalloc(&a, BYTES)); CK(hipMalloc(&b, BYTES)); CK(hipMalloc(&out, 64));
  CK(hipMemset(a, 0, BYTES)); CK(hipMemset(b, 0, BYTES));
  const int T = 256, G = 256 * 32;
  timeit("stream8", BYTES, 0, [&] { k_stream8<<<G, T>>>((const double*)a, out, BYTES / 8); });
  timeit("stream16", BYTES, 0, [&] { k_stream16<<<G, T>>>((const double2*)a, out, BYTES / 16); });
  timeit("copy16", BYTES, BYTES, [&] { k_copy16<<<G, T>>>((const float4*)a, (float4*)b, BYTES / 16); });
  timeit("copy8", BYTES, BYTES, [&] { k_copy8<<<G, T>>>((const double*)a, (double*)b, BYTES / 8); });
  // trace buffer of a 1024 x 1536 element mesh: 1572864 elements x 640 B = 1.0 GB
  const int Kx = 1024, Ky = 1536; const long K = (long)Kx * Ky;
  timeit("trace_nbr32", (double)K * 640, 0, [&] { k_trace<<<K / 5, 128>>>((const double2*)a, out, Kx, Ky, 0); });
  timeit("trace_own32", (double)K * 640, 0, [&] { k_trace<<<K / 5, 128>>>((const double2*)a, out, Kx, Ky, 1); });
  const size_t nrec24 = BYTES / 24, nrec32 = BYTES / 32;
  timeit("store24", 0, (double)nrec24 * 24, [&] { k_store24<<<(unsigned)((nrec24 + T - 1) / T), T>>>((double*)b, nrec24); });
  timeit("store32", 0, (double)nrec32 * 32, [&] { k_store32<<<(unsigned)((nrec32 + T - 1) / T), T>>>((double2*)b, nrec32); });
  timeit("store8", 0, (double)BYTES, [&] { k_store8<<<(unsigned)((BYTES / 8 + T - 1) / T), T>>>((double*)b, BYTES / 8); });
  CK(hipDeviceSynchronize());
  return 0;
}
