// esdg_devmath.hpp -- scalar device helpers shared by the gfx950 kernels (tensor quad and hex paths).
#pragma once
#include <hip/hip_runtime.h>

namespace esdg {
namespace devmath {

// Accuracy-attribution builds (tools/parity_truth.py): -DESDG_IEEE_DIV makes every quotient an
// IEEE division, -DESDG_LIBM_LOG uses the device library's log; neither is a product configuration.
__device__ __forceinline__ double rcp_refined(double x) {
#ifdef ESDG_IEEE_DIV
  return 1.0 / x;
#endif
  double r = __builtin_amdgcn_rcp(x);
  double e = __builtin_fma(-x, r, 1.0);
  r = __builtin_fma(r, e, r);
  e = __builtin_fma(-x, r, 1.0);
  return __builtin_fma(r, e, r);
}

// Natural logarithm for positive normal doubles: fdlibm's e_log.c algorithm (argument reduction to
// [sqrt(1/2), sqrt(2)), s = f/(2+f), degree-14 even polynomial; error < 1 ulp) on v_frexp_* and one refined
// v_rcp_f64: ~40 VALU instructions instead of the ~100 of the device-library log (its double-double
// path).  Five logs per face/volume lane made the library log 2/3 of the phase-0 kernel.
__device__ __forceinline__ double log_pos(double x) {
#ifdef ESDG_LIBM_LOG
  return ::log(x);
#endif
  double m = __builtin_amdgcn_frexp_mant(x);   // [0.5, 1)
  int e = __builtin_amdgcn_frexp_exp(x);
  const bool lo = m < 0.70710678118654752440;
  m = lo ? m + m : m;
  e = lo ? e - 1 : e;
  const double f = m - 1.0;
  const double s = f * rcp_refined(2.0 + f);
  const double z = s * s, w = z * z;
  const double t1 = w * (3.999999999940941908e-01 + w * (2.222219843214978396e-01 + w * 1.531383769920937332e-01));
  const double t2 = z * (6.666666666666735130e-01 +
                         w * (2.857142874366239149e-01 + w * (1.818357216161805012e-01 + w * 1.479819860511658591e-01)));
  const double R = t2 + t1;
  const double hfsq = 0.5 * f * f;
  const double dk = (double)e;
  return dk * 6.93147180369123816490e-01 - ((hfsq - (s * (hfsq + R) + dk * 1.90821492927058770002e-10)) - f);
}

// sqrt for x >= 0 where one ulp does not matter (the LF wavespeed): v_rsq_f64 + two coupled Newton steps, ~9 instructions
// instead of the ~20 of the correctly rounded library sqrt; 0 stays 0
__device__ __forceinline__ double sqrt_fast(double x) {
  const double xs = fmax(x, 1e-300);
  const double y = __builtin_amdgcn_rsq(xs);
  double g = xs * y, h = .5 * y;
  const double r = __builtin_fma(-h, g, .5);
  g = __builtin_fma(g, r, g);
  h = __builtin_fma(h, r, h);
  g = __builtin_fma(__builtin_fma(-g, g, xs), h, g);
  return x > 0.0 ? g : 0.0;
}

// What the tensor kernels rebuild from a face-trace record q[0..3] = (rho, u, v, beta) instead of reading it (round 3: the
// record is 32 B, not 64): q[4..5] = log rho, log beta (the logs of exactly the stored doubles, as phase 0 used to take
// them), q[7] = E, q[6] = the reference's interface wavespeed (euler_variables.jl:7-10 with rhoU_n, cavity :507: the
// pressure of the NORMAL kinetic energy only, sqrt(|u_n|) quirk Q1) for the face normal (nx, ny) / sJ.  gm1 = gamma - 1.
__device__ __forceinline__ void trace_rest(double* q, double nx, double ny, double isJ, double gm1);

// the value lane 0 of the wave holds (wave-uniform afterwards)
__device__ __forceinline__ double first_lane(double x) {
  const int lo = __builtin_amdgcn_readfirstlane(__double2loint(x)), hi = __builtin_amdgcn_readfirstlane(__double2hiint(x));
  return __hiloint2double(hi, lo);
}

// Wave issue priority (s_setprio, 0 = default ... 3) at the two ends of a one-shot workgroup's life: PRIO_ENTRY while a new
// workgroup computes its addresses and issues its global loads, PRIO_EXIT from the gather / projection / store stage on.  0 = no
// instruction.  Measured in round 3 (profiles/experiments/r03_prio_ab.log): entry priority 3 takes 6 % off the hex phase-0 kernel
// kh_project (default on there, ESDG_PRIO_KH_PROJECT) and ADDS 11 % to the 2D phase-0 kernel kt2_project, 1-2 % to kh_rhs, nothing
// to kt2_rhs; exit priority moves nothing anywhere -- so the generic hooks stay off (A/B: -DESDG_PRIO_ENTRY=n -DESDG_PRIO_EXIT=n).
#ifndef ESDG_PRIO_ENTRY
#define ESDG_PRIO_ENTRY 0
#endif
#ifndef ESDG_PRIO_EXIT
#define ESDG_PRIO_EXIT 0
#endif
#ifndef ESDG_PRIO_KH_PROJECT
#define ESDG_PRIO_KH_PROJECT 3
#endif
template <int P = ESDG_PRIO_ENTRY> __device__ __forceinline__ void prio_entry_begin() { if (P > 0) __builtin_amdgcn_s_setprio(P); }
template <int P = ESDG_PRIO_ENTRY> __device__ __forceinline__ void prio_entry_end() { if (P > 0) __builtin_amdgcn_s_setprio(0); }
__device__ __forceinline__ void prio_exit() { if (ESDG_PRIO_EXIT > 0) __builtin_amdgcn_s_setprio(ESDG_PRIO_EXIT); }

__device__ __forceinline__ void lds_add(double* p, double v) {
  __hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}

// (the part without the logarithms: q[6], q[7]; kt3_rhs takes the logs only where some flux of the wave reads them)
__device__ __forceinline__ void trace_rest_nolog(double* q, double nx, double ny, double isJ, double gm1);
__device__ __forceinline__ void trace_rest(double* q, double nx, double ny, double isJ, double gm1) {
  q[4] = log_pos(q[0]);
  q[5] = log_pos(q[3]);
  trace_rest_nolog(q, nx, ny, isJ, gm1);
}
__device__ __forceinline__ void trace_rest_nolog(double* q, double nx, double ny, double isJ, double gm1) {
  const double R = rcp_refined(q[0] * q[3]);
  const double ib = R * q[0], ir = R * q[3];                                 // 1/beta, 1/rho
  const double E = __builtin_fma(.5 * q[0], __builtin_fma(q[1], q[1], q[2] * q[2]), q[0] * ib * (.5 / gm1));
  const double un = __builtin_fma(q[2], ny, q[1] * nx) * isJ;
  const double pn = gm1 * __builtin_fma(-.5 * q[0], un * un, E);
  q[6] = fabs(sqrt_fast(fabs(un)) + sqrt_fast(1.4 * pn * ir));
  q[7] = E;
}

}  // namespace devmath
}  // namespace esdg
