// esdg_setup.cpp -- host-only part of libesdg_hip.so: the one-time DG set-up a reference driver performs before
// its time loop, for hosts that have no SetupDG of their own (C, C++, Fortran).  No device code, no HIP calls.
//
//   esdg_setup_quad  <-  init_reference_quad (src/SetupDG.jl:205-268) + init_mesh (:271-318) [+ the periodic patch
//                        build_periodic_boundary_maps, src/node_map_functions.jl:66-136, as used in
//                        examples/dg2D_euler_quad.jl:38-44] + the driver-level operator assembly
//                        (dg2D_euler_quad.jl:47-91 or CompressibleNS/dg2D_CNS_cavity_optimized.jl:62-90)
//   esdg_setup_uniform_quad_mesh  <-  uniform_quad_mesh (src/UniformQuadMesh.jl:25-50)
//
// Same conventions as the Julia code: column-major (nodes x K) arrays, 1-based int64 maps, nodal LGL basis with r
// fastest, tensor Gauss quadrature with s fastest, faces s=-1, r=+1, s=+1 (reversed), r=-1 (reversed).  Connectivity is
// O(K log K) (sorted face keys) and boundary faces are paired by sorted centroids, as in esdg_cns_amd/setup_dg.py,
// whose results this file must reproduce (maps bit for bit, operators to round-off: tests/test_setup_c.py).
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <map>
#include <string>
#include <vector>

#include "../../include/esdg_hip.h"

namespace {

struct Mat {   // column-major, like Julia
  int64_t r = 0, c = 0;
  std::vector<double> a;
  Mat() {}
  Mat(int64_t r_, int64_t c_) : r(r_), c(c_), a((size_t)(r_ * c_), 0.0) {}
  double& operator()(int64_t i, int64_t j) { return a[(size_t)(j * r + i)]; }
  double operator()(int64_t i, int64_t j) const { return a[(size_t)(j * r + i)]; }
};

Mat mul(const Mat& A, const Mat& B) {
  Mat C(A.r, B.c);
  for (int64_t j = 0; j < B.c; ++j)
    for (int64_t k = 0; k < A.c; ++k) {
      const double b = B(k, j);
      if (b == 0.0) continue;
      for (int64_t i = 0; i < A.r; ++i) C(i, j) += A(i, k) * b;
    }
  return C;
}
Mat tr(const Mat& A) {
  Mat T(A.c, A.r);
  for (int64_t i = 0; i < A.r; ++i)
    for (int64_t j = 0; j < A.c; ++j) T(j, i) = A(i, j);
  return T;
}
// X = A^-1 B by LU with partial pivoting (A small and well conditioned: mass matrices)
Mat solve(Mat A, Mat B) {
  const int64_t n = A.r;
  for (int64_t k = 0; k < n; ++k) {
    int64_t p = k;
    for (int64_t i = k + 1; i < n; ++i)
      if (std::fabs(A(i, k)) > std::fabs(A(p, k))) p = i;
    if (p != k) {
      for (int64_t j = 0; j < n; ++j) std::swap(A(k, j), A(p, j));
      for (int64_t j = 0; j < B.c; ++j) std::swap(B(k, j), B(p, j));
    }
    for (int64_t i = k + 1; i < n; ++i) {
      const double f = A(i, k) / A(k, k);
      if (f == 0.0) continue;
      for (int64_t j = k; j < n; ++j) A(i, j) -= f * A(k, j);
      for (int64_t j = 0; j < B.c; ++j) B(i, j) -= f * B(k, j);
    }
  }
  for (int64_t j = 0; j < B.c; ++j)
    for (int64_t i = n - 1; i >= 0; --i) {
      double s = B(i, j);
      for (int64_t k = i + 1; k < n; ++k) s -= A(i, k) * B(k, j);
      B(i, j) = s / A(i, i);
    }
  return B;
}
void droptol(Mat& A, double tol) {
  for (double& x : A.a)
    if (std::fabs(x) <= tol) x = 0.0;
}

// Legendre P_n and derivatives by recurrence
void legendre(int n, double x, double& P, double& dP, double& d2P) {
  double p0 = 1.0, p1 = x;
  if (n == 0) { P = 1; dP = 0; d2P = 0; return; }
  for (int k = 2; k <= n; ++k) {
    const double p2 = ((2 * k - 1) * x * p1 - (k - 1) * p0) / k;
    p0 = p1; p1 = p2;
  }
  P = p1;
  dP = n * (x * p1 - p0) / (x * x - 1.0);
  d2P = (2 * x * dP - n * (n + 1.0) * p1) / (1.0 - x * x);
}

// Gauss-Legendre rule with n points (gauss_quad(0,0,n-1), src/Basis1D.jl:59-77), exactly symmetric
void gauss_legendre(int n, std::vector<double>& x, std::vector<double>& w) {
  x.assign(n, 0.0); w.assign(n, 0.0);
  for (int i = 0; i < n; ++i) {
    double xi = -std::cos(M_PI * (i + 0.75) / (n + 0.5));
    for (int it = 0; it < 100; ++it) {
      double P, dP, d2P;
      legendre(n, xi, P, dP, d2P);
      const double dx = P / dP;
      xi -= dx;
      if (std::fabs(dx) < 1e-16) break;
    }
    double P, dP, d2P;
    legendre(n, xi, P, dP, d2P);
    x[i] = xi;
    w[i] = 2.0 / ((1.0 - xi * xi) * dP * dP);
  }
  for (int i = 0; i < n / 2; ++i) {
    const double xs = .5 * (x[n - 1 - i] - x[i]), ws = .5 * (w[i] + w[n - 1 - i]);
    x[i] = -xs; x[n - 1 - i] = xs; w[i] = ws; w[n - 1 - i] = ws;
  }
  if (n % 2) x[n / 2] = 0.0;
}

// Legendre-Gauss-Lobatto nodes with N+1 points (gauss_lobatto_quad(0,0,N), src/Basis1D.jl:24-47)
std::vector<double> gauss_lobatto(int N) {
  std::vector<double> x(N + 1, 0.0);
  if (N == 0) return x;
  x[0] = -1.0; x[N] = 1.0;
  for (int i = 1; i < N; ++i) {
    double xi = -std::cos(M_PI * i / N);
    for (int it = 0; it < 100; ++it) {
      double P, dP, d2P;
      legendre(N, xi, P, dP, d2P);
      const double dx = dP / d2P;
      xi -= dx;
      if (std::fabs(dx) < 1e-16) break;
    }
    x[i] = xi;
  }
  for (int i = 0; i <= N / 2; ++i) {
    const double xs = .5 * (x[N - i] - x[i]);
    x[i] = -xs; x[N - i] = xs;
  }
  return x;
}

// L(i,k) = l_k(xs_i) of the Lagrange basis on `nodes`
Mat lagrange_interp(const std::vector<double>& nodes, const std::vector<double>& xs) {
  const int n = (int)nodes.size();
  Mat L((int64_t)xs.size(), n);
  for (size_t i = 0; i < xs.size(); ++i)
    for (int k = 0; k < n; ++k) {
      double v = 1.0;
      for (int m = 0; m < n; ++m)
        if (m != k) v *= (xs[i] - nodes[m]) / (nodes[k] - nodes[m]);
      L((int64_t)i, k) = v;
    }
  return L;
}
// D(i,k) = l_k'(nodes_i)
Mat lagrange_diff(const std::vector<double>& nodes) {
  const int n = (int)nodes.size();
  std::vector<double> bw(n, 1.0);
  for (int k = 0; k < n; ++k)
    for (int m = 0; m < n; ++m)
      if (m != k) bw[k] /= (nodes[k] - nodes[m]);
  Mat D(n, n);
  for (int i = 0; i < n; ++i) {
    double s = 0.0;
    for (int k = 0; k < n; ++k)
      if (k != i) {
        D(i, k) = (bw[k] / bw[i]) / (nodes[i] - nodes[k]);
        s += D(i, k);
      }
    D(i, i) = -s;
  }
  return D;
}

}  // namespace

struct esdg_setup {
  int N = 0, formulation = 0;
  int64_t K = 0, Kglobal = 0, e0 = 0;
  std::map<std::string, Mat> arr;
  std::map<std::string, std::vector<int64_t>> maps;
  std::vector<uint8_t> bkind;   // per mapB entry: 1 on the y = ymax side (the lid of init_BC_funs, cavity_optimized.jl:139)
};

namespace {
thread_local std::string g_setup_err;
int sfail(const char* msg) { g_setup_err = msg; return ESDG_ERR_ARG; }

void reference_quad(int N, std::map<std::string, Mat>& A) {
  const int n1 = N + 1, nq1 = N + 1, Np = n1 * n1, Nq = nq1 * nq1, Nfq = 4 * nq1;
  std::vector<double> x1 = gauss_lobatto(N), r1D, w1D;
  gauss_legendre(nq1, r1D, w1D);
  Mat D1 = lagrange_diff(x1), Iq = lagrange_interp(x1, r1D);
  Mat r(Np, 1), s(Np, 1), Dr(Np, Np), Ds(Np, Np), V1(Np, 4);
  const double rv[4] = {-1, 1, -1, 1}, sv[4] = {-1, -1, 1, 1};
  for (int j = 0; j < n1; ++j)
    for (int i = 0; i < n1; ++i) {
      const int k = i + j * n1;
      r(k, 0) = x1[i]; s(k, 0) = x1[j];
      for (int m = 0; m < n1; ++m) {
        Dr(k, m + j * n1) = D1(i, m);
        Ds(k, i + m * n1) = D1(j, m);
      }
      for (int v = 0; v < 4; ++v) V1(k, v) = 0.25 * (1 + x1[i] * rv[v]) * (1 + x1[j] * sv[v]);
    }
  Mat rf(Nfq, 1), sf(Nfq, 1), wf(Nfq, 1), nrJ(Nfq, 1), nsJ(Nfq, 1);
  for (int i = 0; i < nq1; ++i) {
    rf(i, 0) = r1D[i]; sf(i, 0) = -1; nsJ(i, 0) = -1;
    rf(nq1 + i, 0) = 1; sf(nq1 + i, 0) = r1D[i]; nrJ(nq1 + i, 0) = 1;
    rf(2 * nq1 + i, 0) = -r1D[i]; sf(2 * nq1 + i, 0) = 1; nsJ(2 * nq1 + i, 0) = 1;
    rf(3 * nq1 + i, 0) = -1; sf(3 * nq1 + i, 0) = -r1D[i]; nrJ(3 * nq1 + i, 0) = -1;
    for (int f = 0; f < 4; ++f) wf(f * nq1 + i, 0) = w1D[i];
  }
  Mat rq(Nq, 1), sq(Nq, 1), wq(Nq, 1), Vq(Nq, Np);
  for (int b = 0; b < nq1; ++b)
    for (int a = 0; a < nq1; ++a) {
      const int q = a + b * nq1;
      rq(q, 0) = r1D[b]; sq(q, 0) = r1D[a]; wq(q, 0) = w1D[a] * w1D[b];
      for (int j = 0; j < n1; ++j)
        for (int i = 0; i < n1; ++i) Vq(q, i + j * n1) = Iq(b, i) * Iq(a, j);
    }
  Mat WVq(Nq, Np);
  for (int q = 0; q < Nq; ++q)
    for (int k = 0; k < Np; ++k) WVq(q, k) = wq(q, 0) * Vq(q, k);
  Mat M = mul(tr(Vq), WVq);
  Mat Pq = solve(M, tr(WVq));
  std::vector<double> rfv(rf.a), sfv(sf.a);
  Mat Lr = lagrange_interp(x1, rfv), Ls = lagrange_interp(x1, sfv), Vf(Nfq, Np);
  for (int f = 0; f < Nfq; ++f)
    for (int j = 0; j < n1; ++j)
      for (int i = 0; i < n1; ++i) Vf(f, i + j * n1) = Lr(f, i) * Ls(f, j);
  Mat VfW = tr(Vf);
  for (int k = 0; k < Np; ++k)
    for (int f = 0; f < Nfq; ++f) VfW(k, f) *= wf(f, 0);
  Mat LIFT = solve(M, VfW);
  droptol(Dr, 1e-10); droptol(Ds, 1e-10); droptol(Vf, 1e-10); droptol(LIFT, 1e-10);
  A["r"] = r; A["s"] = s; A["V1"] = V1; A["Dr"] = Dr; A["Ds"] = Ds; A["rf"] = rf; A["sf"] = sf; A["wf"] = wf;
  A["nrJ"] = nrJ; A["nsJ"] = nsJ; A["rq"] = rq; A["sq"] = sq; A["wq"] = wq; A["Vq"] = Vq; A["M"] = M; A["Pq"] = Pq;
  A["Vf"] = Vf; A["LIFT"] = LIFT;
}

// dg2D_euler_quad.jl:47-78 (formulation 0) / dg2D_CNS_cavity_optimized.jl:62-90 (1, 2)
void driver_ops(int formulation, std::map<std::string, Mat>& A) {
  const Mat &M = A["M"], &Pq = A["Pq"], &Vf = A["Vf"], &Vq = A["Vq"], &wf = A["wf"], &wq = A["wq"];
  const int64_t Nq = Vq.r, Np = Vq.c, Nfq = Vf.r, Nh = Nq + Nfq;
  Mat Ef = mul(Vf, Pq);
  Mat PtM = mul(tr(Pq), M);
  const char* dn[2] = {"Dr", "Ds"};
  const char* nn[2] = {"nrJ", "nsJ"};
  const char* on[2] = {"Qrhskew", "Qshskew"};
  for (int d = 0; d < 2; ++d) {
    Mat Q = mul(mul(PtM, A[dn[d]]), Pq);
    const Mat& nJ = A[nn[d]];
    Mat Qh(Nh, Nh);
    for (int64_t i = 0; i < Nq; ++i)
      for (int64_t j = 0; j < Nq; ++j) Qh(i, j) = .5 * (Q(i, j) - Q(j, i));
    for (int64_t i = 0; i < Nq; ++i)
      for (int64_t f = 0; f < Nfq; ++f) {
        const double b = wf(f, 0) * nJ(f, 0);
        Qh(i, Nq + f) = .5 * Ef(f, i) * b;
        Qh(Nq + f, i) = -.5 * b * Ef(f, i);
      }
    for (int64_t f = 0; f < Nfq; ++f) Qh(Nq + f, Nq + f) = .5 * wf(f, 0) * nJ(f, 0);
    Mat S(Nh, Nh);
    for (int64_t i = 0; i < Nh; ++i)
      for (int64_t j = 0; j < Nh; ++j) S(i, j) = .5 * (Qh(i, j) - Qh(j, i));
    A[on[d]] = S;
  }
  A["Ef"] = Ef;
  if (formulation == 0) {
    Mat Vh(Nh, Nq);
    for (int64_t q = 0; q < Nq; ++q) Vh(q, q) = 1.0;
    for (int64_t f = 0; f < Nfq; ++f)
      for (int64_t q = 0; q < Nq; ++q) Vh(Nq + f, q) = Ef(f, q);
    droptol(Vh, 1e-12);
    Mat Ph(Nq, Nh), Lf(Nq, Nfq);
    for (int64_t q = 0; q < Nq; ++q) {
      for (int64_t j = 0; j < Nh; ++j) Ph(q, j) = Vh(j, q) / wq(q, 0);
      for (int64_t f = 0; f < Nfq; ++f) Lf(q, f) = Ef(f, q) * wf(f, 0) / wq(q, 0);
    }
    droptol(Ph, 1e-12); droptol(Lf, 1e-12);
    A["Vh"] = Vh; A["Ph"] = Ph; A["Lf"] = Lf;
  } else {
    Mat Vh(Nh, Np);
    for (int64_t k = 0; k < Np; ++k) {
      for (int64_t q = 0; q < Nq; ++q) Vh(q, k) = Vq(q, k);
      for (int64_t f = 0; f < Nfq; ++f) Vh(Nq + f, k) = Vf(f, k);
    }
    A["Vh"] = Vh;
    A["Ph"] = solve(M, tr(Vh));
    A["VhP"] = mul(Vh, Pq);
  }
}

struct Face { int64_t a, b, id; };

}  // namespace

extern "C" {

const char* esdg_setup_last_error(void) { return g_setup_err.c_str(); }

int esdg_setup_uniform_quad_mesh(int Kx, int Ky, double* VX, double* VY, int64_t* EToV) {
  if (Kx < 1 || Ky < 1 || !VX || !VY || !EToV) return sfail("bad uniform mesh arguments");
  const int64_t Nxp = Kx + 1, Nyp = Ky + 1, K = (int64_t)Kx * Ky;
  for (int64_t i = 0; i < Nxp; ++i)
    for (int64_t j = 0; j < Nyp; ++j) {
      // same values as numpy.linspace(-1, 1, n): start + i*step, last point exact
      VX[i * Nyp + j] = i == Nxp - 1 ? 1.0 : -1.0 + i * (2.0 / Kx);
      VY[i * Nyp + j] = j == Nyp - 1 ? 1.0 : -1.0 + j * (2.0 / Ky);
    }
  for (int64_t e = 0; e < K; ++e) {
    const int64_t jx = e % Kx, iy = e / Kx, v0 = jx * Nyp + iy + 1;
    EToV[e] = v0; EToV[K + e] = v0 + Nyp; EToV[2 * K + e] = v0 + 1; EToV[3 * K + e] = v0 + Nyp + 1;
  }
  return ESDG_OK;
}

int esdg_setup_quad(int N, int formulation, const double* VX, const double* VY, int64_t Nv, const int64_t* EToV, int64_t Kg,
                    int periodic, int64_t e_begin, int64_t e_end, esdg_setup** out) {
  if (!out) return sfail("null output");
  *out = nullptr;
  if (N < 1 || N > 9 || formulation < 0 || formulation > 2 || !VX || !VY || !EToV || Kg < 1 || Nv < 4) return sfail("bad set-up arguments");
  if (e_end <= 0) { e_begin = 0; e_end = Kg; }
  if (e_begin < 0 || e_end > Kg || e_begin >= e_end) return sfail("bad element range");
  esdg_setup* S = new esdg_setup();
  S->N = N; S->formulation = formulation; S->Kglobal = Kg; S->e0 = e_begin; S->K = e_end - e_begin;
  auto& A = S->arr;
  reference_quad(N, A);
  driver_ops(formulation, A);
  const int64_t K = S->K, e0 = e_begin;
  const Mat &V1 = A["V1"], &Vf = A["Vf"], &Vq = A["Vq"], &Dr = A["Dr"], &Ds = A["Ds"];
  const int Np = (int)V1.r, Nfq = (int)Vf.r, Nfp = Nfq / 4, Nq = (int)Vq.r;
  auto ev = [&](int64_t e, int v) { return EToV[(size_t)v * Kg + e] - 1; };
  for (int64_t e = 0; e < Kg; ++e)
    for (int v = 0; v < 4; ++v)
      if (ev(e, v) < 0 || ev(e, v) >= Nv) { delete S; return sfail("EToV entry out of range"); }

  // ---- connect_mesh (src/connect_mesh.jl:17-36): faces with equal sorted vertex pairs are neighbours ----------
  const int fv[4][2] = {{0, 1}, {1, 3}, {2, 3}, {0, 2}};   // src/UniformQuadMesh.jl:67-69 (0-based)
  std::vector<Face> faces((size_t)Kg * 4);
  for (int64_t e = 0; e < Kg; ++e)
    for (int f = 0; f < 4; ++f) {
      int64_t a = ev(e, fv[f][0]), b = ev(e, fv[f][1]);
      if (a > b) std::swap(a, b);
      faces[(size_t)e * 4 + f] = {a, b, e * 4 + f};
    }
  std::vector<Face> sorted(faces);
  std::stable_sort(sorted.begin(), sorted.end(), [](const Face& x, const Face& y) { return x.a != y.a ? x.a < y.a : x.b < y.b; });
  std::vector<int64_t> FToF((size_t)Kg * 4);
  for (size_t i = 0; i < FToF.size(); ++i) FToF[i] = (int64_t)i;
  for (size_t i = 0; i + 1 < sorted.size(); ++i)
    if (sorted[i].a == sorted[i + 1].a && sorted[i].b == sorted[i + 1].b) {
      FToF[(size_t)sorted[i].id] = sorted[i + 1].id;
      FToF[(size_t)sorted[i + 1].id] = sorted[i].id;
    }

  // ---- coordinates of the local elements -------------------------------------------------------------------------
  Mat x(Np, K), y(Np, K);
  for (int64_t e = 0; e < K; ++e)
    for (int k = 0; k < Np; ++k) {
      double sx = 0, sy = 0;
      for (int v = 0; v < 4; ++v) { sx += V1(k, v) * VX[ev(e0 + e, v)]; sy += V1(k, v) * VY[ev(e0 + e, v)]; }
      x(k, e) = sx; y(k, e) = sy;
    }
  Mat xf = mul(Vf, x), yf = mul(Vf, y), xq = mul(Vq, x), yq = mul(Vq, y);
  // face-node coordinates of an arbitrary global face, straight from the vertices
  Mat VfV1 = mul(Vf, V1);
  auto face_xy = [&](int64_t gf, int i, double& px, double& py) {
    const int64_t e = gf / 4; const int f = (int)(gf % 4);
    px = 0; py = 0;
    for (int v = 0; v < 4; ++v) { px += VfV1(f * Nfp + i, v) * VX[ev(e, v)]; py += VfV1(f * Nfp + i, v) * VY[ev(e, v)]; }
  };

  // ---- build_node_maps (src/node_map_functions.jl:23-55) --------------------------------------------------------------
  std::vector<int64_t> mapM((size_t)K * Nfq), mapP((size_t)K * Nfq);
  std::vector<double> X1(2 * Nfp), X2(2 * Nfp);
  for (int64_t lf = 0; lf < K * 4; ++lf) {
    const int64_t f1 = e0 * 4 + lf, f2 = FToF[(size_t)f1];
    for (int i = 0; i < Nfp; ++i) { face_xy(f1, i, X1[i], X1[Nfp + i]); face_xy(f2, i, X2[i], X2[Nfp + i]); }
    double refd = 0;
    for (int i = 0; i < Nfp; ++i)
      for (int j = 0; j < Nfp; ++j) refd = std::max(refd, std::fabs(X1[i] - X2[j]) + std::fabs(X1[Nfp + i] - X2[Nfp + j]));
    for (int i = 0; i < Nfp; ++i) {
      const int64_t m = (e0 * 4 + lf) * Nfp + i + 1;
      mapM[(size_t)(lf * Nfp + i)] = m;
      int64_t p = m;
      if (f2 != f1)
        for (int j = 0; j < Nfp; ++j)
          if (std::fabs(X1[i] - X2[j]) + std::fabs(X1[Nfp + i] - X2[Nfp + j]) < 1e-10 * refd) { p = f2 * Nfp + j + 1; break; }
      mapP[(size_t)(lf * Nfp + i)] = p;
    }
  }
  std::vector<int64_t> mapB;
  for (size_t n = 0; n < mapM.size(); ++n)
    if (mapM[n] == mapP[n]) mapB.push_back(mapM[n]);

  // ---- periodic patch (dg2D_euler_quad.jl:38-44; build_periodic_boundary_maps :66-136) -------------------------------
  if (periodic) {
    double xmin = VX[0], xmax = VX[0], ymin = VY[0], ymax = VY[0];
    for (int64_t v = 0; v < Nv; ++v) { xmin = std::min(xmin, VX[v]); xmax = std::max(xmax, VX[v]); ymin = std::min(ymin, VY[v]); ymax = std::max(ymax, VY[v]); }
    const double LX = xmax - xmin, LY = ymax - ymin, TOL = 1e-12;
    struct BF { int64_t gf; double xc, yc; };
    std::vector<BF> bf;
    for (int64_t gf = 0; gf < Kg * 4; ++gf)
      if (FToF[(size_t)gf] == gf) {
        double sx = 0, sy = 0, px, py;
        for (int i = 0; i < Nfp; ++i) { face_xy(gf, i, px, py); sx += px; sy += py; }
        bf.push_back({gf, sx / Nfp, sy / Nfp});
      }
    std::map<int64_t, int64_t> partner;   // global face -> global partner face
    for (int dir = 0; dir < 2; ++dir) {
      std::vector<BF> lo, hi;
      for (const BF& b : bf) {
        const double c = dir == 0 ? b.yc : b.xc, cmin = dir == 0 ? ymin : xmin, cmax = dir == 0 ? ymax : xmax, L = dir == 0 ? LY : LX;
        if (std::fabs(c - cmin) < TOL * L) lo.push_back(b);
        else if (std::fabs(c - cmax) < TOL * L) hi.push_back(b);
      }
      auto key = [dir](const BF& p, const BF& q) { return dir == 0 ? p.xc < q.xc : p.yc < q.yc; };
      std::stable_sort(lo.begin(), lo.end(), key);
      std::stable_sort(hi.begin(), hi.end(), key);
      if (lo.size() != hi.size()) { delete S; return sfail("periodic boundary faces do not pair up"); }
      for (size_t i = 0; i < lo.size(); ++i) { partner[lo[i].gf] = hi[i].gf; partner[hi[i].gf] = lo[i].gf; }
    }
    for (size_t n = 0; n < mapM.size(); ++n) {
      if (mapM[n] != mapP[n]) continue;
      const int64_t gnode = mapM[n] - 1, gf = gnode / Nfp;
      const int i = (int)(gnode % Nfp);
      auto it = partner.find(gf);
      if (it == partner.end()) { delete S; return sfail("boundary face off the box"); }
      const int64_t pf = it->second;
      double ax, ay, bx, by, fx0, fy0, pcx = 0, pcy = 0, mcx = 0, mcy = 0;
      for (int j = 0; j < Nfp; ++j) { face_xy(gf, j, ax, ay); mcx += ax; mcy += ay; face_xy(pf, j, bx, by); pcx += bx; pcy += by; }
      const bool yface = std::fabs(std::fabs(mcy - pcy) / Nfp - LY) < 1e-9 * LY && std::fabs(mcx - pcx) / Nfp < 1e-9 * LX;
      face_xy(gf, i, fx0, fy0);
      int64_t p = -1;
      for (int j = 0; j < Nfp; ++j) {
        face_xy(pf, j, bx, by);
        const double d = yface ? std::fabs(fx0 - bx) : std::fabs(fy0 - by);
        if (d < 1e-9 * (yface ? LX : LY)) { p = pf * Nfp + j + 1; break; }
      }
      if (p < 0) { delete S; return sfail("periodic node matching failed"); }
      mapP[n] = p;
    }
  }

  // ---- geometric factors (src/geometric_factors.jl:16-27), wJq, normals (SetupDG.jl:303-316) -------------------------
  Mat xr = mul(Dr, x), xs = mul(Ds, x), yr = mul(Dr, y), ys = mul(Ds, y);
  Mat rxJ(Np, K), sxJ(Np, K), ryJ(Np, K), syJ(Np, K), J(Np, K);
  for (size_t n = 0; n < J.a.size(); ++n) {
    J.a[n] = -xs.a[n] * yr.a[n] + xr.a[n] * ys.a[n];
    rxJ.a[n] = ys.a[n]; sxJ.a[n] = -yr.a[n]; ryJ.a[n] = -xs.a[n]; syJ.a[n] = xr.a[n];
  }
  Mat wJq = mul(Vq, J);
  for (int64_t e = 0; e < K; ++e)
    for (int q = 0; q < Nq; ++q) wJq(q, e) *= A["wq"](q, 0);
  Mat frx = mul(Vf, rxJ), fsx = mul(Vf, sxJ), fry = mul(Vf, ryJ), fsy = mul(Vf, syJ);
  Mat nxJ(Nfq, K), nyJ(Nfq, K), sJ(Nfq, K);
  for (int64_t e = 0; e < K; ++e)
    for (int f = 0; f < Nfq; ++f) {
      const double nr = A["nrJ"](f, 0), ns = A["nsJ"](f, 0);
      nxJ(f, e) = frx(f, e) * nr + fsx(f, e) * ns;
      nyJ(f, e) = fry(f, e) * nr + fsy(f, e) * ns;
      sJ(f, e) = std::sqrt(nxJ(f, e) * nxJ(f, e) + nyJ(f, e) * nyJ(f, e));
    }
  // metrics interpolated to the hybrid nodes (dg2D_euler_quad.jl:86-88 / cavity_optimized.jl:85-87)
  const Mat& Vh = A["Vh"];
  if (formulation == 0) {   // Vh acts on the quadrature basis: rxJ etc. live at the Gauss nodes = nodal (Np == Nq) values
    A["rxJ"] = mul(Vh, rxJ); A["sxJ"] = mul(Vh, sxJ); A["ryJ"] = mul(Vh, ryJ); A["syJ"] = mul(Vh, syJ);
  } else {
    A["rxJ"] = mul(Vh, rxJ); A["sxJ"] = mul(Vh, sxJ); A["ryJ"] = mul(Vh, ryJ); A["syJ"] = mul(Vh, syJ);
  }
  A["x"] = x; A["y"] = y; A["xf"] = xf; A["yf"] = yf; A["xq"] = xq; A["yq"] = yq; A["J"] = J; A["wJq"] = wJq;
  A["nxJ"] = nxJ; A["nyJ"] = nyJ; A["sJ"] = sJ;
  std::vector<int64_t> FToFl((size_t)K * 4);
  for (int64_t lf = 0; lf < K * 4; ++lf) FToFl[(size_t)lf] = FToF[(size_t)(e0 * 4 + lf)] + 1;
  {
    double ymax = VY[0];
    for (int64_t v = 0; v < Nv; ++v) ymax = std::max(ymax, VY[v]);
    S->bkind.assign(mapB.size(), 0);
    for (size_t i = 0; i < mapB.size(); ++i) {
      const int64_t l = mapB[i] - 1 - e0 * Nfq;
      S->bkind[i] = std::fabs(yf.a[(size_t)l] - ymax) < 1e-12 ? 1 : 0;
    }
  }
  S->maps["FToF"] = FToFl; S->maps["mapM"] = mapM; S->maps["mapP"] = mapP; S->maps["mapB"] = mapB;
  *out = S;
  return ESDG_OK;
}

const double* esdg_setup_array(const esdg_setup* s, const char* name, int64_t* rows, int64_t* cols) {
  if (!s || !name) return nullptr;
  auto it = s->arr.find(name);
  if (it == s->arr.end()) return nullptr;
  if (rows) *rows = it->second.r;
  if (cols) *cols = it->second.c;
  return it->second.a.data();
}

const int64_t* esdg_setup_map(const esdg_setup* s, const char* name, int64_t* n) {
  if (!s || !name) return nullptr;
  auto it = s->maps.find(name);
  if (it == s->maps.end()) return nullptr;
  if (n) *n = (int64_t)it->second.size();
  return it->second.data();
}

int esdg_setup_fill(const esdg_setup* s, esdg_ops_t* ops, esdg_mesh_t* mesh) {
  if (!s || !ops || !mesh) return sfail("null argument");
  auto g = [&](const char* n) -> const double* { auto it = s->arr.find(n); return it == s->arr.end() ? nullptr : it->second.a.data(); };
  std::memset(ops, 0, sizeof *ops);
  std::memset(mesh, 0, sizeof *mesh);
  const int N1 = s->N + 1;
  ops->N = s->N; ops->Np = N1 * N1; ops->Nq = N1 * N1; ops->Nfq = 4 * N1;
  ops->Qrhskew = g("Qrhskew"); ops->Qshskew = g("Qshskew"); ops->Ph = g("Ph"); ops->wq = g("wq"); ops->wf = g("wf");
  ops->Ef = g("Ef"); ops->Lf = g("Lf"); ops->Vq = g("Vq"); ops->Pq = g("Pq"); ops->VhP = g("VhP"); ops->LIFT = g("LIFT");
  ops->Vf = g("Vf"); ops->Dr = g("Dr"); ops->Ds = g("Ds");
  mesh->K = s->K; mesh->geo_ld = ops->Nq + ops->Nfq;
  mesh->rxJ = g("rxJ"); mesh->sxJ = g("sxJ"); mesh->ryJ = g("ryJ"); mesh->syJ = g("syJ"); mesh->J = g("J"); mesh->wJq = g("wJq");
  mesh->nxJ = g("nxJ"); mesh->nyJ = g("nyJ"); mesh->sJ = g("sJ");
  mesh->mapP = s->maps.at("mapP").data();
  const auto& mb = s->maps.at("mapB");
  bool walls = false;   // boundary nodes that are still self-mapped (no periodic patch) are walls
  const auto& mp = s->maps.at("mapP");
  const auto& mm = s->maps.at("mapM");
  for (size_t n = 0; n < mp.size(); ++n) walls = walls || mp[n] == mm[n];
  mesh->mapB = walls ? mb.data() : nullptr;
  mesh->NmapB = walls ? (int64_t)mb.size() : 0;
  mesh->bkind = walls ? s->bkind.data() : nullptr;
  mesh->elem_offset = s->e0; mesh->Kglobal = s->Kglobal; mesh->nranks = 1; mesh->rank = 0; mesh->rank_offsets = nullptr;
  return ESDG_OK;
}

int esdg_setup_destroy(esdg_setup* s) {
  delete s;
  return ESDG_OK;
}

}  // extern "C"

// =====================================================================================================================
// hexahedra: init_reference_hex (src/SetupDG.jl:323-387), init_mesh 3D (:389-434), uniform_hex_mesh
// (src/UniformHexMesh.jl:25-80), 3D periodic patch (src/node_map_functions.jl:139-213) and the operator assembly of
// examples/dg3D_euler_hex.jl:34-98 -- with the INTENDED face-vertex sets, mirroring
// esdg_cns_amd/setup_dg.py:init_reference_hex / init_mesh_3d / hex_ops / hex_driver_geometry.
// =====================================================================================================================
namespace {

void reference_hex(int N, std::map<std::string, Mat>& A) {
  const int n1 = N + 1, nq1 = N + 1, Np = n1 * n1 * n1, Nq = nq1 * nq1 * nq1, nf = nq1 * nq1, Nfq = 6 * nf;
  std::vector<double> x1 = gauss_lobatto(N), r1D, w1D;
  gauss_legendre(nq1, r1D, w1D);
  Mat D1 = lagrange_diff(x1), Iq = lagrange_interp(x1, r1D);
  Mat r(Np, 1), s(Np, 1), t(Np, 1), Dr(Np, Np), Ds(Np, Np), Dt(Np, Np), V1(Np, 8);
  for (int k = 0; k < n1; ++k)
    for (int j = 0; j < n1; ++j)
      for (int i = 0; i < n1; ++i) {
        const int n = i + n1 * (j + n1 * k);
        s(n, 0) = x1[i]; r(n, 0) = x1[j]; t(n, 0) = x1[k];
        for (int m = 0; m < n1; ++m) {
          Ds(n, m + n1 * (j + n1 * k)) = D1(i, m);
          Dr(n, i + n1 * (m + n1 * k)) = D1(j, m);
          Dt(n, i + n1 * (j + n1 * m)) = D1(k, m);
        }
        for (int v = 0; v < 8; ++v) {
          const double sv = 2.0 * (v % 2) - 1, rv = 2.0 * ((v / 2) % 2) - 1, tv = 2.0 * (v / 4) - 1;
          V1(n, v) = 0.125 * (1 + x1[j] * rv) * (1 + x1[i] * sv) * (1 + x1[k] * tv);
        }
      }
  Mat rf(Nfq, 1), sf(Nfq, 1), tf(Nfq, 1), wf(Nfq, 1), nrJ(Nfq, 1), nsJ(Nfq, 1), ntJ(Nfq, 1);
  for (int m = 0; m < nf; ++m) {
    const double rq_ = r1D[m / nq1], sq_ = r1D[m % nq1], w = w1D[m / nq1] * w1D[m % nq1];
    const double R[6] = {-1, 1, rq_, rq_, rq_, rq_}, S_[6] = {rq_, rq_, -1, 1, sq_, sq_}, T[6] = {sq_, sq_, sq_, sq_, -1, 1};
    for (int f = 0; f < 6; ++f) {
      rf(f * nf + m, 0) = R[f]; sf(f * nf + m, 0) = S_[f]; tf(f * nf + m, 0) = T[f]; wf(f * nf + m, 0) = w;
    }
    nrJ(m, 0) = -1; nrJ(nf + m, 0) = 1; nsJ(2 * nf + m, 0) = -1; nsJ(3 * nf + m, 0) = 1; ntJ(4 * nf + m, 0) = -1; ntJ(5 * nf + m, 0) = 1;
  }
  Mat rq(Nq, 1), sq(Nq, 1), tq(Nq, 1), wq(Nq, 1), Vq(Nq, Np);
  for (int k = 0; k < nq1; ++k)
    for (int j = 0; j < nq1; ++j)
      for (int i = 0; i < nq1; ++i) {
        const int q = i + nq1 * (j + nq1 * k);
        sq(q, 0) = r1D[i]; rq(q, 0) = r1D[j]; tq(q, 0) = r1D[k]; wq(q, 0) = w1D[i] * w1D[j] * w1D[k];
        for (int kk = 0; kk < n1; ++kk)
          for (int jj = 0; jj < n1; ++jj)
            for (int ii = 0; ii < n1; ++ii) Vq(q, ii + n1 * (jj + n1 * kk)) = Iq(k, kk) * Iq(j, jj) * Iq(i, ii);
      }
  Mat WVq(Nq, Np);
  for (int q = 0; q < Nq; ++q)
    for (int n = 0; n < Np; ++n) WVq(q, n) = wq(q, 0) * Vq(q, n);
  Mat M = mul(tr(Vq), WVq);
  Mat Pq = solve(M, tr(WVq));
  Mat Ls = lagrange_interp(x1, sf.a), Lr = lagrange_interp(x1, rf.a), Lt = lagrange_interp(x1, tf.a), Vf(Nfq, Np);
  for (int f = 0; f < Nfq; ++f)
    for (int k = 0; k < n1; ++k)
      for (int j = 0; j < n1; ++j)
        for (int i = 0; i < n1; ++i) Vf(f, i + n1 * (j + n1 * k)) = Lt(f, k) * Lr(f, j) * Ls(f, i);
  Mat VfW = tr(Vf);
  for (int n = 0; n < Np; ++n)
    for (int f = 0; f < Nfq; ++f) VfW(n, f) *= wf(f, 0);
  Mat LIFT = solve(M, VfW);
  droptol(Dr, 1e-12); droptol(Ds, 1e-12); droptol(Dt, 1e-12); droptol(Vf, 1e-12); droptol(LIFT, 1e-12);
  A["r"] = r; A["s"] = s; A["t"] = t; A["V1"] = V1; A["Dr"] = Dr; A["Ds"] = Ds; A["Dt"] = Dt;
  A["rf"] = rf; A["sf"] = sf; A["tf"] = tf; A["wf"] = wf; A["nrJ"] = nrJ; A["nsJ"] = nsJ; A["ntJ"] = ntJ;
  A["rq"] = rq; A["sq"] = sq; A["tq"] = tq; A["wq"] = wq; A["Vq"] = Vq; A["M"] = M; A["Pq"] = Pq; A["Vf"] = Vf; A["LIFT"] = LIFT;
}

// dg3D_euler_hex.jl:34-56, 92-98
void hex_driver_ops(std::map<std::string, Mat>& A) {
  const Mat &M = A["M"], &Pq = A["Pq"], &Vf = A["Vf"], &wf = A["wf"], &wq = A["wq"];
  const int64_t Nq = A["Vq"].r, Nfq = Vf.r, Nh = Nq + Nfq;
  Mat Ef = mul(Vf, Pq), PtM = mul(tr(Pq), M);
  const char* dn[3] = {"Dr", "Ds", "Dt"};
  const char* nn[3] = {"nrJ", "nsJ", "ntJ"};
  const char* on[3] = {"Qrhskew", "Qshskew", "Qthskew"};
  for (int d = 0; d < 3; ++d) {
    Mat Q = mul(mul(PtM, A[dn[d]]), Pq);
    const Mat& nJ = A[nn[d]];
    Mat Qh(Nh, Nh);
    for (int64_t i = 0; i < Nq; ++i)
      for (int64_t j = 0; j < Nq; ++j) Qh(i, j) = .5 * (Q(i, j) - Q(j, i));
    for (int64_t i = 0; i < Nq; ++i)
      for (int64_t f = 0; f < Nfq; ++f) {
        const double b = wf(f, 0) * nJ(f, 0);
        Qh(i, Nq + f) = .5 * Ef(f, i) * b;
        Qh(Nq + f, i) = -.5 * b * Ef(f, i);
      }
    for (int64_t f = 0; f < Nfq; ++f) Qh(Nq + f, Nq + f) = .5 * wf(f, 0) * nJ(f, 0);
    Mat S(Nh, Nh);
    for (int64_t i = 0; i < Nh; ++i)
      for (int64_t j = 0; j < Nh; ++j) S(i, j) = .5 * (Qh(i, j) - Qh(j, i));
    A[on[d]] = S;
  }
  Mat Vh(Nh, Nq);
  for (int64_t q = 0; q < Nq; ++q) Vh(q, q) = 1.0;
  for (int64_t f = 0; f < Nfq; ++f)
    for (int64_t q = 0; q < Nq; ++q) Vh(Nq + f, q) = Ef(f, q);
  droptol(Vh, 1e-12);
  Mat Ph(Nq, Nh), Lf(Nq, Nfq);
  for (int64_t q = 0; q < Nq; ++q) {
    for (int64_t j = 0; j < Nh; ++j) Ph(q, j) = 2 * Vh(j, q) / wq(q, 0);      // the factor 2 of :96
    for (int64_t f = 0; f < Nfq; ++f) Lf(q, f) = Ef(f, q) * wf(f, 0) / wq(q, 0);
  }
  droptol(Ph, 1e-12); droptol(Lf, 1e-12);
  A["Ef"] = Ef; A["Vh"] = Vh; A["Ph"] = Ph; A["Lf"] = Lf;
}

struct Face4 { int64_t v[4]; int64_t id; };

}  // namespace

extern "C" {

int esdg_setup_uniform_hex_mesh(int Kx, int Ky, int Kz, double* VX, double* VY, double* VZ, int64_t* EToV) {
  if (Kx < 1 || Ky < 1 || Kz < 1 || !VX || !VY || !VZ || !EToV) return sfail("bad uniform hex mesh arguments");
  const int64_t Nxp = Kx + 1, Nyp = Ky + 1, Nzp = Kz + 1, K = (int64_t)Kx * Ky * Kz;
  for (int64_t k = 0; k < Nzp; ++k)
    for (int64_t j = 0; j < Nyp; ++j)
      for (int64_t i = 0; i < Nxp; ++i) {
        const int64_t v = i + Nxp * (j + Nyp * k);
        VX[v] = i == Nxp - 1 ? 1.0 : -1.0 + i * (2.0 / Kx);
        VY[v] = j == Nyp - 1 ? 1.0 : -1.0 + j * (2.0 / Ky);
        VZ[v] = k == Nzp - 1 ? 1.0 : -1.0 + k * (2.0 / Kz);
      }
  for (int64_t e = 0; e < K; ++e) {
    const int64_t k = e / ((int64_t)Kx * Ky), j = (e - k * Kx * Ky) / Kx, i = e % Kx;
    const int64_t v0 = i + Nxp * j + Nxp * Nyp * k + 1;
    const int64_t off[8] = {0, 1, Nxp, Nxp + 1, Nxp * Nyp, Nxp * Nyp + 1, Nxp * Nyp + Nxp, Nxp * Nyp + Nxp + 1};
    for (int v = 0; v < 8; ++v) EToV[(size_t)v * K + e] = v0 + off[v];
  }
  return ESDG_OK;
}

int esdg_setup_hex(int N, const double* VX, const double* VY, const double* VZ, int64_t Nv, const int64_t* EToV, int64_t Kg,
                   int periodic, int64_t e_begin, int64_t e_end, esdg_setup** out) {
  if (!out) return sfail("null output");
  *out = nullptr;
  if (N < 1 || N > 3 || !VX || !VY || !VZ || !EToV || Kg < 1 || Nv < 8) return sfail("bad hex set-up arguments");
  if (e_end <= 0) { e_begin = 0; e_end = Kg; }
  if (e_begin < 0 || e_end > Kg || e_begin >= e_end) return sfail("bad element range");
  esdg_setup* S = new esdg_setup();
  S->N = N; S->formulation = ESDG_EULER_HEX_COLLOCATED; S->Kglobal = Kg; S->e0 = e_begin; S->K = e_end - e_begin;
  auto& A = S->arr;
  reference_hex(N, A);
  hex_driver_ops(A);
  const int64_t K = S->K, e0 = e_begin;
  const Mat &V1 = A["V1"], &Vf = A["Vf"], &Vq = A["Vq"], &Dr = A["Dr"], &Ds = A["Ds"], &Dt = A["Dt"];
  const int Np = (int)V1.r, Nfq = (int)Vf.r, Nfp = Nfq / 6, Nq = (int)Vq.r;
  const double* VV[3] = {VX, VY, VZ};
  auto ev = [&](int64_t e, int v) { return EToV[(size_t)v * Kg + e] - 1; };
  for (int64_t e = 0; e < Kg; ++e)
    for (int v = 0; v < 8; ++v)
      if (ev(e, v) < 0 || ev(e, v) >= Nv) { delete S; return sfail("EToV entry out of range"); }

  // connect_mesh with the intended hex_face_vertices (0-based here)
  const int fv[6][4] = {{0, 1, 4, 5}, {2, 3, 6, 7}, {0, 2, 4, 6}, {1, 3, 5, 7}, {0, 1, 2, 3}, {4, 5, 6, 7}};
  std::vector<Face4> faces((size_t)Kg * 6);
  for (int64_t e = 0; e < Kg; ++e)
    for (int f = 0; f < 6; ++f) {
      Face4 F;
      for (int m = 0; m < 4; ++m) F.v[m] = ev(e, fv[f][m]);
      std::sort(F.v, F.v + 4);
      F.id = e * 6 + f;
      faces[(size_t)e * 6 + f] = F;
    }
  std::vector<Face4> sorted(faces);
  std::stable_sort(sorted.begin(), sorted.end(), [](const Face4& x, const Face4& y) {
    for (int m = 0; m < 4; ++m)
      if (x.v[m] != y.v[m]) return x.v[m] < y.v[m];
    return false;
  });
  std::vector<int64_t> FToF((size_t)Kg * 6);
  for (size_t i = 0; i < FToF.size(); ++i) FToF[i] = (int64_t)i;
  for (size_t i = 0; i + 1 < sorted.size(); ++i)
    if (std::equal(sorted[i].v, sorted[i].v + 4, sorted[i + 1].v)) {
      FToF[(size_t)sorted[i].id] = sorted[i + 1].id;
      FToF[(size_t)sorted[i + 1].id] = sorted[i].id;
    }

  Mat X[3] = {Mat(Np, K), Mat(Np, K), Mat(Np, K)};
  for (int64_t e = 0; e < K; ++e)
    for (int n = 0; n < Np; ++n)
      for (int c = 0; c < 3; ++c) {
        double sum = 0;
        for (int v = 0; v < 8; ++v) sum += V1(n, v) * VV[c][ev(e0 + e, v)];
        X[c](n, e) = sum;
      }
  Mat VfV1 = mul(Vf, V1);
  auto face_xyz = [&](int64_t gf, int i, double* p) {
    const int64_t e = gf / 6; const int f = (int)(gf % 6);
    for (int c = 0; c < 3; ++c) {
      p[c] = 0;
      for (int v = 0; v < 8; ++v) p[c] += VfV1(f * Nfp + i, v) * VV[c][ev(e, v)];
    }
  };
  std::vector<int64_t> mapM((size_t)K * Nfq), mapP((size_t)K * Nfq);
  std::vector<double> P1(3 * Nfp), P2(3 * Nfp);
  for (int64_t lf = 0; lf < K * 6; ++lf) {
    const int64_t f1 = e0 * 6 + lf, f2 = FToF[(size_t)f1];
    for (int i = 0; i < Nfp; ++i) { face_xyz(f1, i, &P1[3 * i]); face_xyz(f2, i, &P2[3 * i]); }
    auto dist = [&](int i, int j) { return std::fabs(P1[3 * i] - P2[3 * j]) + std::fabs(P1[3 * i + 1] - P2[3 * j + 1]) + std::fabs(P1[3 * i + 2] - P2[3 * j + 2]); };
    double refd = 0;
    for (int i = 0; i < Nfp; ++i)
      for (int j = 0; j < Nfp; ++j) refd = std::max(refd, dist(i, j));
    for (int i = 0; i < Nfp; ++i) {
      const int64_t m = f1 * Nfp + i + 1;
      mapM[(size_t)(lf * Nfp + i)] = m;
      int64_t p = m;
      if (f2 != f1)
        for (int j = 0; j < Nfp; ++j)
          if (dist(i, j) < 1e-10 * refd) { p = f2 * Nfp + j + 1; break; }
      mapP[(size_t)(lf * Nfp + i)] = p;
    }
  }
  std::vector<int64_t> mapB;
  for (size_t n = 0; n < mapM.size(); ++n)
    if (mapM[n] == mapP[n]) mapB.push_back(mapM[n]);

  if (periodic) {   // dg3D_euler_hex.jl:59-65
    double lo[3], hi[3];
    for (int c = 0; c < 3; ++c) {
      lo[c] = hi[c] = VV[c][0];
      for (int64_t v = 0; v < Nv; ++v) { lo[c] = std::min(lo[c], VV[c][v]); hi[c] = std::max(hi[c], VV[c][v]); }
    }
    struct BF { int64_t gf; double c[3]; };
    std::vector<BF> bf;
    for (int64_t gf = 0; gf < Kg * 6; ++gf)
      if (FToF[(size_t)gf] == gf) {
        BF b{gf, {0, 0, 0}};
        double p[3];
        for (int i = 0; i < Nfp; ++i) { face_xyz(gf, i, p); for (int c = 0; c < 3; ++c) b.c[c] += p[c] / Nfp; }
        bf.push_back(b);
      }
    std::map<int64_t, std::pair<int64_t, int>> partner;   // face -> (partner face, normal direction)
    for (int d = 0; d < 3; ++d) {
      const int a = d == 0 ? 1 : 0, b = d == 2 ? 1 : 2;
      const double L = hi[d] - lo[d], La = hi[a] - lo[a], Lb = hi[b] - lo[b];
      std::vector<BF> flo, fhi;
      for (const BF& f : bf) {
        if (std::fabs(f.c[d] - lo[d]) < 1e-12 * L) flo.push_back(f);
        else if (std::fabs(f.c[d] - hi[d]) < 1e-12 * L) fhi.push_back(f);
      }
      auto key = [&](const BF& p, const BF& q) {
        const double pa = std::round(p.c[a] / La * 1e9), qa = std::round(q.c[a] / La * 1e9);
        if (pa != qa) return pa < qa;
        return std::round(p.c[b] / Lb * 1e9) < std::round(q.c[b] / Lb * 1e9);
      };
      std::stable_sort(flo.begin(), flo.end(), key);
      std::stable_sort(fhi.begin(), fhi.end(), key);
      if (flo.size() != fhi.size()) { delete S; return sfail("periodic boundary faces do not pair up"); }
      for (size_t i = 0; i < flo.size(); ++i) { partner[flo[i].gf] = {fhi[i].gf, d}; partner[fhi[i].gf] = {flo[i].gf, d}; }
    }
    for (size_t n = 0; n < mapM.size(); ++n) {
      if (mapM[n] != mapP[n]) continue;
      const int64_t gnode = mapM[n] - 1, gf = gnode / Nfp;
      auto it = partner.find(gf);
      if (it == partner.end()) { delete S; return sfail("boundary face off the box"); }
      const int64_t pf = it->second.first;
      const int d = it->second.second, a = d == 0 ? 1 : 0, b = d == 2 ? 1 : 2;
      double p0[3], p1[3];
      face_xyz(gf, (int)(gnode % Nfp), p0);
      int64_t p = -1;
      for (int j = 0; j < Nfp; ++j) {
        face_xyz(pf, j, p1);
        if (std::fabs(p0[a] - p1[a]) + std::fabs(p0[b] - p1[b]) < 1e-9 * std::max(hi[a] - lo[a], hi[b] - lo[b])) { p = pf * Nfp + j + 1; break; }
      }
      if (p < 0) { delete S; return sfail("periodic node matching failed"); }
      mapP[n] = p;
    }
  }

  // geometric factors (curl form, src/geometric_factors.jl:34-67), normals, then the driver's post-processing (:88-98)
  const Mat &x = X[0], &y = X[1], &z = X[2];
  auto had = [](const Mat& a_, const Mat& b_) { Mat c_(a_.r, a_.c); for (size_t i = 0; i < c_.a.size(); ++i) c_.a[i] = a_.a[i] * b_.a[i]; return c_; };
  auto sub = [](const Mat& a_, const Mat& b_, double sgn) { Mat c_(a_.r, a_.c); for (size_t i = 0; i < c_.a.size(); ++i) c_.a[i] = sgn * (a_.a[i] - b_.a[i]); return c_; };
  auto curl = [&](const Mat& u, const Mat& w, double sgn, Mat& gr, Mat& gs, Mat& gt) {
    Mat Fr = had(mul(Dr, u), w), Fs = had(mul(Ds, u), w), Ft = had(mul(Dt, u), w);
    gr = sub(mul(Dt, Fs), mul(Ds, Ft), sgn);
    gs = sub(mul(Dr, Ft), mul(Dt, Fr), sgn);
    gt = sub(mul(Ds, Fr), mul(Dr, Fs), sgn);
  };
  Mat g[9];   // rxJ sxJ txJ ryJ syJ tyJ rzJ szJ tzJ
  curl(y, z, 1.0, g[0], g[1], g[2]);
  curl(x, z, -1.0, g[3], g[4], g[5]);
  curl(y, x, -1.0, g[6], g[7], g[8]);
  Mat xr = mul(Dr, x), xs = mul(Ds, x), xt = mul(Dt, x), yr = mul(Dr, y), ys = mul(Ds, y), yt = mul(Dt, y),
      zr = mul(Dr, z), zs = mul(Ds, z), zt = mul(Dt, z);
  Mat J(Np, K);
  for (size_t i = 0; i < J.a.size(); ++i)
    J.a[i] = xr.a[i] * (ys.a[i] * zt.a[i] - zs.a[i] * yt.a[i]) - yr.a[i] * (xs.a[i] * zt.a[i] - zs.a[i] * xt.a[i]) +
             zr.a[i] * (xs.a[i] * yt.a[i] - ys.a[i] * xt.a[i]);
  Mat fg[9];
  for (int m = 0; m < 9; ++m) fg[m] = mul(Vf, g[m]);
  Mat nxJ(Nfq, K), nyJ(Nfq, K), nzJ(Nfq, K), sJ(Nfq, K);
  for (int64_t e = 0; e < K; ++e)
    for (int f = 0; f < Nfq; ++f) {
      const double nr = A["nrJ"](f, 0), ns = A["nsJ"](f, 0), nt = A["ntJ"](f, 0);
      nxJ(f, e) = nr * fg[0](f, e) + ns * fg[1](f, e) + nt * fg[2](f, e);
      nyJ(f, e) = nr * fg[3](f, e) + ns * fg[4](f, e) + nt * fg[5](f, e);
      nzJ(f, e) = nr * fg[6](f, e) + ns * fg[7](f, e) + nt * fg[8](f, e);
      sJ(f, e) = std::sqrt(nxJ(f, e) * nxJ(f, e) + nyJ(f, e) * nyJ(f, e) + nzJ(f, e) * nzJ(f, e));
    }
  const int64_t Nh = Nq + Nfq;
  Mat Vhg(Nh, Np);
  for (int n = 0; n < Np; ++n) {
    for (int q = 0; q < Nq; ++q) Vhg(q, n) = Vq(q, n);
    for (int f = 0; f < Nfq; ++f) Vhg(Nq + f, n) = Vf(f, n);
  }
  const char* gn[9] = {"rxJ", "sxJ", "txJ", "ryJ", "syJ", "tyJ", "rzJ", "szJ", "tzJ"};
  for (int m = 0; m < 9; ++m) A[gn[m]] = mul(Vhg, g[m]);
  Mat Jq = mul(Vq, J), wJq(Nq, K);
  for (int64_t e = 0; e < K; ++e)
    for (int q = 0; q < Nq; ++q) wJq(q, e) = A["wq"](q, 0) * Jq(q, e);
  A["x"] = x; A["y"] = y; A["z"] = z; A["xq"] = mul(Vq, x); A["yq"] = mul(Vq, y); A["zq"] = mul(Vq, z);
  A["xf"] = mul(Vf, x); A["yf"] = mul(Vf, y); A["zf"] = mul(Vf, z);
  A["J"] = Jq; A["wJq"] = wJq; A["nxJ"] = nxJ; A["nyJ"] = nyJ; A["nzJ"] = nzJ; A["sJ"] = sJ;
  std::vector<int64_t> FToFl((size_t)K * 6);
  for (int64_t lf = 0; lf < K * 6; ++lf) FToFl[(size_t)lf] = FToF[(size_t)(e0 * 6 + lf)] + 1;
  S->maps["FToF"] = FToFl; S->maps["mapM"] = mapM; S->maps["mapP"] = mapP; S->maps["mapB"] = mapB;
  *out = S;
  return ESDG_OK;
}

int esdg_setup_fill_hex(const esdg_setup* s, esdg_hex_ops_t* ops, esdg_hex_mesh_t* mesh) {
  if (!s || !ops || !mesh || s->formulation != ESDG_EULER_HEX_COLLOCATED) return sfail("not a hex set-up");
  auto g = [&](const char* n) -> const double* { auto it = s->arr.find(n); return it == s->arr.end() ? nullptr : it->second.a.data(); };
  std::memset(ops, 0, sizeof *ops);
  std::memset(mesh, 0, sizeof *mesh);
  const int N1 = s->N + 1;
  ops->N = s->N; ops->Nq = N1 * N1 * N1; ops->Nfq = 6 * N1 * N1;
  ops->Qrhskew = g("Qrhskew"); ops->Qshskew = g("Qshskew"); ops->Qthskew = g("Qthskew"); ops->Ph = g("Ph"); ops->Lf = g("Lf");
  ops->Ef = g("Ef"); ops->wq = g("wq"); ops->wf = g("wf");
  mesh->K = s->K; mesh->geo_ld = ops->Nq + ops->Nfq;
  mesh->rxJ = g("rxJ"); mesh->sxJ = g("sxJ"); mesh->txJ = g("txJ"); mesh->ryJ = g("ryJ"); mesh->syJ = g("syJ"); mesh->tyJ = g("tyJ");
  mesh->rzJ = g("rzJ"); mesh->szJ = g("szJ"); mesh->tzJ = g("tzJ"); mesh->J = g("J"); mesh->wJq = g("wJq");
  mesh->nxJ = g("nxJ"); mesh->nyJ = g("nyJ"); mesh->nzJ = g("nzJ"); mesh->sJ = g("sJ");
  mesh->mapP = s->maps.at("mapP").data();
  mesh->elem_offset = s->e0; mesh->Kglobal = s->Kglobal; mesh->nranks = 1; mesh->rank = 0; mesh->rank_offsets = nullptr;
  return ESDG_OK;
}

}  // extern "C"
