// esdg_hex_tables.hpp -- layout of the 1D operator tables of the hexahedral kernels (esdg_kernels_hex.hip),
// shared by the host (esdg_api.hip builds and verifies them entry by entry from the dense matrices the
// driver passes) and the device (staged in LDS once per workgroup).
//
// Conventions: Gauss node q = i0 + N1*i1 + N1^2*i2 (s fastest, then r, then t: vec.(meshgrid(r1D,r1D,r1D)),
// src/SetupDG.jl:361).  Direction d walks i_d (stride N1^d); a line of direction d is named by its transverse
// index o = the remaining two indices, lower stride first.  node(d,i,o) is the node at position i of line o.
//   Q_op(d)[node(d,i,o), node(d,j,o)]        = S[d][i][j]  * WT[d][o]        (volume-volume SBP weight)
//   Q_op(d)[node(d,i,o), Nq + FN[d][t][o]]   = SF[d][t][i] * WTF[d][t][o]    (volume-face weight, t = 0,1 line ends)
//   Ph[node(d,i,o), Nq + FN[d][t][o]]        = PF[d][t][i] * PTF[d][t][o]
//   Lf[q, f] = Ph[q, Nq+f] * WFAC[f],   Ph[q, q] = PD[q],   Ef[FN[d][t][o], node(d,i,o)] = EE[d][t][i]
// op(d) in {0,1,2}: which of (Qrhskew, Qshskew, Qthskew) / metric rows (r, s, t) belongs to direction d.
#pragma once

namespace esdg {

struct HexLayout {
  int N1, S, WT, SF, WTF, PF, PTF, PD, WFAC, EE, NDBL;  // offsets in doubles
  int FN, FINV, NINT;                                    // offsets in int32
  __host__ __device__ constexpr explicit HexLayout(int n)
      : N1(n),
        S(0),
        WT(S + 3 * n * n),
        SF(WT + 3 * n * n),
        WTF(SF + 6 * n),
        PF(WTF + 6 * n * n),
        PTF(PF + 6 * n),
        PD(PTF + 6 * n * n),
        WFAC(PD + n * n * n),
        EE(WFAC + 6 * n * n),
        NDBL(EE + 6 * n),
        FN(0),
        FINV(6 * n * n),   // per face node: d | t<<2 | o<<3
        NINT(12 * n * n) {}
};

struct HexTables {
  const double* dbl;
  const int* ints;
  int op[3];  // operator family of direction 0/1/2
};

// Per-element affine geometry record of the hex path (doubles):
//   [0..8]  rxJ sxJ txJ ryJ syJ tyJ rzJ szJ tzJ   (first row of the driver's metric arrays)
//   [9]     J at the quadrature nodes (dg3D_euler_hex.jl:94)
//   [10+4f .. 13+4f]  nxJ nyJ nzJ sJ of face f
//   [34] scale of the packed per-node metric differences, [35] scale of the packed per-node normal differences (geometry
//        mode 2 of kh_rhs, MeshDev::hdv / hdf / hdn; 0 otherwise)
constexpr int HEX_GEO_STRIDE = 36;
constexpr int HEX_NFLD = 5;
constexpr int HEX_AU_NC = 5;  // face trace record: (rho, u, v, w, beta)

template <int N1>
__host__ __device__ constexpr int hex_node(int d, int i, int o) {
  return d == 0 ? i + N1 * o : (d == 1 ? (o % N1) + N1 * i + N1 * N1 * (o / N1) : o + N1 * N1 * i);
}
inline int hex_node_rt(int N1, int d, int i, int o) {
  return d == 0 ? i + N1 * o : (d == 1 ? (o % N1) + N1 * i + N1 * N1 * (o / N1) : o + N1 * N1 * i);
}

}  // namespace esdg
