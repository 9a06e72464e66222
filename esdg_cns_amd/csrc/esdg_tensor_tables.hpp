// esdg_tensor_tables.hpp -- layout of the 1D operator tables of the tensor kernels, shared by the host
// (esdg_api.hip builds and verifies them from the driver's dense matrices) and the device
// (esdg_kernels_tensor.hip stages them in LDS once per workgroup).
//
// Conventions: volume (Gauss) node q = a + N1*b.  Direction d = 0 walks a (stride 1), d = 1 walks b
// (stride N1).  node(d,i,o) is the node at position i of the line with transverse index o.
//   Q_op(d)[node(d,i,o), node(d,j,o)]          = S[d][i][j]   * WT[d][o]         (volume-volume SBP weight)
//   Q_op(d)[node(d,i,o), Nq + FN[d][t][o]]     = SF[d][t][i]  * WTF[d][t][o]     (volume-face weight, t = 0,1)
//   (Vq*Ph)[node(d,i,o), Nq + FN[d][t][o]]     = PF[d][t][i]  * PTF[d][t][o]     (collocated projection)
//   (Vq*LIFT)[q, f]                            = (Vq*Ph)[q, Nq+f] * WFAC[f]      (collocated lift)
//   (Vq*Ph)[q, q]                              = PD[q]
//   Ef[FN[d][t][o], node(d,i,o)]               = EE[d][t][i]                      (face interpolation)
//   (Vq*D_op(d)*Pq)[node(d,i,o), node(d,j,o)]  = DG[d][i][j]                      (collocated derivative)
//   Vq = IQ (x) IQ,  Pq = IP (x) IP                                               (modal <-> Gauss)
// where op(d) in {0: the "r" operators (Qr, Dr, metrics rxJ, ryJ), 1: the "s" operators}.
#pragma once

namespace esdg {

struct TensorLayout {
  int N1, S, WT, SF, WTF, PF, PTF, PD, WFAC, EE, DG, IQ, IP, NDBL;  // offsets in doubles
  int FN, FINV, NINT;                                                // offsets in int32
  __host__ __device__ constexpr explicit TensorLayout(int n1)
      : N1(n1),
        S(0),
        WT(S + 2 * n1 * n1),
        SF(WT + 2 * n1),
        WTF(SF + 4 * n1),
        PF(WTF + 4 * n1),
        PTF(PF + 4 * n1),
        PD(PTF + 4 * n1),
        WFAC(PD + n1 * n1),
        EE(WFAC + 4 * n1),
        DG(EE + 4 * n1),
        IQ(DG + 2 * n1 * n1),
        IP(IQ + n1 * n1),
        NDBL(IP + n1 * n1),
        FN(0),
        FINV(4 * n1),   // packed d | t<<1 | o<<2 per face node
        NINT(8 * n1) {}
};

// device-side handle: one buffer of NDBL doubles followed by NINT int32
struct TensorTables {
  const double* dbl;
  const int* ints;
  int op0, op1;  // operator family of direction 0 / 1
};

}  // namespace esdg
