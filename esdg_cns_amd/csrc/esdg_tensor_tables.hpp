// esdg_tensor_tables.hpp -- layout of the 1D operator tables of the tensor kernels, shared by the host
// (esdg_api.hip builds and verifies them from the driver's dense matrices) and the device
// (kt3_rhs stages them in LDS once per workgroup; kt2_* read the per-node rows derived from them).
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

// Per-node tables of the v2 tensor kernels (esdg_kernels_tensor2.hip): everything a lane needs about ITS node as one
// row, derived on the host from the 1D tables above, so the kernels do no index arithmetic or table chasing.
// Faces are numbered k = 2 d + t (the face at end t of the d-lines); circulant round r = d * NFULL + i pairs the node
// at position pos of its d-line with position (pos + i + 1) mod N1; even N1 has one more, antipodal round (pos + N1/2)
// in which a node serves ONE of its two directions (see kt2_rhs).
//   volume node q = a + N1 b, NodeLayout::LD doubles:
//     IQ[N1] = IQ[a][:]          IPL[N1] = IP[a][:]      IPH[N1] = IP[b][:]
//     DG0[N1] = DG[0][a][:]      DG1[N1] = DG[1][b][:]
//     LW[4]  = PF*PTF*WFAC of face k (collocated lift)      PW[4] = PF*PTF (collocated projection)
//     SVF[4] = SF*WTF (volume-face SBP weight)              SVV[NRND] = S*WT of round r (volume-volume SBP weight)
//     PD     = (Vq*Ph)[q,q]
//   and NodeLayout::LI ints: FQ[4] = face node of face k, PID[NRND] = partner node of round r,
//     AD = direction served in the antipodal round (even N1)
//   face node fn, FaceLayout::LD doubles: EE[N1] (interpolation weights along its line), WFAC,
//     SVF[N1] = SF*WTF: volume-face SBP weight towards the j-th node of its line (the pairs as the FACE node sees them)
//   and FaceLayout::LI ints: NODE0 = first node of its line, STRIDE = node stride along the line, K = face number k
struct NodeLayout {
  int N1, NFULL, NRND, IQ, IPL, IPH, DG0, DG1, LW, PW, SVF, SVV, PD, LD;
  int FQ, PID, AD, LI;
  __host__ __device__ constexpr explicit NodeLayout(int n1)
      : N1(n1), NFULL((n1 - 1) / 2), NRND(2 * ((n1 - 1) / 2) + (n1 % 2 == 0 ? 1 : 0)),
        IQ(0), IPL(n1), IPH(2 * n1), DG0(3 * n1), DG1(4 * n1), LW(5 * n1), PW(5 * n1 + 4), SVF(5 * n1 + 8), SVV(5 * n1 + 12),
        PD(5 * n1 + 12 + 2 * ((n1 - 1) / 2) + (n1 % 2 == 0 ? 1 : 0)),
        LD(5 * n1 + 13 + 2 * ((n1 - 1) / 2) + (n1 % 2 == 0 ? 1 : 0)),
        FQ(0), PID(4), AD(4 + 2 * ((n1 - 1) / 2) + (n1 % 2 == 0 ? 1 : 0)), LI(5 + 2 * ((n1 - 1) / 2) + (n1 % 2 == 0 ? 1 : 0)) {}
};
struct FaceLayout {
  int N1, EE, WFAC, SVF, LD, NODE0, STRIDE, K, LI;
  __host__ __device__ constexpr explicit FaceLayout(int n1)
      : N1(n1), EE(0), WFAC(n1), SVF(n1 + 1), LD(2 * n1 + 1), NODE0(0), STRIDE(1), K(2), LI(3) {}
};

// Packed rows of the one-shot last-phase kernel kt2_rhs, which reloads its rows in every workgroup: a lane fetches its
// doubles as NPV (volume node) / NPF (face node) coalesced 16-byte loads from pair planes [pair][node] and its indices as
// one int4 / one int:
//   volume doubles: SVV[NRND], PW[4], PD         face doubles: SVF[N1], WFAC
//   volume int4: { PID[0..3] bytes, PID[4..7] bytes, FQ[0..3] bytes, AD }      face int: NODE0 | STRIDE << 8 | K << 16
struct RhsRows {
  int NRND, NDV, NPV, NDF, NPF;
  __host__ __device__ constexpr explicit RhsRows(int n1)
      : NRND(2 * ((n1 - 1) / 2) + (n1 % 2 == 0 ? 1 : 0)), NDV(NRND + 5), NPV((NRND + 6) / 2), NDF(n1 + 1), NPF((n1 + 2) / 2) {}
};

// device-side handle: one buffer of NDBL doubles followed by NINT int32
struct TensorTables {
  const double* dbl;
  const int* ints;
  int op0, op1;  // operator family of direction 0 / 1
  // v2 kernels: per-node rows (NodeLayout / FaceLayout) and the mesh face (0..3, position in the normals of the geometry
  // record) of face k = 2 d + t
  const double* node_d;
  const int* node_i;
  const double* face_d;
  const int* face_i;
  int gface[4];
  const double* rhs_vd;   // RhsRows: [NPV][Nq] double2
  const int* rhs_vi;      // [Nq] int4
  const double* rhs_fd;   // [NPF][Nfq] double2
  const int* rhs_fi;      // [Nfq] int
};

}  // namespace esdg
