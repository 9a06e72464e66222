"""Host-side restatement of the lane -> line maps of the line-per-lane kernels (kt3_rhs in esdg_kernels_tensor3.hip, kh_rhs_l in
esdg_kernels_hex.hip): every node of an element lies on exactly one line per direction, every pair of a line is met once, the flux
counts are the pair-once counts DESIGN.md quotes, and the odd LDS pitch of kh_rhs_l gives 16 consecutive lanes of any direction 16
distinct slots mod 16 (= distinct bank groups of a 16-byte record)."""
import itertools

import numpy as np
import pytest


# ---- 2D: t3::G3 and the line of lane (element, direction d, line o) ------------------------------------------------------------
def g3(N1):
    Nq, NLN = N1 * N1, 2 * N1
    E = 64 // NLN
    NV = E * Nq
    return dict(E=E, NV=NV, LL=E * NLN, NR=-(-NV // 64))


def line_nodes_2d(N1, lane):
    NLN, Nq = 2 * N1, N1 * N1
    el, lr = divmod(lane, NLN)
    d, o = divmod(lr, N1)
    n0, st = el * Nq + (o if d else N1 * o), (N1 if d else 1)
    return d, [n0 + i * st for i in range(N1)]


@pytest.mark.parametrize("N1", range(2, 9))
def test_2d_lines_cover_every_node_once_per_direction(N1):
    c = g3(N1)
    assert c["LL"] <= 64 and c["E"] >= 1 and c["NR"] * 64 >= c["NV"]
    seen = np.zeros((2, c["NV"]), dtype=int)
    for lane in range(c["LL"]):
        d, nodes = line_nodes_2d(N1, lane)
        for n in nodes:
            seen[d, n] += 1
    assert (seen == 1).all()
    # fluxes of a line: C(N1,2) volume-volume + 2 N1 volume-face + 2 interface; per element: the pair-once counts of DESIGN.md
    per_line = N1 * (N1 - 1) // 2 + 2 * N1 + 2
    vv, vf, itf = 2 * N1 * (N1 * (N1 - 1) // 2), 2 * N1 * 2 * N1, 4 * N1
    assert 2 * N1 * per_line == vv + vf + itf
    if N1 == 5:
        assert per_line == 22 and vv + vf == 200 and itf == 20 and c["E"] == 6 and c["NR"] == 3


# ---- 3D: hdev::LCfg, line_of, line_slots ----------------------------------------------------------------------------------------
def lcfg(N1, E5=3, E6=2):
    E = {2: 16, 3: 7, 4: 4, 5: E5, 6: E6}.get(N1, 1)
    NN = N1 * N1
    P = N1 + 1 if N1 % 2 == 0 else N1
    T = -(-3 * E * NN // 64) * 64
    return dict(E=E, NN=NN, Nq=NN * N1, P=P, NQP=P * NN, LLD=E * NN, T=T)


def line_of(N1, d, o):
    NN = N1 * N1
    stride = 1 if d == 0 else (N1 if d == 1 else NN)
    base = N1 * o if d == 0 else ((o % N1) + NN * (o // N1) if d == 1 else o)
    return base, stride


def line_slots(N1, d, o):
    P = lcfg(N1)["P"]
    stride = 1 if d == 0 else (P if d == 1 else P * N1)
    base = P * o if d == 0 else ((o % N1) + P * N1 * (o // N1) if d == 1 else (o % N1) + P * (o // N1))
    return base, stride


def slot_of_node(N1, q):
    P = lcfg(N1)["P"]
    return q + (q // N1 if P != N1 else 0)


@pytest.mark.parametrize("N1", range(2, 9))
def test_3d_lines_cover_every_node_once_per_direction_and_slots_match_the_node_rounds(N1):
    c = lcfg(N1)
    assert 3 * c["LLD"] <= c["T"] <= 3 * c["LLD"] + 63
    for d in range(3):
        seen = np.zeros(c["Nq"], dtype=int)
        for o in range(c["NN"]):
            (b, st), (sb, sst) = line_of(N1, d, o), line_slots(N1, d, o)
            for i in range(N1):
                node = b + i * st
                seen[node] += 1
                assert sb + i * sst == slot_of_node(N1, node) < c["NQP"]      # what the line lanes read is what the node lanes wrote
        assert (seen == 1).all()
    slots = [slot_of_node(N1, q) for q in range(c["Nq"])]
    assert len(set(slots)) == c["Nq"]
    per_line = N1 * (N1 - 1) // 2 + 2 * N1 + 2
    if N1 == 4:
        assert per_line == 16 and 3 * c["NN"] * (N1 * (N1 - 1) // 2 + 2 * N1) == 672 and 3 * c["NN"] * 2 == 96


@pytest.mark.parametrize("N1", [4, 6, 8])
def test_odd_pitch_spreads_the_lanes_of_every_direction_over_the_banks(N1):
    """16 consecutive lines of one direction, same position i along the line: with pitch P = N1 + 1 their slots are pairwise
    distinct mod 16 (a ds_read_b128 serves 16 lanes per LDS cycle group, 16 bytes each = the 64 banks once) for the directions whose
    lanes stride through memory; without the padding the direction-0 lanes of N1 = 4 share banks four to one."""
    c = lcfg(N1)
    worst_padded = worst_plain = 1
    for d in range(3):
        for o0 in range(0, c["NN"] - 15, 16):
            for i in range(N1):
                padded = [(line_slots(N1, d, o)[0] + i * line_slots(N1, d, o)[1]) % 16 for o in range(o0, o0 + 16)]
                plain = [(line_of(N1, d, o)[0] + i * line_of(N1, d, o)[1]) % 16 for o in range(o0, o0 + 16)]
                worst_padded = max(worst_padded, max(np.bincount(padded)))
                worst_plain = max(worst_plain, max(np.bincount(plain)))
    assert worst_padded <= 2          # (direction 1 wraps once per N1 lines: at most two lanes per bank group)
    # without the padding: four lanes per bank group at N1 = 4 (the conflicts the SQ counters showed, profiles/experiments/README.md),
    # eight at N1 = 8, two at N1 = 6
    assert worst_plain == {4: 4, 6: 2, 8: 8}[N1]
