"""Host-side restatement of two workgroup -> element-group maps of esdg_kernels_tensor2.hip (A/B hook ESDG_T2_XCD_REMAP, off by
default; measured late in round 3, profiles/experiments/README.md): whatever the grid size, every group must be visited exactly once."""
import numpy as np
import pytest

NXCD = 8


def xcd_group(b, n):
    """t2::xcd_group: the workgroups of XCD x = b % 8 walk the x-th contiguous eighth of the n groups."""
    if n < 8 * NXCD:
        return b
    x, q, r = b % NXCD, n // NXCD, n % NXCD
    return x * q + min(x, r) + b // NXCD


def sigma_groups(w, G, nfull):
    """kt2_sigma's persistent walk with the remap: workgroup w of G takes groups i, i + nx, ... of its XCD's chunk."""
    x, i = w % NXCD, w // NXCD
    nx = G // NXCD + (1 if x < G % NXCD else 0)
    q, r = nfull // NXCD, nfull % NXCD
    start, count = x * q + min(x, r), q + (1 if x < r else 0)
    return [start + j for j in range(i, count, nx)]


@pytest.mark.parametrize("n", [1, 7, 63, 64, 65, 71, 1000, 13108, 52429])
def test_one_shot_remap_is_a_bijection(n):
    g = np.array([xcd_group(b, n) for b in range(n)])
    assert np.array_equal(np.sort(g), np.arange(n))


@pytest.mark.parametrize("G,nfull", [(1024, 52429), (1024, 1024), (960, 13108), (64, 64), (8, 100), (1536, 6554), (1000, 1003)])
def test_persistent_remap_covers_every_group_once(G, nfull):
    seen = np.zeros(nfull, dtype=int)
    for w in range(G):
        for g in sigma_groups(w, G, nfull):
            seen[g] += 1
    assert seen.min() == 1 and seen.max() == 1
