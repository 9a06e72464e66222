"""Generates tests/golden/physics_kat_3d.json: known answers for the fixed inputs of the reference's
3D test set (examples/EntropyStableEuler.jl/test/runtests.jl:130-200), evaluated in 50-digit arithmetic
(mpmath) from the closed-form definitions -- NOT from any code in this repository.

  python tests/golden/make_physics_kat_3d.py
"""
import json
import os

from mpmath import mp, mpf, log

mp.dps = 50
g = mpf("1.4")


def logmean(a, b):
    return (b - a) / (log(b) - log(a)) if a != b else a


def prim_to_cons(rho, u, v, w, p):
    return rho, rho * u, rho * v, rho * w, p / (g - 1) + rho * (u * u + v * v + w * w) / 2


def v_of_u(rho, rhou, rhov, rhow, E):
    rhoe = E - (rhou ** 2 + rhov ** 2 + rhow ** 2) / (2 * rho)
    s = log((g - 1) * rhoe / rho ** g)
    return (-E + rhoe * (g + 1 - s)) / rhoe, rhou / rhoe, rhov / rhoe, rhow / rhoe, -rho / rhoe


def ec_flux(L, R):
    (rL, uL, vL, wL, pL), (rR, uR, vR, wR, pR) = L, R
    bL, bR = rL / (2 * pL), rR / (2 * pR)
    rlog, blog = logmean(rL, rR), logmean(bL, bR)
    ra, ua, va, wa = (rL + rR) / 2, (uL + uR) / 2, (vL + vR) / 2, (wL + wR) / 2
    un = uL * uR + vL * vR + wL * wR
    pa = ra / (bL + bR)
    f5 = rlog / (2 * (g - 1) * blog) + pa + rlog * un / 2
    Fx = (rlog * ua, rlog * ua * ua + pa, rlog * ua * va, rlog * ua * wa, f5 * ua)
    Fy = (rlog * va, rlog * ua * va, rlog * va * va + pa, rlog * va * wa, f5 * va)
    Fz = (rlog * wa, rlog * ua * wa, rlog * va * wa, rlog * wa * wa + pa, f5 * wa)
    return Fx, Fy, Fz


L = (mpf(1), mpf("0.1"), mpf("0.2"), mpf("0.3"), mpf(2))            # runtests.jl:134
R = (mpf("1.1"), mpf("0.2"), mpf("0.3"), mpf("0.4"), mpf("2.1"))    # runtests.jl:156
UL, UR = prim_to_cons(*L), prim_to_cons(*R)
VL, VR = v_of_u(*UL), v_of_u(*UR)
Fx, Fy, Fz = ec_flux(L, R)
rho, u, v, w, p = L
E = UL[4]
S = lambda x: [str(t) for t in x]
out = {
    "comment": "50-digit closed-form values for the 3D inputs of examples/EntropyStableEuler.jl/test/runtests.jl:130-200",
    "primL": S(L), "primR": S(R), "UL": S(UL), "UR": S(UR), "VL": S(VL), "VR": S(VR),
    "betaL": str(L[0] / (2 * L[4])), "betaR": str(R[0] / (2 * R[4])),
    "Fx": S(Fx), "Fy": S(Fy), "Fz": S(Fz),
    "exact_flux_x_L": S((rho * u, rho * u * u + p, rho * u * v, rho * u * w, u * (E + p))),
    "exact_flux_y_L": S((rho * v, rho * v * u, rho * v * v + p, rho * v * w, v * (E + p))),
    "exact_flux_z_L": S((rho * w, rho * w * u, rho * w * v, rho * w * w + p, w * (E + p))),
    "psi_jump": S([(g - 1) * (UL[d] - UR[d]) for d in (1, 2, 3)]),
}
with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "physics_kat_3d.json"), "w") as fh:
    json.dump(out, fh, indent=1)
print("wrote physics_kat_3d.json")
