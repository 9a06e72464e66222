"""Generates tests/golden/physics_kat.json: known answers for the fixed inputs of the reference's
own test-suite (examples/EntropyStableEuler.jl/test/runtests.jl:4-9, 55-127), evaluated in 50-digit
arithmetic (mpmath) from the closed-form definitions -- NOT from any code in this repository.

  python tests/golden/make_physics_kat.py
"""
import json
import os

from mpmath import mp, mpf, log, exp

mp.dps = 50
g = mpf("1.4")


def logmean(a, b):
    return (b - a) / (log(b) - log(a)) if a != b else a


def prim_to_cons(rho, u, v, p):
    return rho, rho * u, rho * v, p / (g - 1) + rho * (u * u + v * v) / 2


def v_of_u(rho, rhou, rhov, E):
    rhoe = E - (rhou ** 2 + rhov ** 2) / (2 * rho)
    s = log((g - 1) * rhoe / rho ** g)
    return (-E + rhoe * (g + 1 - s)) / rhoe, rhou / rhoe, rhov / rhoe, -rho / rhoe


def ec_flux(L, R):
    (rL, uL, vL, pL), (rR, uR, vR, pR) = L, R
    bL, bR = rL / (2 * pL), rR / (2 * pR)
    rlog, blog = logmean(rL, rR), logmean(bL, bR)
    ra, ua, va = (rL + rR) / 2, (uL + uR) / 2, (vL + vR) / 2
    un = uL * uR + vL * vR
    pa = ra / (bL + bR)
    f4 = rlog / (2 * (g - 1) * blog) + pa + rlog * un / 2
    Fx = (rlog * ua, rlog * ua * ua + pa, rlog * ua * va, f4 * ua)
    Fy = (rlog * va, rlog * ua * va, rlog * va * va + pa, f4 * va)
    return Fx, Fy


L = (mpf(1), mpf("0.1"), mpf("0.2"), mpf(2))          # runtests.jl:59
R = (mpf("1.1"), mpf("0.2"), mpf("0.3"), mpf("2.1"))  # runtests.jl:80
UL, UR = prim_to_cons(*L), prim_to_cons(*R)
VL, VR = v_of_u(*UL), v_of_u(*UR)
Fx, Fy = ec_flux(L, R)
psi = lambda U, d: (g - 1) * U[d]
out = {
    "comment": "50-digit closed-form values for the inputs of examples/EntropyStableEuler.jl/test/runtests.jl",
    "logmean_1_2": str(1 / log(mpf(2))),
    "primL": [str(x) for x in L], "primR": [str(x) for x in R],
    "UL": [str(x) for x in UL], "UR": [str(x) for x in UR],
    "VL": [str(x) for x in VL], "VR": [str(x) for x in VR],
    "betaL": str(L[0] / (2 * L[3])), "betaR": str(R[0] / (2 * R[3])),
    "Fx": [str(x) for x in Fx], "Fy": [str(x) for x in Fy],
    "exact_flux_x_L": [str(x) for x in (L[0] * L[1], L[0] * L[1] ** 2 + L[3], L[0] * L[1] * L[2], L[1] * (UL[3] + L[3]))],
    "exact_flux_y_L": [str(x) for x in (L[0] * L[2], L[0] * L[1] * L[2], L[0] * L[2] ** 2 + L[3], L[2] * (UL[3] + L[3]))],
    "psi_jump_x": str(psi(UL, 1) - psi(UR, 1)), "psi_jump_y": str(psi(UL, 2) - psi(UR, 2)),
    "vTFx": str(sum((a - b) * f for a, b, f in zip(VL, VR, Fx))),
    "vTFy": str(sum((a - b) * f for a, b, f in zip(VL, VR, Fy))),
}
with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "physics_kat.json"), "w") as fh:
    json.dump(out, fh, indent=1)
print("wrote physics_kat.json")
