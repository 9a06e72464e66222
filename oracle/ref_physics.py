"""ORACLE (test infrastructure, NOT product code) -- numpy restatement of the reference's
pointwise physics, module examples/EntropyStableEuler/ (gamma = 1.4).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this.
Pinned by the reference's own property tests (examples/EntropyStableEuler.jl/test/runtests.jl)
re-stated in tests/test_oracle_physics.py with closed-form expected values.
"""
import numpy as np

GAMMA = 1.4   # examples/EntropyStableEuler/EntropyStableEuler.jl:9


def logmean(aL, aR, logL=None, logR=None):
    """examples/EntropyStableEuler/logmean.jl:5-28."""
    aL = np.asarray(aL, dtype=float)
    aR = np.asarray(aR, dtype=float)
    if logL is None:
        logL, logR = np.log(aL), np.log(aR)
    da = aR - aL
    aavg = 0.5 * (aR + aL)
    f = da / aavg
    v = f ** 2
    series = aavg * (1 + v * (-.2 - v * (.0512 - v * 0.026038857142857)))
    with np.errstate(divide="ignore", invalid="ignore"):
        exact = -da / (logL - logR)
    return np.where(np.abs(f) < 1e-4, series, exact)


def euler_fluxes_2D(rhoL, uL, vL, betaL, rhoR, uR, vR, betaR, rhologL, betalogL, rhologR, betalogR):
    """examples/EntropyStableEuler/euler_fluxes.jl:23-48."""
    rholog = logmean(rhoL, rhoR, rhologL, rhologR)
    betalog = logmean(betaL, betaR, betalogL, betalogR)
    rhoavg = .5 * (rhoL + rhoR)
    uavg = .5 * (uL + uR)
    vavg = .5 * (vL + vR)
    unorm = uL * uR + vL * vR
    pa = rhoavg / (betaL + betaR)
    f4aux = rholog / (2 * (GAMMA - 1) * betalog) + pa + .5 * rholog * unorm
    FxS1 = rholog * uavg
    FxS2 = FxS1 * uavg + pa
    FxS3 = FxS1 * vavg
    FxS4 = f4aux * uavg
    FyS1 = rholog * vavg
    FyS2 = FxS3
    FyS3 = FyS1 * vavg + pa
    FyS4 = f4aux * vavg
    return (FxS1, FxS2, FxS3, FxS4), (FyS1, FyS2, FyS3, FyS4)


def euler_fluxes_UL_UR(UL, UR):
    """examples/EntropyStableEuler/euler_fluxes.jl:9-15 (logs computed on the fly)."""
    return euler_fluxes_2D(*UL, *UR, np.log(UL[0]), np.log(UL[3]), np.log(UR[0]), np.log(UR[3]))


def pfun(rho, rhou, rhov, E):
    """euler_variables.jl:42-48."""
    rhounorm = (rhou ** 2 + rhov ** 2) / rho
    return (GAMMA - 1) * (E - .5 * rhounorm)


def betafun(rho, rhou, rhov, E):
    """euler_variables.jl:30-36."""
    return rho / (2 * pfun(rho, rhou, rhov, E))


def wavespeed(rho, rhou, E):
    """euler_variables.jl:7-10 -- 1D call; note sqrt(abs(u_n)) (quirk Q1)."""
    p = (GAMMA - 1) * (E - .5 * (rhou ** 2) / rho)
    cvel = np.sqrt(GAMMA * p / rho)
    return np.sqrt(np.abs(rhou / rho)) + cvel


def primitive_to_conservative(rho, u, v, p):
    """euler_variables.jl:15-24."""
    unorm = u ** 2 + v ** 2
    E = p / (GAMMA - 1) + .5 * rho * unorm
    return rho, rho * u, rho * v, E


def rhoefun(rho, rhou, rhov, E):
    """euler_variables.jl:59-62."""
    return E - .5 * (rhou ** 2 + rhov ** 2) / rho


def sfun(rho, rhou, rhov, E):
    """euler_variables.jl:65-68."""
    return np.log((GAMMA - 1) * rhoefun(rho, rhou, rhov, E) / rho ** GAMMA)


def Sfun(rho, rhou, rhov, E):
    """euler_variables.jl:71-73."""
    return -rho * sfun(rho, rhou, rhov, E)


def v_ufun(rho, rhou, rhov, E):
    """euler_variables.jl:79-89."""
    rhoe = rhoefun(rho, rhou, rhov, E)
    sU = sfun(rho, rhou, rhov, E)
    v1 = (-E + rhoe * (GAMMA + 1 - sU)) / rhoe
    return v1, rhou / rhoe, rhov / rhoe, (-rho) / rhoe


def u_vfun(v1, v2, v3, v4):
    """euler_variables.jl:95-117."""
    vUnorm = v2 ** 2 + v3 ** 2
    s = GAMMA - v1 + vUnorm / (2 * v4)
    rhoeV = ((GAMMA - 1) / ((-v4) ** GAMMA)) ** (1 / (GAMMA - 1)) * np.exp(-s / (GAMMA - 1))
    return rhoeV * (-v4), rhoeV * v2, rhoeV * v3, rhoeV * (1 - vUnorm / (2 * v4))


def vortex(x, y, t, gamma=1.4):
    """examples/EntropyStableEuler/EntropyStableEuler.jl:21-35."""
    x0, y0, beta = 5, 0, 5
    r2 = (x - x0 - t) ** 2 + (y - y0) ** 2
    u = 1 - beta * np.exp(1 - r2) * (y - y0) / (2 * np.pi)
    v = beta * np.exp(1 - r2) * (x - x0 - t) / (2 * np.pi)
    rho = 1 - (1 / (8 * gamma * np.pi ** 2)) * (gamma - 1) / 2 * (beta * np.exp(1 - r2)) ** 2
    rho = rho ** (1 / (gamma - 1))
    p = rho ** gamma
    return rho, u, v, p


# ----------------------------------------------------------------------------------
# 3D members of the same module (used by examples/dg3D_euler_hex.jl)
# ----------------------------------------------------------------------------------
def euler_fluxes_3D(rhoL, uL, vL, wL, betaL, rhoR, uR, vR, wR, betaR, rhologL, betalogL, rhologR, betalogR):
    """examples/EntropyStableEuler/euler_fluxes.jl:51-89."""
    rholog = logmean(rhoL, rhoR, rhologL, rhologR)
    betalog = logmean(betaL, betaR, betalogL, betalogR)
    rhoavg = .5 * (rhoL + rhoR)
    uavg = .5 * (uL + uR)
    vavg = .5 * (vL + vR)
    wavg = .5 * (wL + wR)
    unorm = uL * uR + vL * vR + wL * wR
    pa = rhoavg / (betaL + betaR)
    E_plus_p = rholog / (2 * (GAMMA - 1) * betalog) + pa + .5 * rholog * unorm
    FxS1 = rholog * uavg
    FxS2 = FxS1 * uavg + pa
    FxS3 = FxS1 * vavg
    FxS4 = FxS1 * wavg
    FxS5 = E_plus_p * uavg
    FyS1 = rholog * vavg
    FyS2 = FxS3
    FyS3 = FyS1 * vavg + pa
    FyS4 = FyS1 * wavg
    FyS5 = E_plus_p * vavg
    FzS1 = rholog * wavg
    FzS2 = FxS4
    FzS3 = FyS4
    FzS4 = FzS1 * wavg + pa
    FzS5 = E_plus_p * wavg
    return (FxS1, FxS2, FxS3, FxS4, FxS5), (FyS1, FyS2, FyS3, FyS4, FyS5), (FzS1, FzS2, FzS3, FzS4, FzS5)


def euler_fluxes_UL_UR_3D(UL, UR):
    """euler_fluxes.jl:9-20 with 5-tuples (rho,u,v,w,beta): logs computed on the fly."""
    return euler_fluxes_3D(*UL, *UR, np.log(UL[0]), np.log(UL[4]), np.log(UR[0]), np.log(UR[4]))


def pfun_3D(rho, rhou, rhov, rhow, E):
    """euler_variables.jl:42-48 with rhoU = (rhou,rhov,rhow)."""
    rhounorm = (rhou ** 2 + rhov ** 2 + rhow ** 2) / rho
    return (GAMMA - 1) * (E - .5 * rhounorm)


def betafun_3D(rho, rhou, rhov, rhow, E):
    """euler_variables.jl:30-39."""
    return rho / (2 * pfun_3D(rho, rhou, rhov, rhow, E))


def primitive_to_conservative_3D(rho, u, v, w, p):
    """euler_variables.jl:15-27."""
    unorm = u ** 2 + v ** 2 + w ** 2
    return rho, rho * u, rho * v, rho * w, p / (GAMMA - 1) + .5 * rho * unorm


def rhoefun_3D(rho, rhou, rhov, rhow, E):
    """euler_variables.jl:59-62."""
    return E - .5 * (rhou ** 2 + rhov ** 2 + rhow ** 2) / rho


def sfun_3D(rho, rhou, rhov, rhow, E):
    """euler_variables.jl:65-68."""
    return np.log((GAMMA - 1) * rhoefun_3D(rho, rhou, rhov, rhow, E) / rho ** GAMMA)


def Sfun_3D(rho, rhou, rhov, rhow, E):
    """euler_variables.jl:74-76."""
    return -rho * sfun_3D(rho, rhou, rhov, rhow, E)


def v_ufun_3D(rho, rhou, rhov, rhow, E):
    """euler_variables.jl:79-92."""
    rhoe = rhoefun_3D(rho, rhou, rhov, rhow, E)
    sU = sfun_3D(rho, rhou, rhov, rhow, E)
    v1 = (-E + rhoe * (GAMMA + 1 - sU)) / rhoe
    return v1, rhou / rhoe, rhov / rhoe, rhow / rhoe, (-rho) / rhoe


def u_vfun_3D(v1, v2, v3, v4, v5):
    """euler_variables.jl:95-120."""
    vUnorm = v2 ** 2 + v3 ** 2 + v4 ** 2
    s = GAMMA - v1 + vUnorm / (2 * v5)
    rhoeV = ((GAMMA - 1) / ((-v5) ** GAMMA)) ** (1 / (GAMMA - 1)) * np.exp(-s / (GAMMA - 1))
    return rhoeV * (-v5), rhoeV * v2, rhoeV * v3, rhoeV * v4, rhoeV * (1 - vUnorm / (2 * v5))
