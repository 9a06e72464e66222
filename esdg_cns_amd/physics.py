"""Host-side helpers mirroring the exported names of examples/EntropyStableEuler/EntropyStableEuler.jl
that drivers call outside the hot loop (initial conditions and variable changes), in numpy.
The hot-path versions of these formulas live in csrc/esdg_kernels.hip."""
import numpy as np

GAMMA = 1.4  # EntropyStableEuler.jl:9


def vortex(x, y, t, gamma=1.4):
    """Isentropic vortex, EntropyStableEuler.jl:21-35 (x0=5, y0=0, beta=5)."""
    x0, y0, beta = 5, 0, 5
    r2 = (x - x0 - t) ** 2 + (y - y0) ** 2
    u = 1 - beta * np.exp(1 - r2) * (y - y0) / (2 * np.pi)
    v = beta * np.exp(1 - r2) * (x - x0 - t) / (2 * np.pi)
    rho = (1 - (1 / (8 * gamma * np.pi ** 2)) * (gamma - 1) / 2 * (beta * np.exp(1 - r2)) ** 2) ** (1 / (gamma - 1))
    return rho, u, v, rho ** gamma


def primitive_to_conservative(rho, u, v, p):
    """euler_variables.jl:15-24."""
    return rho, rho * u, rho * v, p / (GAMMA - 1) + .5 * rho * (u ** 2 + v ** 2)


def v_ufun(rho, rhou, rhov, E):
    """Entropy variables, euler_variables.jl:79-89."""
    rhoe = E - .5 * (rhou ** 2 + rhov ** 2) / rho
    s = np.log((GAMMA - 1) * rhoe / rho ** GAMMA)
    return (-E + rhoe * (GAMMA + 1 - s)) / rhoe, rhou / rhoe, rhov / rhoe, -rho / rhoe


def primitive_to_conservative_3d(rho, u, v, w, p):
    """euler_variables.jl:15-27 (3D method)."""
    return rho, rho * u, rho * v, rho * w, p / (GAMMA - 1) + .5 * rho * (u ** 2 + v ** 2 + w ** 2)


def v_ufun_3d(rho, rhou, rhov, rhow, E):
    """Entropy variables, euler_variables.jl:79-92 (3D method)."""
    rhoe = E - .5 * (rhou ** 2 + rhov ** 2 + rhow ** 2) / rho
    s = np.log((GAMMA - 1) * rhoe / rho ** GAMMA)
    return (-E + rhoe * (GAMMA + 1 - s)) / rhoe, rhou / rhoe, rhov / rhoe, rhow / rhoe, -rho / rhoe
