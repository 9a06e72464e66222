"""esdg_cns_amd -- MI355X-native RHS engine for the entropy-stable DG Euler / compressible
Navier-Stokes solvers of yiminllin/ESDG-CNS (hot path only)."""
from . import physics, setup_dg  # noqa: F401

__all__ = ["setup_dg", "physics", "engine", "build"]
