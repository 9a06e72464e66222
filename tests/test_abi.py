"""The C-ABI shared library loads and exports every symbol include/esdg_hip.h declares (no compute
calls here: there is no GPU in the CPU test environment and the library has no CPU path)."""
import ctypes as C
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_functions():
    src = open(os.path.join(ROOT, "include", "esdg_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(esdg_[A-Za-z0-9_]+)\s*\(", src)))


def test_header_symbols_are_exported():
    from esdg_cns_amd import _lib, build
    build.build()
    L = C.CDLL(_lib.LIB_PATH)
    names = _declared_functions()
    assert len(names) >= 30
    for n in names:
        assert hasattr(L, n), f"{n} declared in include/esdg_hip.h but not exported by libesdg_hip.so"
    # the Python binding covers exactly the header
    assert sorted(_lib.SYMBOLS) == names


def test_no_cpu_fallback():
    """Without a GPU the product must fail loudly, not fall back to the oracle or any CPU path."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    import numpy as np
    from esdg_cns_amd import _lib, engine
    from common import product_euler_problem
    rd, md, ops, Q = product_euler_problem(2, 3, 3)
    with pytest.raises(_lib.EsdgError):
        engine.RhsEngine(rd, md, ops, engine.EULER_COLLOCATED)
    L = _lib.lib()
    assert L.esdg_device_count() == 0
    # product sources never import the oracle
    for dirpath, _, files in os.walk(os.path.join(ROOT, "esdg_cns_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".hpp")):
                txt = open(os.path.join(dirpath, f)).read()
                assert "import oracle" not in txt and "from oracle" not in txt and "liboracle" not in txt, f


def test_error_codes_and_messages():
    from esdg_cns_amd import _lib
    L = _lib.lib()
    assert L.esdg_version().startswith(b"esdg_hip")
    plan = C.c_void_p()
    rc = L.esdg_halo_plan_create(None, 4, 8, 0, 4, 1, None, C.byref(plan))
    assert rc == -1 and b"bad halo plan" in L.esdg_last_error()
