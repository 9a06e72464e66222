"""ctypes binding of libesdg_hip.so (C ABI: include/esdg_hip.h).  There is no fallback: if the HIP
extension is missing or no GPU is present, the product path raises."""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# ESDG_HIP_LIB: another build of the same library (A/B variants under esdg_cns_amd/variants/, tools/); default in-tree
LIB_PATH = os.environ.get("ESDG_HIP_LIB") or os.path.join(_HERE, "libesdg_hip.so")

c_double_p = C.POINTER(C.c_double)
c_int64_p = C.POINTER(C.c_int64)
c_uint8_p = C.POINTER(C.c_uint8)


class esdg_ops_t(C.Structure):
    _fields_ = [("N", C.c_int32), ("Np", C.c_int32), ("Nq", C.c_int32), ("Nfq", C.c_int32)] + [
        (n, c_double_p) for n in ("Qrhskew", "Qshskew", "Ph", "wq", "wf", "Ef", "Lf", "Vq", "Pq", "VhP", "LIFT", "Vf", "Dr", "Ds")]


class esdg_mesh_t(C.Structure):
    _fields_ = [("K", C.c_int64), ("geo_ld", C.c_int32),
                ("rxJ", c_double_p), ("sxJ", c_double_p), ("ryJ", c_double_p), ("syJ", c_double_p),
                ("J", c_double_p), ("wJq", c_double_p), ("nxJ", c_double_p), ("nyJ", c_double_p), ("sJ", c_double_p),
                ("mapP", c_int64_p), ("mapB", c_int64_p), ("NmapB", C.c_int64), ("bkind", c_uint8_p),
                ("elem_offset", C.c_int64), ("Kglobal", C.c_int64), ("nranks", C.c_int32), ("rank", C.c_int32),
                ("rank_offsets", c_int64_p), ("vlid", c_double_p)]


class esdg_err_ops_t(C.Structure):
    _fields_ = [("Nq2", C.c_int32)] + [(n, c_double_p) for n in ("Vq2", "wq2", "x", "y", "J", "Vf", "wf")]


class esdg_hex_ops_t(C.Structure):
    _fields_ = [("N", C.c_int32), ("Nq", C.c_int32), ("Nfq", C.c_int32)] + [
        (n, c_double_p) for n in ("Qrhskew", "Qshskew", "Qthskew", "Ph", "Lf", "Ef", "wq", "wf")]


class esdg_hex_mesh_t(C.Structure):
    _fields_ = ([("K", C.c_int64), ("geo_ld", C.c_int32)]
                + [(n, c_double_p) for n in ("rxJ", "sxJ", "txJ", "ryJ", "syJ", "tyJ", "rzJ", "szJ", "tzJ", "J", "wJq",
                                             "nxJ", "nyJ", "nzJ", "sJ")]
                + [("mapP", c_int64_p), ("elem_offset", C.c_int64), ("Kglobal", C.c_int64), ("nranks", C.c_int32),
                   ("rank", C.c_int32), ("rank_offsets", c_int64_p)])


class esdg_phys_t(C.Structure):
    _fields_ = [("formulation", C.c_int32), ("lf_scale", C.c_double), ("inviscid_dissp", C.c_int32),
                ("viscous_dissp", C.c_int32), ("BCTYPE", C.c_int32), ("Re", C.c_double), ("mu", C.c_double),
                ("lambda_", C.c_double), ("Pr", C.c_double),
                ("inflow_rho", C.c_double), ("inflow_u", C.c_double), ("inflow_v", C.c_double), ("inflow_p", C.c_double)]


# every symbol include/esdg_hip.h declares: (restype, argtypes)
_vp = C.c_void_p
_szp = C.POINTER(C.c_size_t)
_i32p = C.POINTER(C.c_int32)
SYMBOLS = {
    "esdg_create": (C.c_int, [C.POINTER(esdg_ops_t), C.POINTER(esdg_mesh_t), C.POINTER(esdg_phys_t), C.POINTER(_vp)]),
    "esdg_create_hex": (C.c_int, [C.POINTER(esdg_hex_ops_t), C.POINTER(esdg_hex_mesh_t), C.POINTER(esdg_phys_t), C.POINTER(_vp)]),
    "esdg_num_fields": (C.c_int, [_vp]),
    "esdg_destroy": (C.c_int, [_vp]),
    "esdg_last_error": (C.c_char_p, []),
    "esdg_version": (C.c_char_p, []),
    "esdg_workspace_bytes": (C.c_size_t, [_vp]),
    "esdg_bind_workspace": (C.c_int, [_vp, _vp, C.c_size_t]),
    "esdg_num_phases": (C.c_int, [_vp]),
    "esdg_uses_tensor_kernels": (C.c_int, [_vp]),
    "esdg_rhs_phase": (C.c_int, [_vp, C.c_int, _vp, _vp, _vp]),
    "esdg_rhs": (C.c_int, [_vp, _vp, _vp, _vp]),
    "esdg_rhs_lsrk": (C.c_int, [_vp, _vp, _vp, C.c_double, C.c_double, C.c_double, _vp]),
    "esdg_rhs_phase_lsrk": (C.c_int, [_vp, C.c_int, _vp, _vp, C.c_double, C.c_double, C.c_double, _vp]),
    "esdg_rhstest": (C.c_int, [_vp, _vp, _vp, c_double_p, _vp]),
    "esdg_check_state": (C.c_int, [_vp, _vp, c_double_p, _vp]),
    "esdg_abi_sizeof": (C.c_int64, [C.c_char_p]),
    "esdg_error_setup": (C.c_int, [_vp, C.POINTER(esdg_err_ops_t)]),
    "esdg_error_l2": (C.c_int, [_vp, _vp, C.c_int32, c_double_p, C.c_double, c_double_p, _vp]),
    "esdg_error_nodal": (C.c_int, [_vp, _vp, C.c_int32, c_double_p, C.c_double, c_double_p, _vp]),
    "esdg_error_boundary_velocity": (C.c_int, [_vp, _vp, C.c_double, c_double_p, _vp]),
    "esdg_set_parts": (C.c_int, [_vp, C.c_int]),
    "esdg_viscous_entropy_test": (C.c_int, [_vp, _vp, c_double_p, _vp]),
    "esdg_rhs_host": (C.c_int, [_vp, C.POINTER(c_double_p), C.POINTER(c_double_p)]),
    "esdg_halo_num_neighbors": (C.c_int, [_vp]),
    "esdg_num_exchanges": (C.c_int, [_vp]),
    "esdg_exchange_info": (C.c_int, [_vp, C.c_int, _i32p, _i32p, _i32p]),
    "esdg_halo_segment": (C.c_int, [_vp, C.c_int, C.c_int, _i32p, _szp, _szp, _szp, _szp]),
    "esdg_interior_range": (C.c_int, [_vp, c_int64_p, c_int64_p]),
    "esdg_rhs_phase_range": (C.c_int, [_vp, C.c_int, C.c_int64, C.c_int64, _vp, _vp, _vp]),
    "esdg_rhs_phase_range_lsrk": (C.c_int, [_vp, C.c_int, C.c_int64, C.c_int64, _vp, _vp, C.c_double, C.c_double, C.c_double, _vp]),
    "esdg_halo_pack": (C.c_int, [_vp, C.c_int, _vp]),
    "esdg_comm_unique_id": (C.c_int, [_vp]),
    "esdg_comm_init": (C.c_int, [_vp, _vp, C.c_int, C.c_int]),
    "esdg_comm_set_loopback": (C.c_int, [_vp, C.c_int]),
    "esdg_comm_size": (C.c_int, [_vp]),
    "esdg_comm_destroy": (C.c_int, [_vp]),
    "esdg_halo_exchange": (C.c_int, [_vp, C.c_int, _vp]),
    "esdg_halo_wait": (C.c_int, [_vp, C.c_int, _vp]),
    "esdg_comm_allreduce": (C.c_int, [_vp, c_double_p, C.c_int, C.c_int, _vp]),
    "esdg_halo_plan_create": (C.c_int, [c_int64_p, C.c_int64, C.c_int32, C.c_int64, C.c_int64, C.c_int32, c_int64_p, C.POINTER(_vp)]),
    "esdg_halo_plan_destroy": (C.c_int, [_vp]),
    "esdg_halo_plan_num_neighbors": (C.c_int, [_vp]),
    "esdg_halo_plan_num_ghosts": (C.c_int64, [_vp]),
    "esdg_halo_plan_num_sends": (C.c_int64, [_vp]),
    "esdg_halo_plan_neighbor": (C.c_int, [_vp, C.c_int, _i32p, c_int64_p, c_int64_p, c_int64_p, c_int64_p]),
    "esdg_halo_plan_mapP": (_i32p, [_vp]),
    "esdg_halo_plan_sendlist": (_i32p, [_vp]),
    "esdg_lsrk_update": (C.c_int, [_vp, _vp, _vp, C.c_double, C.c_double, C.c_double, C.c_int64, _vp]),
    "esdg_axpy_stages": (C.c_int, [_vp, _vp, C.POINTER(_vp), c_double_p, C.c_int, C.c_double, C.c_int64, _vp]),
    "esdg_dopri_error": (C.c_int, [_vp, C.POINTER(_vp), c_double_p, C.c_int, C.c_double, C.c_int64, c_double_p, _vp]),
    "esdg_dopri_error_fields": (C.c_int, [_vp, C.POINTER(_vp), c_double_p, C.c_int, C.c_double, C.c_int64, C.c_int, c_double_p, _vp]),
    "esdg_lsrk45_step": (C.c_int, [_vp, _vp, _vp, C.c_double, _vp]),
    "esdg_dopri45_attempt": (C.c_int, [_vp, _vp, _vp, C.POINTER(_vp), C.c_double, C.c_double, c_double_p, _vp]),
    "esdg_dopri45_next_dt": (C.c_double, [C.c_double, C.c_double, C.c_double, C.c_double, C.c_int64]),
    "esdg_setup_uniform_quad_mesh": (C.c_int, [C.c_int, C.c_int, c_double_p, c_double_p, c_int64_p]),
    "esdg_setup_quad": (C.c_int, [C.c_int, C.c_int, c_double_p, c_double_p, C.c_int64, c_int64_p, C.c_int64, C.c_int, C.c_int64,
                                  C.c_int64, C.POINTER(_vp)]),
    "esdg_setup_array": (c_double_p, [_vp, C.c_char_p, c_int64_p, c_int64_p]),
    "esdg_setup_map": (c_int64_p, [_vp, C.c_char_p, c_int64_p]),
    "esdg_setup_fill": (C.c_int, [_vp, C.POINTER(esdg_ops_t), C.POINTER(esdg_mesh_t)]),
    "esdg_setup_uniform_hex_mesh": (C.c_int, [C.c_int, C.c_int, C.c_int, c_double_p, c_double_p, c_double_p, c_int64_p]),
    "esdg_setup_hex": (C.c_int, [C.c_int, c_double_p, c_double_p, c_double_p, C.c_int64, c_int64_p, C.c_int64, C.c_int, C.c_int64,
                                 C.c_int64, C.POINTER(_vp)]),
    "esdg_setup_fill_hex": (C.c_int, [_vp, C.POINTER(esdg_hex_ops_t), C.POINTER(esdg_hex_mesh_t)]),
    "esdg_setup_destroy": (C.c_int, [_vp]),
    "esdg_setup_last_error": (C.c_char_p, []),
    "esdg_dmalloc": (_vp, [C.c_size_t]),
    "esdg_dfree": (C.c_int, [_vp]),
    "esdg_memcpy_h2d": (C.c_int, [_vp, _vp, C.c_size_t]),
    "esdg_memcpy_d2h": (C.c_int, [_vp, _vp, C.c_size_t]),
    "esdg_device_synchronize": (C.c_int, []),
    "esdg_device_count": (C.c_int, []),
    "esdg_set_device": (C.c_int, [C.c_int]),
}

_LIB = None


class EsdgError(RuntimeError):
    pass


AB_LIB_PATH = os.path.join(_HERE, "libesdg_hip_ab.so")   # the same kernels, environment switches compiled in (build.py)
_LIBS = {}


def lib(ab=False):
    """Load libesdg_hip.so (ab=True: the A/B build libesdg_hip_ab.so, whose esdg_create reads the ESDG_* environment
    switches -- the shipped library reads none); raise loudly when the HIP extension has not been built."""
    global _LIB
    path = AB_LIB_PATH if (ab and not os.environ.get("ESDG_HIP_LIB")) else LIB_PATH
    if path not in _LIBS:
        if not os.path.exists(path):
            raise EsdgError(f"{path} not found: the HIP extension is not built "
                            "(python -c 'import __graft_entry__ as g; g.build()'). There is no CPU fallback.")
        L = C.CDLL(path)
        for name, (res, args) in SYMBOLS.items():
            fn = getattr(L, name)      # AttributeError if the ABI and the header drifted apart
            fn.restype = res
            fn.argtypes = args
        for st in (esdg_ops_t, esdg_mesh_t, esdg_phys_t, esdg_hex_ops_t, esdg_hex_mesh_t, esdg_err_ops_t):
            want = L.esdg_abi_sizeof(st.__name__.encode())
            if want != C.sizeof(st):
                raise EsdgError(f"ABI drift: {st.__name__} is {want} bytes in {os.path.basename(path)}, {C.sizeof(st)} in _lib.py")
        _LIBS[path] = L
        if path == LIB_PATH:
            _LIB = L
    return _LIBS[path]


def check_on(L, rc):
    """Raise with the message of the library the failing call went to (two builds may be loaded side by side)."""
    if rc != 0:
        raise EsdgError(f"libesdg_hip error {rc}: {L.esdg_last_error().decode()}")


def check(rc):
    check_on(lib(), rc)
