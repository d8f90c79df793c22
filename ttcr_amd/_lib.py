"""ctypes binding of the C ABI declared in include/ttcr_amd.h (libttcr_amd.so).

This is the stub a ttcrpy maintainer would mirror in Cython.  The library
is the product: if it is missing or has no HIP device, calls fail loudly -- there is no CPU
or oracle fallback anywhere in this package.
"""
import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("TTCR_AMD_LIB") or os.path.join(HERE, "libttcr_amd.so")  # override: tuning builds only

TTCR_F32, TTCR_F64 = 0, 1
OK, ERR_VALUE, ERR_RUNTIME, ERR_DEVICE, ERR_UNSUPPORTED = 0, 1, 2, 3, 4

# every symbol include/ttcr_amd.h declares: name -> (restype, argtypes)
_P = C.c_void_p
_U32 = C.c_uint32
_D = C.c_double
_I = C.c_int


class Timing(C.Structure):
    _fields_ = [("sweep_ms", C.c_double), ("total_ms", C.c_double), ("kernel_launches", C.c_longlong),
                ("node_updates", C.c_longlong), ("evaluated_updates", C.c_longlong), ("iterations", C.c_int),
                ("n_sources", C.c_int)]


SYMBOLS = {
    "ttcr_fsm_device_count": (_I, []),
    "ttcr_fsm_last_error": (C.c_char_p, []),
    "ttcr_fsm3d_create": (_I, [C.POINTER(_P), _I, _I, _U32, _U32, _U32, _D, _D, _D, _D, _D, _I, _I, _I, _I, _I]),
    "ttcr_fsm2d_create": (_I, [C.POINTER(_P), _I, _I, _U32, _U32, _D, _D, _D, _D, _D, _I, _I, _I, _I, _I]),
    "ttcr_fsm3d_create_multi": (_I, [C.POINTER(_P), _I, _I, _U32, _U32, _U32, _D, _D, _D, _D, _D, _I, _I, _I, _I, C.POINTER(_I), _I]),
    "ttcr_fsm2d_create_multi": (_I, [C.POINTER(_P), _I, _I, _U32, _U32, _D, _D, _D, _D, _D, _I, _I, _I, _I, C.POINTER(_I), _I]),
    "ttcr_fsm_n_devices": (_I, [_P]),
    "ttcr_fsm_destroy": (None, [_P]),
    "ttcr_fsm_set_slowness": (_I, [_P, _P, C.c_size_t]),
    "ttcr_fsm_set_slowness_device": (_I, [_P, _P, C.c_size_t]),
    "ttcr_fsm_set_slowness_c_order": (_I, [_P, _P, C.c_size_t]),
    "ttcr_fsm_get_slowness": (_I, [_P, _P, C.c_size_t]),
    "ttcr_fsm_raytrace": (_I, [_P, _I, _I, _P, _P, _I, _P, _P]),
    "ttcr_fsm_raytrace_multi": (_I, [_P, _I, _P, _P, _P, _P, _P, _P]),
    "ttcr_fsm_get_tt": (_I, [_P, _I, _P, C.c_size_t]),
    "ttcr_fsm_get_tt_device": (_I, [_P, _I, C.POINTER(_P)]),
    "ttcr_fsm_get_tt_device_view": (_I, [_P, _I, C.POINTER(_P), C.POINTER(C.c_size_t)]),
    "ttcr_fsm_interp": (_I, [_P, _I, _I, _P, _P]),
    "ttcr_fsm_compute_slowness": (_I, [_P, _I, _P, _I, _P]),
    "ttcr_fsm_get_niter": (_I, [_P, _I, C.POINTER(_I), C.POINTER(_I)]),
    "ttcr_fsm_get_changes": (_I, [_P, _I, C.POINTER(_D), _I, C.POINTER(_D), _I]),
    "ttcr_fsm_get_reference_changes": (_I, [_P, _I, C.POINTER(_D), _I, C.POINTER(_D), _I]),
    "ttcr_fsm_n_slots": (_I, [_P]),
    "ttcr_fsm_n_nodes": (C.c_size_t, [_P]),
    "ttcr_fsm_n_cells": (C.c_size_t, [_P]),
    "ttcr_fsm_set_option": (_I, [_P, C.c_char_p, _D]),
    "ttcr_fsm_last_timing": (_I, [_P, C.POINTER(Timing)]),
    "ttcr_fsm_last_kernel": (_I, [_P, C.c_char_p, C.c_size_t]),
    "ttcr_fsm_build_id": (C.c_char_p, []),
    "ttcr_fsm_stopping_stats": (_I, [_P, C.POINTER(C.c_longlong), C.POINTER(C.c_longlong), C.POINTER(C.c_longlong)]),
    "ttcr_fsm_prefill_swaps": (_I, [_P, C.POINTER(C.c_longlong)]),
    "ttcr_fsm_reference_change": (_I, [_P, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]),
    "ttcr_fsm_rays_size": (_I, [_P, C.POINTER(C.c_size_t), C.POINTER(C.c_size_t)]),
    "ttcr_fsm_get_rays": (_I, [_P, _P, _P]),
    "ttcr_fsm_raytrace_rays": (_I, [_P, _I, _I, _P, _P, _I, _P, _P]),
    "ttcr_fsm_slot_rays_size": (_I, [_P, _I, C.POINTER(C.c_size_t), C.POINTER(C.c_size_t)]),
    "ttcr_fsm_get_slot_rays": (_I, [_P, _I, _P, _P]),
    "ttcr_fsm_raytrace_m": (_I, [_P, _I, _I, _P, _P, _I, _P, _P]),
    "ttcr_fsm_raytrace_rm": (_I, [_P, _I, _I, _P, _P, _I, _P, _P]),
    "ttcr_fsm_raytrace_multi_m": (_I, [_P, _I, _P, _P, _P, _P, _P, _P, _I]),
    "ttcr_fsm_multi_m_size": (_I, [_P, C.POINTER(C.c_size_t), C.POINTER(C.c_size_t)]),
    "ttcr_fsm_get_multi_m": (_I, [_P, _P, _P, _P]),
    "ttcr_fsm_raytrace_multi_l": (_I, [_P, _I, _P, _P, _P, _P, _P, _P, _I]),
    "ttcr_fsm_multi_l_size": (_I, [_P, C.POINTER(C.c_size_t), C.POINTER(C.c_size_t)]),
    "ttcr_fsm_get_multi_l": (_I, [_P, _P, _P, _P]),
    "ttcr_fsm_slot_m_size": (_I, [_P, _I, C.POINTER(C.c_size_t), C.POINTER(C.c_size_t)]),
    "ttcr_fsm_get_slot_m": (_I, [_P, _I, _P, _P, _P]),
    "ttcr_fsm_raytrace_l": (_I, [_P, _I, _I, _P, _P, _I, _P, _P, _I]),
    "ttcr_fsm_slot_l_size": (_I, [_P, _I, C.POINTER(C.c_size_t), C.POINTER(C.c_size_t)]),
    "ttcr_fsm_get_slot_l": (_I, [_P, _I, _P, _P, _P]),
}

_lib = None


def load():
    """Load libttcr_amd.so (built in-tree by ttcr_amd.build / __graft_entry__.build)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(
                f"{LIB_PATH} is missing: build the HIP extension first (python -m ttcr_amd.build). "
                "ttcr_amd has no CPU fallback.")
        lib = C.CDLL(LIB_PATH)
        for name, (res, args) in SYMBOLS.items():
            fn = getattr(lib, name)  # AttributeError if the ABI is incomplete
            fn.restype = res
            fn.argtypes = args
        # build provenance: the library in the tree must have been compiled from the sources in the tree (a tuning build named by
        # TTCR_AMD_LIB is exempt: it is never what the tests, smoke() or bench.py measure without saying so)
        if not os.environ.get("TTCR_AMD_LIB"):
            from . import build as _b

            have, want = lib.ttcr_fsm_build_id().decode(), _b.source_hash()
            if have != want:
                raise ImportError(f"{LIB_PATH} was built from other sources (build id {have}, the sources here hash to {want}): "
                                  "rebuild it (python -m ttcr_amd.build)")
        _lib = lib
    return _lib


def build_id():
    """16-hex-digit hash of the kernel sources + compiler flags the loaded library was compiled from (ttcr_fsm_build_id)."""
    return load().ttcr_fsm_build_id().decode()


def last_error():
    return load().ttcr_fsm_last_error().decode(errors="replace")


class UnsupportedError(NotImplementedError):
    pass


class DeviceError(RuntimeError):
    pass


def check(status):
    """Map a C status to the exception the reference's Python layer would raise."""
    if status == OK:
        return
    msg = last_error()
    if status == ERR_VALUE:
        raise ValueError(msg)
    if status == ERR_UNSUPPORTED:
        raise UnsupportedError(msg)
    if status == ERR_DEVICE:
        raise DeviceError(msg)
    raise RuntimeError(msg)  # std::runtime_error / length_error / logic_error via Cython `except +`
