"""Loader of the C-ABI shared library ``libos2r.so`` (HIP kernels for gfx950).

There is no CPU fallback: if the library has not been built (``python -c 'import
__graft_entry__ as g; g.build()'`` or ``make -C gym-os2r_amd/csrc``) importing the
device path raises.
"""
from __future__ import annotations

import ctypes as C
import os

from . import abi

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("OS2R_LIBRARY") or os.path.join(_HERE, "libos2r.so")   # OS2R_LIBRARY: A/B builds of the same ABI
_lib = None

# every symbol include/os2r.h declares
SYMBOLS = ["os2r_abi_version", "os2r_create", "os2r_destroy", "os2r_reset", "os2r_step",
           "os2r_rollout", "os2r_get_state", "os2r_set_state", "os2r_get_solver_state", "os2r_set_solver_state", "os2r_get_action_history", "os2r_set_action_history",
           "os2r_set_params", "os2r_get_params", "os2r_get_episode_info", "os2r_set_episode_info", "os2r_get_action_violations",
           "os2r_get_step_count",
           "os2r_set_step_count", "os2r_bench_steps", "os2r_bench_steps_multi", "os2r_set_work_counters", "os2r_set_done_reasons", "os2r_set_done_mask", "os2r_get_violation_mirror", "os2r_model_is_compiled_in",
           "os2r_register_model_kernels", "os2r_last_error"]


class Os2rLibraryMissing(ImportError):
    pass


def load():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise Os2rLibraryMissing(
            f"{LIB_PATH} not found: build the HIP extension first (make -C gym-os2r_amd/csrc). "
            "The stepper has no CPU fallback.")
    # torch first: the library needs libamdhip64, and the process must hold ONE HIP runtime -- the one torch brings.  Loaded before
    # torch (e.g. build() and smoke() of __graft_entry__ in one process) the library pulls in the system's runtime, torch then
    # its own, and os2r_create sees no device through the first ("no HIP device visible").
    import torch  # noqa: F401
    lib = C.CDLL(LIB_PATH)
    vp, u8p = C.c_void_p, C.c_void_p
    lib.os2r_abi_version.restype = C.c_int
    lib.os2r_create.argtypes = [C.POINTER(abi.Os2rConfig), C.POINTER(vp)]
    lib.os2r_destroy.argtypes = [vp]
    lib.os2r_reset.argtypes = [vp, u8p, vp, vp]
    lib.os2r_step.argtypes = [vp, vp, vp, vp, u8p, vp, vp]
    lib.os2r_rollout.argtypes = [vp, C.c_int, vp, vp, vp, u8p, vp, vp, vp]
    lib.os2r_get_solver_state.argtypes = [vp, vp, vp, vp]
    lib.os2r_set_solver_state.argtypes = [vp, vp, vp, vp]
    lib.os2r_get_state.argtypes = [vp, vp, vp, vp]
    lib.os2r_set_state.argtypes = [vp, vp, vp, vp]
    lib.os2r_get_action_history.argtypes = [vp, C.c_int, vp, vp]
    lib.os2r_set_action_history.argtypes = [vp, C.c_int, vp, vp]
    lib.os2r_set_params.argtypes = [vp, C.c_int, vp, vp]
    lib.os2r_get_params.argtypes = [vp, C.c_int, vp, vp]
    lib.os2r_get_episode_info.argtypes = [vp, vp, vp, vp, vp]
    lib.os2r_set_episode_info.argtypes = [vp, vp, vp, vp, vp]
    lib.os2r_get_action_violations.argtypes = [vp, vp, C.c_int32, vp]
    lib.os2r_get_step_count.argtypes = [vp, C.POINTER(C.c_uint64)]
    lib.os2r_set_step_count.argtypes = [vp, C.c_uint64]
    lib.os2r_bench_steps.argtypes = [vp, C.c_int, vp, C.POINTER(C.c_float)]
    lib.os2r_bench_steps_multi.argtypes = [C.POINTER(vp), C.POINTER(vp), C.c_int, C.c_int]
    lib.os2r_set_work_counters.argtypes = [vp, vp]
    lib.os2r_set_done_reasons.argtypes = [vp, vp]
    lib.os2r_set_done_mask.argtypes = [vp, u8p]
    lib.os2r_get_violation_mirror.argtypes = [vp, C.POINTER(vp)]
    lib.os2r_model_is_compiled_in.argtypes = [C.POINTER(abi.Os2rModel)]
    lib.os2r_register_model_kernels.argtypes = [C.POINTER(abi.Os2rModel), C.c_int32, C.c_int32, C.c_char_p]
    lib.os2r_last_error.argtypes = [vp]
    lib.os2r_last_error.restype = C.c_char_p
    for name in SYMBOLS:
        getattr(lib, name)  # AttributeError here means header and library disagree
        if name != "os2r_last_error":
            getattr(lib, name).restype = C.c_int
    if lib.os2r_abi_version() != abi.ABI_VERSION:
        raise ImportError("libos2r.so ABI version does not match gym_os2r_amd.abi")
    _lib = lib
    return lib
