"""ctypes binding of libegoprior_hip.so (the C ABI declared in include/egoprior.h).

There is no CPU fallback: if the HIP library has not been built, importing this module's `lib()` raises.
"""
from __future__ import annotations

import ctypes as C
import subprocess
from pathlib import Path

_PKG = Path(__file__).resolve().parent
CSRC = _PKG.parent / "csrc"
import os

# EGOPRIOR_LIB: file name of another build of the library in csrc/ (the diagnostic build of the profiling tools)
LIB_PATH = CSRC / os.environ.get("EGOPRIOR_LIB", "libegoprior_hip.so")

MAX_K = 20
NOALLOC_MAX_SAMPLES = 8 * 1024 * 1024 + 256 * 1024
TILE = 8

E_OK, E_INVALID, E_WORKSPACE, E_HIP, E_UNSUPPORTED = 0, -1, -2, -3, -4
SELECT_STATIC, SELECT_DYNAMIC, SELECT_ALL = 0, 1, 2
CAM_PINHOLE, CAM_FISHEYE = 0, 1
GREY_PIL_L, GREY_FLOAT = 0, 1
ABI_VERSION = 2


class UnprojectParams(C.Structure):
    _fields_ = [("H", C.c_int32), ("W", C.c_int32), ("subsample", C.c_int32), ("select", C.c_int32),
                ("fx", C.c_float), ("fy", C.c_float), ("cx", C.c_float), ("cy", C.c_float),
                ("c2w", C.c_float * 12), ("T", C.c_double * 12)]


class Camera(C.Structure):
    _fields_ = [("model", C.c_int32), ("H", C.c_int32), ("W", C.c_int32),
                ("R", C.c_float * 9), ("T", C.c_float * 3),
                ("fx", C.c_float), ("fy", C.c_float), ("px", C.c_float), ("py", C.c_float),
                ("radial", C.c_float * 6), ("tangential", C.c_float * 2), ("thin_prism", C.c_float * 4)]


class RasterParams(C.Structure):
    _fields_ = [("radius", C.c_float), ("radius_sq_w", C.c_float), ("K", C.c_int32), ("rotate_cw", C.c_int32),
                ("colors_gt1", C.c_int32)]


class EgoPriorError(RuntimeError):
    def __init__(self, code: int, msg: str):
        super().__init__(f"egoprior error {code}: {msg}")
        self.code = code


_vp, _i32, _i64, _f32, _sz = C.c_void_p, C.c_int32, C.c_int64, C.c_float, C.c_size_t

# name -> (restype, argtypes); every symbol include/egoprior.h declares
SIGNATURES = {
    "egoprior_last_error": (C.c_char_p, []),
    "egoprior_abi_version": (C.c_int, []),
    "egoprior_reliable_depth_mask": (C.c_int, [_vp, _i32, _i32, _vp, _vp]),
    "egoprior_reliable_depth_mask_f16": (C.c_int, [_vp, _i32, _i32, _vp, _vp]),
    "egoprior_unproject_workspace_bytes": (_sz, [_i32, _i32, _i32]),
    "egoprior_unproject_frame": (C.c_int, [C.POINTER(UnprojectParams), _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _sz,
                                           _vp]),
    "egoprior_unproject_frame_f16": (C.c_int, [C.POINTER(UnprojectParams), _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp,
                                               _sz, _vp]),
    "egoprior_unproject_frame_noalloc": (C.c_int, [C.POINTER(UnprojectParams), _vp, _vp, _vp, _vp, _vp, _vp, _vp,
                                                   _vp]),
    "egoprior_unproject_frame_rays": (C.c_int, [C.POINTER(UnprojectParams), _vp, _i32, _vp, _vp, _vp, _vp, _vp, _vp,
                                                _vp, _vp, _sz, _vp]),
    "egoprior_meanbg_state_bytes": (_sz, [_i32, _i32]),
    "egoprior_meanbg_accumulate": (C.c_int, [_vp, _vp, _vp, _vp, _i32, _i64, _i32, _i32, _vp]),
    "egoprior_meanbg_accumulate_f16": (C.c_int, [_vp, _vp, _vp, _vp, _i32, _i64, _i32, _i32, _vp]),
    "egoprior_meanbg_finalize": (C.c_int, [_vp, _i32, _i32, _vp, _vp, _vp]),
    "egoprior_meanbg_finalize_f32": (C.c_int, [_vp, _i32, _i32, _i32, _vp, _vp, _vp]),
    "egoprior_meanbg_export": (C.c_int, [_vp, _i32, _i32, _vp, _vp, _vp]),
    "egoprior_meanbg_import": (C.c_int, [_vp, _i32, _i32, _vp, _vp, _vp]),
    "egoprior_filter_workspace_bytes": (_sz, [_i64]),
    "egoprior_filter_cloud": (C.c_int, [_vp, _vp, _i64, _vp, _vp, _vp, _vp, _sz, _vp]),
    "egoprior_project_points": (C.c_int, [C.POINTER(Camera), _vp, _i64, _vp, _vp, _i64, _vp, _vp, _vp]),
    "egoprior_raster_workspace_bytes": (_sz, [_i32, _i32, _i64, _f32]),
    "egoprior_rasterize_composite": (C.c_int, [C.POINTER(RasterParams), _i32, _i32, _vp, _i64, _vp, _vp, _vp, _vp, _vp,
                                               _vp, _vp, _vp, _vp, _vp, _sz, _vp]),
    "egoprior_render_workspace_bytes": (_sz, [_i32, _i32, _i64, _f32]),
    "egoprior_raster_workspace_bytes_entries": (_sz, [_i32, _i32, _i64, _i64]),
    "egoprior_render_workspace_bytes_entries": (_sz, [_i32, _i32, _i64, _i64]),
    "egoprior_count_list_entries": (C.c_int, [C.POINTER(Camera), C.POINTER(RasterParams), _i32, _i32, _vp, _i64, _vp,
                                              _vp, _i64, _vp, _vp, _vp, _sz, _vp]),
    "egoprior_render_frame": (C.c_int, [C.POINTER(Camera), C.POINTER(RasterParams), _vp, _i64, _vp, _vp, _vp, _i64,
                                        _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _sz, _vp]),
    "egoprior_white_pixel_counts": (C.c_int, [_vp, _i32, _i32, _i32, _f32, _i32, _vp, _vp, _vp]),
    "egoprior_postprocess_background": (C.c_int, [_vp, _vp, _i32, _i32, _i32, _vp, _vp]),
    "egoprior_set_option": (C.c_int, [C.c_char_p, _i32]),
    "egoprior_get_option": (_i32, [C.c_char_p]),
    "egoprior_profile_begin": (C.c_int, []),
    "egoprior_profile_end": (C.c_int, [C.c_char_p, _sz, _vp, _vp, _i32, _vp]),
}

_LIB = None


def build(force: bool = False) -> Path:
    """Compile the HIP library in-tree for gfx950 (hipcc cross-compiles without a GPU)."""
    targets = ["libegoprior_hip.so", "libegoprior_codec.so"]  # not the optional torch shim (`make ext`)
    subprocess.check_call(["make", "-C", str(CSRC), "-s"] + (["-B"] if force else []) + targets)
    return LIB_PATH


def lib():
    global _LIB
    if _LIB is None:
        if not LIB_PATH.exists():
            raise ImportError(f"{LIB_PATH} is missing: build it with `make -C {CSRC}` "
                              "(there is no CPU fallback for the ego-prior renderer)")
        # torch first: the extension must bind to the HIP runtime torch ships and initialises, not to a second copy
        # (with two runtimes in one process this library's launches do not see torch's device)
        import torch  # noqa: F401
        L = C.CDLL(str(LIB_PATH))
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(L, name)  # AttributeError if the library does not export a declared symbol
            fn.restype = res
            fn.argtypes = args
        if L.egoprior_abi_version() != ABI_VERSION:
            raise ImportError("libegoprior_hip.so ABI version mismatch")
        _LIB = L
        # EGOPRIOR_OPTIONS="name=value,name=value": tuning switches (egoprior_set_option) for programs that have no flag
        # for them, applied once when the library is loaded
        for item in filter(None, os.environ.get("EGOPRIOR_OPTIONS", "").split(",")):
            name, _, value = item.partition("=")
            if L.egoprior_set_option(name.strip().encode(), int(value)) != 0:
                raise ValueError(f"EGOPRIOR_OPTIONS: unknown option {name.strip()!r}")
    return _LIB


def check(code: int) -> None:
    if code != 0:
        raise EgoPriorError(code, lib().egoprior_last_error().decode("utf-8", "replace"))


def profile_begin() -> None:
    check(lib().egoprior_profile_begin())


def profile_end() -> dict:
    """{kernel name: (total milliseconds, launches)} since profile_begin()."""
    names = C.create_string_buffer(4096)
    ms = (C.c_float * 64)()
    calls = (C.c_int32 * 64)()
    n = C.c_int32(0)
    check(lib().egoprior_profile_end(names, 4096, C.cast(ms, C.c_void_p), C.cast(calls, C.c_void_p), 64,
                                     C.cast(C.byref(n), C.c_void_p)))
    out = {}
    for i, nm in enumerate(names.value.decode().split("\n")[:n.value]):
        if nm:
            out[nm] = (float(ms[i]), int(calls[i]))
    return out


def set_option(name: str, value: int) -> None:
    check(lib().egoprior_set_option(name.encode(), int(value)))


def get_option(name: str) -> int:
    return int(lib().egoprior_get_option(name.encode()))
