"""ctypes binding of libdnp.so (include/dnp.h).  Plumbing only: device memory and streams are
torch's, arithmetic is the library's.  There is no CPU fallback: if the library is missing or
no HIP device is visible, the product path raises."""
import ctypes
import os
import threading

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("DNP_LIB", os.path.join(_HERE, "libdnp.so"))

ABI_VERSION = 503          # DNP_VERSION of include/dnp.h this binding matches (0.5.0: fp64 patch-driver entry points)

_c_i64 = ctypes.c_int64
_c_p = ctypes.c_void_p
_c_sz = ctypes.c_size_t

# name -> (restype, argtypes); must list every symbol include/dnp.h declares
SIGNATURES = {
    "dnp_version": (ctypes.c_int, []),
    "dnp_device_count": (ctypes.c_int, []),
    "dnp_last_error": (ctypes.c_char_p, []),
    "dnp_field_grad_workspace_bytes": (_c_sz, [_c_i64, _c_i64, _c_i64]),
    "dnp_potential_workspace_bytes": (_c_sz, [_c_i64, _c_i64, _c_i64]),
    "dnp_field_grad_f32": (ctypes.c_int, [_c_p, _c_i64, _c_i64, _c_p, _c_p, _c_i64, _c_i64, _c_p, ctypes.c_float,
                                          _c_i64, _c_p, _c_i64, ctypes.c_int, ctypes.c_int, _c_p, _c_p, _c_p, _c_sz, _c_p]),
    "dnp_field_grad_f64": (ctypes.c_int, [_c_p, _c_i64, _c_i64, _c_p, _c_p, _c_i64, _c_i64, _c_p, ctypes.c_double,
                                          _c_i64, _c_p, _c_i64, ctypes.c_int, ctypes.c_int, _c_p, _c_p, _c_p, _c_sz, _c_p]),
    "dnp_potential_f32": (ctypes.c_int, [_c_p, _c_i64, _c_i64, _c_p, _c_p, _c_i64, _c_i64, _c_p, _c_i64, _c_p,
                                         _c_i64, _c_p, _c_sz, _c_p]),
    "dnp_potential_f64": (ctypes.c_int, [_c_p, _c_i64, _c_i64, _c_p, _c_p, _c_i64, _c_i64, _c_p, _c_i64, _c_p,
                                         _c_i64, _c_p, _c_sz, _c_p]),
    "dnp_reference_field_f32": (ctypes.c_int, [_c_p, _c_i64, _c_i64, _c_p, _c_i64, _c_i64, ctypes.c_int, ctypes.c_float, _c_i64,
                                               _c_p, _c_i64, _c_p, _c_p, _c_sz, _c_p]),
    "dnp_reference_field_f64": (ctypes.c_int, [_c_p, _c_i64, _c_i64, _c_p, _c_i64, _c_i64, ctypes.c_int, ctypes.c_double, _c_i64,
                                               _c_p, _c_i64, _c_p, _c_p, _c_sz, _c_p]),
    "dnp_patch_fields_f32": (ctypes.c_int, [_c_p, _c_i64, _c_i64, _c_p, _c_p, _c_i64, _c_p, _c_i64, _c_i64,
                                            ctypes.c_float, _c_p, _c_p]),
    "dnp_patch_boxes_f32": (ctypes.c_int, [_c_p, _c_i64, _c_i64, _c_p, _c_p, _c_i64, _c_p, _c_p]),
    "dnp_patch_fields_boxed_f32": (ctypes.c_int, [_c_p, _c_i64, _c_i64, _c_p, _c_p, _c_i64, _c_p, _c_p, _c_i64, _c_i64,
                                                  ctypes.c_float, _c_p, _c_p]),
    "dnp_patch_tile_rows": (_c_i64, []),
    "dnp_tile_boxes_f32": (ctypes.c_int, [_c_p, _c_i64, _c_i64, _c_i64, _c_p, _c_p]),
    "dnp_patch_fields_tiled_f32": (ctypes.c_int, [_c_p, _c_i64, _c_i64, _c_p, _c_p, _c_i64, _c_p, _c_p, _c_p, _c_i64, _c_i64,
                                                  ctypes.c_float, _c_p, _c_p, ctypes.c_int, ctypes.c_int, _c_p, _c_sz, _c_p]),
    "dnp_patch_fields_tiled_f64": (ctypes.c_int, [_c_p, _c_i64, _c_i64, _c_p, _c_p, _c_i64, _c_p, _c_p, _c_p, _c_i64, _c_i64,
                                                  ctypes.c_double, _c_p, _c_p, ctypes.c_int, _c_p]),
    "dnp_patch_fields_ordered_f32": (ctypes.c_int, [_c_p, _c_i64, _c_i64, _c_p, _c_i64, _c_p, _c_p, _c_p, _c_i64, _c_i64, _c_p,
                                                    ctypes.c_float, _c_p, _c_p, ctypes.c_int, ctypes.c_int, _c_p, _c_sz, _c_p]),
    "dnp_patch_fields_ordered_f64": (ctypes.c_int, [_c_p, _c_i64, _c_i64, _c_p, _c_i64, _c_p, _c_p, _c_p, _c_i64, _c_i64, _c_p,
                                                    ctypes.c_double, _c_p, _c_p, ctypes.c_int, _c_p]),
    "dnp_patch_boxes_f64": (ctypes.c_int, [_c_p, _c_i64, _c_i64, _c_p, _c_p, _c_i64, _c_p, _c_p]),
    "dnp_tile_boxes_f64": (ctypes.c_int, [_c_p, _c_i64, _c_i64, _c_i64, _c_p, _c_p]),
    "dnp_patch_exchange_bytes": (_c_sz, [_c_i64, _c_i64]),
    "dnp_exchange_init": (ctypes.c_int, [_c_p, _c_sz, _c_p]),
    "dnp_check_tile_groups": (ctypes.c_int, [_c_p, _c_i64, ctypes.c_int, _c_p, _c_p]),
    "dnp_interactions_from_tiles": (ctypes.c_int, [_c_p, ctypes.c_int, _c_i64, _c_i64, _c_p, _c_p, _c_i64, _c_p, _c_p]),
    "dnp_interactions_f32": (ctypes.c_int, [_c_p, _c_i64, _c_i64, _c_p, _c_i64, _c_p, _c_p, _c_i64, _c_p, _c_p]),
    "dnp_interactions_f64": (ctypes.c_int, [_c_p, _c_i64, _c_i64, _c_p, _c_i64, _c_p, _c_p, _c_i64, _c_p, _c_p]),
    "dnp_combine_fields_f32": (ctypes.c_int, [_c_p, _c_i64, _c_i64, _c_p, _c_p, _c_i64, _c_p, ctypes.c_int, _c_p]),
    "dnp_xie_pairs_f32": (ctypes.c_int, [_c_p, _c_i64, _c_i64, _c_p, _c_i64, _c_i64, ctypes.c_float, ctypes.c_int,
                                         _c_p, _c_p]),
    "dnp_xie_pairs_f64": (ctypes.c_int, [_c_p, _c_i64, _c_i64, _c_p, _c_i64, _c_i64, ctypes.c_double, ctypes.c_int,
                                         _c_p, _c_p]),
    "dnp_xie_knn_f32": (ctypes.c_int, [_c_p, _c_i64, _c_i64, _c_p, _c_i64, _c_i64, _c_i64, _c_p, _c_p, _c_p]),
    "dnp_xie_knn_f64": (ctypes.c_int, [_c_p, _c_i64, _c_i64, _c_p, _c_i64, _c_i64, _c_i64, _c_p, _c_p, _c_p]),
    "dnp_xie_pairs_knn_f32": (ctypes.c_int, [_c_p, _c_i64, _c_i64, _c_p, _c_i64, _c_i64, ctypes.c_float, ctypes.c_int,
                                             _c_p, _c_p, _c_p, _c_p]),
    "dnp_xie_pairs_knn_f64": (ctypes.c_int, [_c_p, _c_i64, _c_i64, _c_p, _c_i64, _c_i64, ctypes.c_double, ctypes.c_int,
                                             _c_p, _c_p, _c_p, _c_p]),
    "dnp_xie_order_f32": (ctypes.c_int, [_c_p, _c_i64, _c_p, _c_i64, _c_p, _c_p, _c_p]),
    "dnp_xie_order_f64": (ctypes.c_int, [_c_p, _c_i64, _c_p, _c_i64, _c_p, _c_p, _c_p]),
    "dnp_xie_order_workspace_bytes": (ctypes.c_size_t, [_c_i64, _c_i64, ctypes.c_int]),
    "dnp_xie_order_blocked_f32": (ctypes.c_int, [_c_p, _c_i64, _c_p, _c_i64, _c_p, _c_p, _c_p, ctypes.c_size_t, _c_p]),
    "dnp_xie_order_blocked_f64": (ctypes.c_int, [_c_p, _c_i64, _c_p, _c_i64, _c_p, _c_p, _c_p, ctypes.c_size_t, _c_p]),
    "dnp_xie_rowdots_f32": (ctypes.c_int, [_c_p, _c_i64, _c_p, _c_i64, _c_p, _c_p]),
    "dnp_xie_rowdots_f64": (ctypes.c_int, [_c_p, _c_i64, _c_p, _c_i64, _c_p, _c_p]),
    "dnp_point_greedy_workspace_bytes": (_c_sz, [_c_i64, ctypes.c_int]),
    "dnp_point_greedy_max_points": (ctypes.c_int, []),
    "dnp_point_greedy_f32": (ctypes.c_int, [_c_p, _c_i64, _c_i64, _c_i64, ctypes.c_float, ctypes.c_int, _c_p, _c_p,
                                            ctypes.c_int, ctypes.c_int, _c_p, _c_sz, _c_p]),
    "dnp_point_greedy_f64": (ctypes.c_int, [_c_p, _c_i64, _c_i64, _c_i64, ctypes.c_double, ctypes.c_int, _c_p, _c_p,
                                            ctypes.c_int, ctypes.c_int, _c_p, _c_sz, _c_p]),
    "dnp_patch_pca_f32": (ctypes.c_int, [_c_p, _c_i64, _c_p, _c_p, _c_i64, _c_p, _c_p, _c_p, _c_p]),
    "dnp_patch_pca_f64": (ctypes.c_int, [_c_p, _c_i64, _c_p, _c_p, _c_i64, _c_p, _c_p, _c_p, _c_p]),
    "dnp_patch_greedy_max_patches": (ctypes.c_int, []),
    "dnp_patch_greedy": (ctypes.c_int, [_c_p, _c_i64, _c_p, _c_p, _c_p, _c_p, _c_p]),
    "dnp_combine_signed_f32": (ctypes.c_int, [_c_p, _c_i64, _c_i64, _c_p, _c_i64, _c_i64, _c_p, ctypes.c_int, _c_p]),
    "dnp_patch_layout_f32": (ctypes.c_int, [_c_p, _c_i64, _c_p, _c_p, _c_i64, _c_p, _c_p, _c_p]),
    "dnp_patch_layout_f64": (ctypes.c_int, [_c_p, _c_i64, _c_p, _c_p, _c_i64, _c_p, _c_p, _c_p]),
    "dnp_combine_signed_f64": (ctypes.c_int, [_c_p, _c_i64, _c_i64, _c_p, _c_i64, _c_i64, _c_p, ctypes.c_int, _c_p]),
    "dnp_patch_finish_f64": (ctypes.c_int, [_c_p, _c_i64, _c_i64, _c_p, _c_p, _c_p, _c_p, _c_p, _c_p, _c_p, _c_i64,
                                            ctypes.c_int, _c_p]),
    "dnp_patch_finish_f32": (ctypes.c_int, [_c_p, _c_i64, _c_i64, _c_p, _c_p, _c_p, _c_p, _c_p, _c_p, _c_p, _c_i64,
                                            ctypes.c_int, _c_p]),
    "dnp_rest_finish_f32": (ctypes.c_int, [_c_p, _c_i64, _c_p, _c_p, _c_i64, _c_p, _c_p, _c_p]),
    "dnp_rest_finish_f64": (ctypes.c_int, [_c_p, _c_i64, _c_p, _c_p, _c_i64, _c_p, _c_p, _c_p]),
    "dnp_xyz_format_bound": (_c_i64, [_c_i64, _c_i64]),
    "dnp_xyz_format_f32": (_c_i64, [_c_p, _c_i64, _c_i64, _c_p, _c_i64]),
    "dnp_xyz_parse_f32": (_c_i64, [_c_p, _c_i64, _c_p, _c_i64, _c_p]),
    "dnp_merge_cells": (ctypes.c_int, [_c_p, _c_p, _c_i64, _c_i64, _c_p, _c_p, _c_p, _c_p]),
}

_lib = None
_lock = threading.Lock()


class DnpError(RuntimeError):
    pass


def load():
    """dlopen libdnp.so and declare every prototype.  Raises if the library is not built."""
    global _lib
    if _lib is not None:
        return _lib
    with _lock:
        if _lib is not None:
            return _lib
        stale = False
        if "DNP_LIB" not in os.environ and os.path.exists(LIB_PATH):
            try:                                        # a source edited after the last build: never run yesterday's kernels
                from . import build as _build
                stale = _build.needs_build()
            except Exception:
                stale = False
        if not os.path.exists(LIB_PATH) or stale:
            # a fresh checkout (or an edited source): compile the HIP library (hipcc is part of the ROCm image); there is
            # no CPU fallback, so a failed build is an error, not a degradation
            if "DNP_LIB" in os.environ:
                raise DnpError(f"DNP_LIB={LIB_PATH} does not exist")
            try:
                from . import build as _build
                _build.build(verbose=False)
            except Exception as exc:
                if not os.path.exists(LIB_PATH):
                    raise DnpError(f"{LIB_PATH} not found and building it failed ({exc}); run "
                                   "`python -m dipole_normal_prop_amd.build` - there is no CPU fallback for the "
                                   "field kernels") from exc
                # a library exists but a source looks newer and the rebuild failed (no hipcc on this box, a read-only tree):
                # say so and use the library that is there - its dnp_version() is checked against the header below
                import warnings
                warnings.warn(f"libdnp.so is older than its sources and rebuilding it failed ({exc}); using the existing library")
        lib = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            try:
                fn = getattr(lib, name)
            except AttributeError as exc:               # an older library than this binding: never call it with a shifted ABI
                raise DnpError(f"{LIB_PATH} does not export {name}: it was built from other sources than this package; "
                               "run `python -m dipole_normal_prop_amd.build --force`") from exc
            fn.restype = res
            fn.argtypes = args
        lib.dnp_version.restype = ctypes.c_int
        if lib.dnp_version() != ABI_VERSION:
            raise DnpError(f"{LIB_PATH} reports ABI version {lib.dnp_version()}, this binding is written for {ABI_VERSION}; "
                           "run `python -m dipole_normal_prop_amd.build --force`")
        _lib = lib
    return _lib


def check(rc):
    if rc != 0:
        msg = load().dnp_last_error().decode("utf-8", "replace")
        raise DnpError(f"libdnp error {rc}: {msg}")


_device_ok = False


def require_device():
    """The loaded library, after checking once per process that a HIP device is visible."""
    global _device_ok
    lib = _lib if _lib is not None else load()
    if not _device_ok:
        if not torch.cuda.is_available() or lib.dnp_device_count() < 1:
            raise DnpError("no HIP device visible: the dipole field kernels run on MI355X (gfx950) only; "
                           "there is no CPU fallback")
        _device_ok = True
    return lib


def ptr(t):
    return ctypes.c_void_p(t.data_ptr()) if t is not None else None


def current_stream():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
