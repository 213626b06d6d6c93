"""ctypes binding of libl3d_hip.so (the C ABI declared in include/l3d.h).

There is no CPU fallback: if the shared library is missing or cannot be loaded
the import of this module raises, and every wrapper raises ``L3DError`` when
the library reports a failure (including "no HIP device").
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libl3d_hip.so")


class L3DError(RuntimeError):
    pass


class SgbmParams(C.Structure):
    """l3d_sgbm_params: cv2.StereoSGBM_create's keyword set
    (reference camera/single_usb_stereo_camera.py:252-274)."""

    _fields_ = [
        ("minDisparity", C.c_int),
        ("numDisparities", C.c_int),
        ("blockSize", C.c_int),
        ("P1", C.c_int),
        ("P2", C.c_int),
        ("disp12MaxDiff", C.c_int),
        ("preFilterCap", C.c_int),
        ("uniquenessRatio", C.c_int),
        ("speckleWindowSize", C.c_int),
        ("speckleRange", C.c_int),
        ("mode", C.c_int),
    ]

    def as_dict(self):
        return {k: getattr(self, k) for k, _ in self._fields_}


class WlsParams(C.Structure):
    _fields_ = [
        ("lambda_", C.c_double),
        ("sigma_color", C.c_double),
        ("min_disparity", C.c_int),
        ("num_disparities", C.c_int),
        ("lrc_thresh", C.c_int),
        ("disc_radius", C.c_int),
        ("disc_rolloff", C.c_float),
        ("fgs_iters", C.c_int),
        ("fgs_attenuation", C.c_float),
    ]


def _load():
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(hipcc --offload-arch=gfx950); there is no CPU fallback."
        )
    # One process can hold only one HIP runtime (SONAME libamdhip64.so.7).  PyTorch wheels bundle
    # their own copy; if the system copy were loaded first, a later `import torch` would find no
    # GPU.  So bind to PyTorch's copy when PyTorch is installed, without importing torch itself.
    try:
        import importlib.util

        spec = importlib.util.find_spec("torch")
        if spec is not None and spec.submodule_search_locations:
            cand = os.path.join(list(spec.submodule_search_locations)[0], "lib", "libamdhip64.so")
            if os.path.exists(cand):
                C.CDLL(cand, mode=C.RTLD_GLOBAL)
    except Exception:
        pass
    lib = C.CDLL(LIB_PATH)
    vp, i, sz, ss = C.c_void_p, C.c_int, C.c_size_t, C.c_ssize_t
    pi = C.POINTER(C.c_int)
    sig = {
        "l3d_last_error": (C.c_char_p, []),
        "l3d_version": (C.c_char_p, []),
        "l3d_device_count": (i, [pi]),
        "l3d_sgbm_create": (i, [C.POINTER(SgbmParams), i, i, i, i, C.POINTER(vp)]),
        "l3d_sgbm_destroy": (None, [vp]),
        "l3d_sgbm_compute": (i, [vp, vp, vp, i, ss, ss, vp]),
        "l3d_sgbm_compute_device": (i, [vp, vp, vp, i, vp, vp]),
        "l3d_sgbm_compute_depth_device": (i, [vp, vp, vp, i, vp, vp, vp, vp]),
        "l3d_sgbm_postfilter": (i, [vp, vp, i, vp]),
        "l3d_sgbm_fetch_stage": (i, [vp, i, vp, sz]),
        "l3d_sgbm_geometry": (i, [C.POINTER(SgbmParams), i, i, pi]),
        "l3d_sgbm_profile": (i, [vp, i]),
        "l3d_sgbm_profile_read": (i, [vp, C.POINTER(C.c_double), i, C.POINTER(C.c_long)]),
        "l3d_sgbm_profile_kernel_name": (C.c_char_p, [i]),
        "l3d_sgbm_profile_kernel_count": (i, []),
        "l3d_wls_create": (i, [C.POINTER(WlsParams), i, i, i, i, C.POINTER(vp)]),
        "l3d_wls_destroy": (None, [vp]),
        "l3d_wls_filter": (i, [vp, vp, vp, vp, i, vp]),
        "l3d_wls_filter_device": (i, [vp, vp, vp, vp, i, vp, vp]),
        "l3d_wls_fetch_confidence": (i, [vp, vp, sz]),
        "l3d_bgr2gray_device": (i, [vp, i, i, i, vp, i, vp]),
        "l3d_bgr2gray": (i, [vp, i, i, i, vp, i]),
        "l3d_depth_from_disp16_device": (i, [vp, vp, i, i, i, vp, vp, i, vp]),
        "l3d_depth_from_disp16": (i, [vp, vp, i, i, i, vp, vp, i]),
        "l3d_steger_create": (i, [C.c_double, i, i, i, i, i, C.POINTER(vp)]),
        "l3d_steger_destroy": (None, [vp]),
        "l3d_steger_extract": (i, [vp, vp, i, i, i, i, pi, vp, i, vp]),
        "l3d_steger_extract_device": (i, [vp, vp, i, i, i, i, pi, vp, i, vp, vp]),
        "l3d_steger_extract_device_static": (i, [C.c_double, i, vp, i, i, i, i, pi, vp, i, vp, i, vp]),
        "l3d_simple_create": (i, [i, i, i, C.POINTER(vp)]),
        "l3d_simple_destroy": (None, [vp]),
        "l3d_simple_extract": (i, [vp, vp, i, i, pi, pi, i, C.c_double, vp, pi, vp]),
        "l3d_simple_extract_device": (i, [vp, vp, i, i, pi, pi, i, C.c_double, vp, pi, vp]),
        "l3d_simple_create_batch": (i, [i, i, i, i, C.POINTER(vp)]),
        "l3d_simple_extract_batch": (i, [vp, vp, i, i, i, pi, pi, i, C.c_double, vp, vp]),
        "l3d_simple_extract_batch_device": (i, [vp, vp, i, i, i, pi, pi, i, C.c_double, vp, vp, vp]),
        "l3d_steger_create_improved": (i, [C.c_double, i, C.c_double, i, i, i, i, C.POINTER(vp)]),
        "l3d_steger_extract_improved": (i, [vp, vp, i, i, i, i, i, vp, i, vp]),
        "l3d_steger_extract_improved_device": (i, [vp, vp, i, i, i, i, i, vp, i, vp, vp]),
        "l3d_hybrid_extract": (i, [vp, vp, vp, i, i, pi, pi, i, vp, i, pi, vp]),
        "l3d_hybrid_extract_device": (i, [vp, vp, vp, i, i, pi, pi, i, vp, i, vp, vp]),
        "l3d_remap_create": (i, [vp, vp, i, i, i, C.POINTER(vp)]),
        "l3d_remap_create_rectify": (i, [vp, vp, i, vp, i, i, i, C.POINTER(vp)]),
        "l3d_remap_destroy": (None, [vp]),
        "l3d_remap_fetch_maps": (i, [vp, vp, vp]),
        "l3d_remap_apply": (i, [vp, vp, i, i, i, ss, ss, i, vp, vp]),
        "l3d_remap_apply_device": (i, [vp, vp, i, i, i, ss, ss, i, vp, vp, vp]),
        "l3d_reconstruct_from_depth": (i, [vp, i, vp, i, i, vp, vp, pi, i]),
        "l3d_triangulate_disparity": (i, [vp, i, vp, i, i, i, vp, i, C.c_double, vp, pi, i]),
        "l3d_triangulate_disparity_device": (i, [vp, i, vp, i, i, vp, i, C.c_double, vp, vp, vp]),
        "l3d_laser_depth_map": (i, [vp, i, vp, i, i, i, vp, vp, i]),
        "l3d_voxel_downsample": (i, [vp, i, C.c_double, vp, pi, i]),
        "l3d_sor_simple": (i, [vp, i, i, C.c_double, vp, pi, i]),
        "l3d_nn_distances": (i, [vp, i, vp, i]),
        "l3d_reconstruct_plane": (i, [vp, i, vp, vp, i, C.c_double, vp, i]),
        "l3d_reconstruct_from_depth_device": (i, [vp, vp, i, i, vp, i, i, vp, vp, vp, vp]),
        "l3d_reconstruct_from_depth_uv64_device": (i, [vp, i, vp, i, i, vp, vp, vp, vp]),
        "l3d_reconstruct_plane_device": (i, [vp, vp, i, i, vp, vp, i, C.c_double, vp, vp]),
    }
    for name, (res, args) in sig.items():
        fn = getattr(lib, name)  # AttributeError here = header and library disagree
        fn.restype = res
        fn.argtypes = args
    return lib, tuple(sig)


lib, EXPORTED = _load()


def check(rc: int, what: str = "l3d call"):
    if rc != 0:
        msg = lib.l3d_last_error()
        raise L3DError(f"{what} failed (status {rc}): {msg.decode(errors='replace') if msg else ''}")


def device_count() -> int:
    n = C.c_int(0)
    rc = lib.l3d_device_count(C.byref(n))
    return n.value if rc == 0 else 0


def ptr(a: np.ndarray):
    return a.ctypes.data_as(C.c_void_p)
