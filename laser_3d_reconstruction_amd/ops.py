"""Element-wise GPU operators of the depth path (gray conversion, disparity ->
depth) and the laser-pixel -> 3-D operators, as thin wrappers over the C ABI."""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _native as N


def _stream(t):
    import torch

    return C.c_void_p(torch.cuda.current_stream(t.device).cuda_stream)


def bgr2gray(bgr: np.ndarray, device: int = 0) -> np.ndarray:
    """cv2.cvtColor(bgr, cv2.COLOR_BGR2GRAY) for (H,W,3) or (B,H,W,3) uint8."""
    bgr = np.ascontiguousarray(bgr)
    if bgr.dtype != np.uint8 or bgr.shape[-1] != 3 or bgr.ndim not in (3, 4):
        raise ValueError("bgr2gray expects (H,W,3) or (B,H,W,3) uint8")
    shp = bgr.shape[:-1]
    b = shp[0] if bgr.ndim == 4 else 1
    h, w = shp[-2:]
    out = np.empty(shp, np.uint8)
    N.check(N.lib.l3d_bgr2gray(N.ptr(bgr), w, h, b, N.ptr(out), device), "l3d_bgr2gray")
    return out


def bgr2gray_device(bgr, out=None):
    import torch

    assert bgr.is_cuda and bgr.dtype == torch.uint8 and bgr.is_contiguous() and bgr.shape[-1] == 3
    shp = bgr.shape[:-1]
    b = shp[0] if bgr.dim() == 4 else 1
    h, w = shp[-2:]
    if out is None:
        out = torch.empty(tuple(shp), dtype=torch.uint8, device=bgr.device)
    N.check(N.lib.l3d_bgr2gray_device(C.c_void_p(bgr.data_ptr()), w, h, b, C.c_void_p(out.data_ptr()),
                                      bgr.device.index or 0, _stream(bgr)), "l3d_bgr2gray_device")
    return out


def depth_from_disparity16(disp16: np.ndarray, Q=None, want_xyz: bool = False, device: int = 0):
    """`disp16.astype(float32)/16` -> cv2.reprojectImageTo3D(.., Q)[:,:,2] -> validity masks
    (reference camera/single_usb_stereo_camera.py:335-357).  Q=None = uncalibrated fallback."""
    disp16 = np.ascontiguousarray(disp16)
    if disp16.dtype != np.int16 or disp16.ndim not in (2, 3):
        raise ValueError("depth_from_disparity16 expects (H,W) or (B,H,W) int16")
    b = disp16.shape[0] if disp16.ndim == 3 else 1
    h, w = disp16.shape[-2:]
    depth = np.empty(disp16.shape, np.float32)
    xyz = np.empty(disp16.shape + (3,), np.float32) if want_xyz else None
    q = np.ascontiguousarray(Q, np.float64).reshape(16) if Q is not None else None
    N.check(N.lib.l3d_depth_from_disp16(N.ptr(disp16), N.ptr(q) if q is not None else None, w, h, b, N.ptr(depth),
                                        N.ptr(xyz) if want_xyz else None, device), "l3d_depth_from_disp16")
    return (depth, xyz) if want_xyz else depth


def depth_from_disparity16_device(disp16, Q=None, out=None, xyz=None):
    import torch

    assert disp16.is_cuda and disp16.dtype == torch.int16 and disp16.is_contiguous()
    b = disp16.shape[0] if disp16.dim() == 3 else 1
    h, w = disp16.shape[-2:]
    if out is None:
        out = torch.empty(tuple(disp16.shape), dtype=torch.float32, device=disp16.device)
    q = np.ascontiguousarray(Q, np.float64).reshape(16) if Q is not None else None
    N.check(N.lib.l3d_depth_from_disp16_device(C.c_void_p(disp16.data_ptr()), N.ptr(q) if q is not None else None, w, h, b,
                                               C.c_void_p(out.data_ptr()),
                                               C.c_void_p(xyz.data_ptr()) if xyz is not None else None,
                                               disp16.device.index or 0, _stream(disp16)),
            "l3d_depth_from_disp16_device")
    return out


def reconstruct_from_depth(uv, depth: np.ndarray, K, device: int = 0) -> np.ndarray:
    uv = np.ascontiguousarray(uv, np.float64).reshape(-1, 2)
    depth = np.ascontiguousarray(depth, np.float32)
    K = np.ascontiguousarray(K, np.float64).reshape(9)
    h, w = depth.shape
    out = np.empty((len(uv), 3), np.float64)
    n = C.c_int(0)
    N.check(N.lib.l3d_reconstruct_from_depth(N.ptr(uv), len(uv), N.ptr(depth), w, h, N.ptr(K), N.ptr(out), C.byref(n), device),
            "l3d_reconstruct_from_depth")
    return out[: n.value]


def reconstruct_plane(uv, K_inv, plane, refraction: bool, n_water: float = 1.33, device: int = 0) -> np.ndarray:
    uv = np.ascontiguousarray(uv, np.float64).reshape(-1, 2)
    kinv = np.ascontiguousarray(K_inv, np.float64).reshape(9)
    plane = np.ascontiguousarray(plane, np.float64).reshape(4)
    out = np.empty((len(uv), 3), np.float64)
    N.check(N.lib.l3d_reconstruct_plane(N.ptr(uv), len(uv), N.ptr(kinv), N.ptr(plane), int(bool(refraction)),
                                        C.c_double(n_water), N.ptr(out), device), "l3d_reconstruct_plane")
    return out
