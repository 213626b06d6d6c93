"""Stereo rectification: the calibration-time geometry on the host, the maps and
the per-frame remap on the GPU.

``stereo_rectify`` restates cv2.stereoRectify (Bouguet's algorithm; the reference
calls it with flags=CALIB_ZERO_DISPARITY, alpha=0,
camera/single_usb_stereo_camera.py:176-187).  It runs once per calibration on a
handful of 3x3 matrices, so it is plain numpy.  OpenCV is not available where
this was written: the restatement follows modules/calib3d/src/calibration.cpp
(cvStereoRectify, icvGetRectangles, cvUndistortPoints with 5 fixed iterations),
including the float32 storage of the intermediate point arrays, and is checked
by geometric properties only.

``Rectifier`` holds the maps of one camera on the device
(cv2.initUndistortRectifyMap, :190-206) and applies cv2.remap(INTER_LINEAR)
(:313-314) to batches of frames.
"""
from __future__ import annotations

import ctypes as C
import math
from typing import Optional, Tuple

import numpy as np

from .. import _native as N

CALIB_ZERO_DISPARITY = 1024  # cv2.CALIB_ZERO_DISPARITY


# ----------------------------------------------------------------------- host geometry
def rodrigues_to_matrix(om) -> np.ndarray:
    om = np.asarray(om, np.float64).reshape(3)
    theta = float(np.linalg.norm(om))
    if theta < np.finfo(np.float64).eps:
        return np.eye(3)
    c, s = math.cos(theta), math.sin(theta)
    r = om / theta
    rx = np.array([[0, -r[2], r[1]], [r[2], 0, -r[0]], [-r[1], r[0], 0]])
    return c * np.eye(3) + (1 - c) * np.outer(r, r) + s * rx


def rodrigues_to_vector(R) -> np.ndarray:
    R = np.asarray(R, np.float64).reshape(3, 3)
    U, _, Vt = np.linalg.svd(R)
    R = U @ Vt
    r = np.array([R[2, 1] - R[1, 2], R[0, 2] - R[2, 0], R[1, 0] - R[0, 1]])
    s = math.sqrt(float(r @ r) * 0.25)
    c = min(max((R[0, 0] + R[1, 1] + R[2, 2] - 1) * 0.5, -1.0), 1.0)
    theta = math.acos(c)
    if s < 1e-5:
        if c > 0:
            return np.zeros(3)
        t = (R[0, 0] + 1) * 0.5
        rx = math.sqrt(max(t, 0.0))
        t = (R[1, 1] + 1) * 0.5
        ry = math.sqrt(max(t, 0.0)) * (-1.0 if R[0, 1] < 0 else 1.0)
        t = (R[2, 2] + 1) * 0.5
        rz = math.sqrt(max(t, 0.0)) * (-1.0 if R[0, 2] < 0 else 1.0)
        if abs(rx) < abs(ry) and abs(rx) < abs(rz) and (R[1, 2] > 0) != (ry * rz > 0):
            rz = -rz
        v = np.array([rx, ry, rz])
        return v * (theta / np.linalg.norm(v))
    return r * (theta / (2 * s))


def _dist12(dist) -> np.ndarray:
    d = np.zeros(12)
    if dist is not None:
        v = np.asarray(dist, np.float64).reshape(-1)
        if v.size not in (0, 4, 5, 8, 12):
            raise ValueError(f"unsupported number of distortion coefficients: {v.size}")
        d[: v.size] = v
    return d


def undistort_points(pts, K, dist, R=None, P=None, iters: int = 5) -> np.ndarray:
    """cvUndistortPoints with a fixed iteration count; pts (n,2) -> (n,2) float32."""
    K = np.asarray(K, np.float64).reshape(3, 3)
    k = _dist12(dist)
    # OpenCV's coefficient order: k1 k2 p1 p2 k3 k4 k5 k6 s1 s2 s3 s4
    k1, k2, p1, p2, k3, k4, k5, k6, s1, s2, s3, s4 = k
    fx, fy, cx, cy = K[0, 0], K[1, 1], K[0, 2], K[1, 2]
    RR = np.eye(3) if R is None else np.asarray(R, np.float64).reshape(3, 3)
    if P is not None:
        RR = np.asarray(P, np.float64)[:3, :3] @ RR
    out = np.empty((len(pts), 2), np.float32)
    for i, (u, v) in enumerate(np.asarray(pts, np.float64)):
        x = x0 = (u - cx) / fx
        y = y0 = (v - cy) / fy
        if np.any(k != 0):
            for _ in range(iters):
                r2 = x * x + y * y
                icdist = (1 + ((k6 * r2 + k5) * r2 + k4) * r2) / (1 + ((k3 * r2 + k2) * r2 + k1) * r2)
                if icdist < 0:
                    x, y = x0, y0
                    break
                dx = 2 * p1 * x * y + p2 * (r2 + 2 * x * x) + s1 * r2 + s2 * r2 * r2
                dy = p1 * (r2 + 2 * y * y) + 2 * p2 * x * y + s3 * r2 + s4 * r2 * r2
                x = (x0 - dx) * icdist
                y = (y0 - dy) * icdist
        xx = RR[0, 0] * x + RR[0, 1] * y + RR[0, 2]
        yy = RR[1, 0] * x + RR[1, 1] * y + RR[1, 2]
        ww = 1.0 / (RR[2, 0] * x + RR[2, 1] * y + RR[2, 2])
        out[i] = (xx * ww, yy * ww)
    return out


def _rectangles(K, dist, R, P, size):
    """icvGetRectangles: inscribed and circumscribed rectangle of the rectified image of a 9x9 grid."""
    w, h = size
    n = 9
    pts = np.array([[np.float32(x) * np.float32(w) / np.float32(n - 1), np.float32(y) * np.float32(h) / np.float32(n - 1)]
                    for y in range(n) for x in range(n)], np.float32)
    p = undistort_points(pts, K, dist, R, P).reshape(n, n, 2)
    f = np.float32
    ix0, ix1 = f(p[:, 0, 0].max()), f(p[:, n - 1, 0].min())
    iy0, iy1 = f(p[0, :, 1].max()), f(p[n - 1, :, 1].min())
    ox0, ox1 = f(p[:, :, 0].min()), f(p[:, :, 0].max())
    oy0, oy1 = f(p[:, :, 1].min()), f(p[:, :, 1].max())
    inner = (float(ix0), float(iy0), float(f(ix1 - ix0)), float(f(iy1 - iy0)))
    outer = (float(ox0), float(oy0), float(f(ox1 - ox0)), float(f(oy1 - oy0)))
    return inner, outer


def _rect_and(a, b):
    x0, y0 = max(a[0], b[0]), max(a[1], b[1])
    x1, y1 = min(a[0] + a[2], b[0] + b[2]), min(a[1] + a[3], b[1] + b[3])
    if x1 <= x0 or y1 <= y0:
        return (0, 0, 0, 0)
    return (x0, y0, x1 - x0, y1 - y0)


def stereo_rectify(K1, D1, K2, D2, image_size, R, T, flags: int = CALIB_ZERO_DISPARITY, alpha: float = -1.0,
                   new_image_size: Tuple[int, int] = (0, 0)):
    """cv2.stereoRectify -> (R1, R2, P1, P2, Q, roi1, roi2); image_size = (width, height)."""
    K1 = np.asarray(K1, np.float64).reshape(3, 3)
    K2 = np.asarray(K2, np.float64).reshape(3, 3)
    R = np.asarray(R, np.float64)
    T = np.asarray(T, np.float64).reshape(3)
    nx, ny = int(image_size[0]), int(image_size[1])
    om = rodrigues_to_vector(R) if R.shape == (3, 3) else R.reshape(3).copy()
    om = om * -0.5                          # each camera takes half of the rotation
    r_r = rodrigues_to_matrix(om)
    t = r_r @ T
    idx = 0 if abs(t[0]) > abs(t[1]) else 1  # horizontal or vertical stereo
    c, nt = t[idx], float(np.linalg.norm(t))
    if not nt > 0:
        raise ValueError("stereo_rectify: zero baseline")
    uu = np.zeros(3)
    uu[idx] = 1.0 if c > 0 else -1.0
    ww = np.cross(t, uu)                     # rotation that takes the baseline onto the image axis
    nw = float(np.linalg.norm(ww))
    if nw > 0:
        ww = ww * (math.acos(abs(c) / nt) / nw)
    wR = rodrigues_to_matrix(ww)
    R1 = wR @ r_r.T
    R2 = wR @ r_r
    t = R2 @ T

    dists = (D1, D2)
    Ks = (K1, K2)
    fc_new = np.finfo(np.float64).max
    for k in range(2):
        d = _dist12(dists[k])
        fc = Ks[k][idx ^ 1, idx ^ 1]
        if d[0] < 0:
            fc *= 1 + d[0] * (nx * nx + ny * ny) / (4 * fc * fc)
        fc_new = min(fc_new, fc)

    cc_new = [None, None]
    for k in range(2):
        corners = np.array([[(i % 2) * (nx - 1), (0 if i < 2 else 1) * (ny - 1)] for i in range(4)], np.float32)
        p = undistort_points(corners, Ks[k], dists[k])             # normalised, float32
        p3 = np.concatenate([p, np.ones((4, 1), np.float32)], axis=1).astype(np.float64)
        Rk = R1 if k == 0 else R2
        # cvProjectPoints2 takes the rotation as a vector and rebuilds the matrix from it
        Rk = rodrigues_to_matrix(rodrigues_to_vector(Rk))
        X = p3 @ Rk.T
        proj = np.stack([fc_new * X[:, 0] / X[:, 2], fc_new * X[:, 1] / X[:, 2]], axis=1).astype(np.float32)
        avg = proj.astype(np.float64).mean(axis=0)
        cc_new[k] = [(nx - 1) / 2 - avg[0], (ny - 1) / 2 - avg[1]]
    if flags & CALIB_ZERO_DISPARITY:
        cc_new[0][0] = cc_new[1][0] = (cc_new[0][0] + cc_new[1][0]) * 0.5
        cc_new[0][1] = cc_new[1][1] = (cc_new[0][1] + cc_new[1][1]) * 0.5
    elif idx == 0:
        cc_new[0][1] = cc_new[1][1] = (cc_new[0][1] + cc_new[1][1]) * 0.5
    else:
        cc_new[0][0] = cc_new[1][0] = (cc_new[0][0] + cc_new[1][0]) * 0.5

    P1 = np.zeros((3, 4))
    P1[0, 0] = P1[1, 1] = fc_new
    P1[0, 2], P1[1, 2], P1[2, 2] = cc_new[0][0], cc_new[0][1], 1.0
    P2 = np.zeros((3, 4))
    P2[0, 0] = P2[1, 1] = fc_new
    P2[0, 2], P2[1, 2], P2[2, 2] = cc_new[1][0], cc_new[1][1], 1.0
    P2[idx, 3] = t[idx] * fc_new             # baseline * focal length

    alpha = min(alpha, 1.0)
    inner1, outer1 = _rectangles(K1, D1, R1, P1, (nx, ny))
    inner2, outer2 = _rectangles(K2, D2, R2, P2, (nx, ny))
    nw_, nh_ = new_image_size if new_image_size[0] * new_image_size[1] != 0 else (nx, ny)
    cx1_0, cy1_0, cx2_0, cy2_0 = cc_new[0][0], cc_new[0][1], cc_new[1][0], cc_new[1][1]
    cx1, cy1 = nw_ * cx1_0 / nx, nh_ * cy1_0 / ny
    cx2, cy2 = nw_ * cx2_0 / nx, nh_ * cy2_0 / ny
    s = 1.0
    if alpha >= 0:
        def spans(cx, cy, cx0, cy0, r):
            return (cx / (cx0 - r[0]), cy / (cy0 - r[1]), (nw_ - 1 - cx) / (r[0] + r[2] - cx0), (nh_ - 1 - cy) / (r[1] + r[3] - cy0))
        s0 = max(max(spans(cx1, cy1, cx1_0, cy1_0, inner1)), max(spans(cx2, cy2, cx2_0, cy2_0, inner2)))
        s1 = min(min(spans(cx1, cy1, cx1_0, cy1_0, outer1)), min(spans(cx2, cy2, cx2_0, cy2_0, outer2)))
        s = s0 * (1 - alpha) + s1 * alpha
    fc_new *= s
    for Pm, cx, cy in ((P1, cx1, cy1), (P2, cx2, cy2)):
        Pm[0, 0] = Pm[1, 1] = fc_new
        Pm[0, 2], Pm[1, 2] = cx, cy
    P2[idx, 3] = s * P2[idx, 3]

    def roi(inner, cx0, cy0, cx, cy):
        r = (math.ceil((inner[0] - cx0) * s + cx), math.ceil((inner[1] - cy0) * s + cy), math.floor(inner[2] * s), math.floor(inner[3] * s))
        return _rect_and(r, (0, 0, nw_, nh_))
    roi1 = roi(inner1, cx1_0, cy1_0, cx1, cy1)
    roi2 = roi(inner2, cx2_0, cy2_0, cx2, cy2)

    Q = np.array([[1, 0, 0, -cx1],
                  [0, 1, 0, -cy1],
                  [0, 0, 0, fc_new],
                  [0, 0, -1.0 / t[idx], (cx1 - cx2 if idx == 0 else cy1 - cy2) / t[idx]]])
    return R1, R2, P1, P2, Q, roi1, roi2


# ------------------------------------------------------------------------------ device side
class Rectifier:
    """Rectification maps of one camera, resident on the GPU, and their application.

    ``Rectifier.from_calibration(K, dist, R, P, (w, h))`` is cv2.initUndistortRectifyMap;
    ``Rectifier.from_maps(map_x, map_y)`` takes ready maps; ``remap`` is cv2.remap with
    INTER_LINEAR and the default constant border."""

    def __init__(self, handle, size, device):
        self._h = handle
        self.size = size  # (width, height)
        self.device = device
        self._maps = None

    @classmethod
    def from_calibration(cls, K, dist, R, P, size, device: int = 0) -> "Rectifier":
        w, h = int(size[0]), int(size[1])
        K = np.ascontiguousarray(K, np.float64).reshape(3, 3)
        d = np.ascontiguousarray(np.asarray(dist, np.float64).reshape(-1)) if dist is not None else np.zeros(0)
        if d.size == 14 and not np.any(d[12:]):
            d = np.ascontiguousarray(d[:12])
        iR = np.ascontiguousarray(np.linalg.inv(np.asarray(P, np.float64)[:3, :3] @ np.asarray(R, np.float64).reshape(3, 3)))
        hd = C.c_void_p()
        N.check(N.lib.l3d_remap_create_rectify(N.ptr(K), N.ptr(d) if d.size else None, int(d.size), N.ptr(iR), w, h, device, C.byref(hd)),
                "l3d_remap_create_rectify")
        return cls(hd, (w, h), device)

    @classmethod
    def from_maps(cls, map_x, map_y, device: int = 0) -> "Rectifier":
        mx = np.ascontiguousarray(map_x, np.float32)
        my = np.ascontiguousarray(map_y, np.float32)
        if mx.shape != my.shape or mx.ndim != 2:
            raise ValueError("map_x and map_y must be (H, W) arrays of the same shape")
        h, w = mx.shape
        hd = C.c_void_p()
        N.check(N.lib.l3d_remap_create(N.ptr(mx), N.ptr(my), w, h, device, C.byref(hd)), "l3d_remap_create")
        return cls(hd, (w, h), device)

    def __del__(self):
        try:
            if self._h is not None:
                N.lib.l3d_remap_destroy(self._h)
                self._h = None
        except Exception:
            pass

    @property
    def maps(self) -> Tuple[np.ndarray, np.ndarray]:
        """(map_x, map_y) float32 (H, W), as cv2.initUndistortRectifyMap(..., CV_32FC1) returns them."""
        if self._maps is None:
            w, h = self.size
            mx, my = np.empty((h, w), np.float32), np.empty((h, w), np.float32)
            N.check(N.lib.l3d_remap_fetch_maps(self._h, N.ptr(mx), N.ptr(my)), "l3d_remap_fetch_maps")
            self._maps = (mx, my)
        return self._maps

    def remap(self, src: np.ndarray, want_gray: bool = False):
        """src: (H,W), (H,W,3), (B,H,W,3) uint8, any strides along rows/images -> remapped array of the map's size
        (plus the BGR2GRAY plane with want_gray)."""
        if src.dtype != np.uint8:
            raise TypeError("remap expects uint8 images")
        if src.ndim == 2:
            ch, batched, a = 1, False, src[None, :, :, None]
        elif src.ndim == 3 and src.shape[-1] == 3:
            ch, batched, a = 3, False, src[None]
        elif src.ndim == 4 and src.shape[-1] in (1, 3):
            ch, batched, a = src.shape[-1], True, src
        else:
            raise ValueError(f"unsupported image shape {src.shape}")
        b, sh, sw = a.shape[:3]
        if a.strides[3] != 1 or a.strides[2] != ch or a.strides[1] < sw * ch or (b > 1 and a.strides[0] < a.strides[1] * sh):
            a = np.ascontiguousarray(a)
        w, h = self.size
        dst = np.empty((b, h, w, ch), np.uint8)
        gray = np.empty((b, h, w), np.uint8) if want_gray else None
        N.check(N.lib.l3d_remap_apply(self._h, C.c_void_p(a.ctypes.data), ch, sw, sh, a.strides[1], a.strides[0] if b > 1 else a.strides[1] * sh,
                                      b, N.ptr(dst), N.ptr(gray) if want_gray else None), "l3d_remap_apply")
        out = dst if batched else dst[0]
        if src.ndim == 2:
            out = out[..., 0]
        if want_gray:
            return out, (gray if batched else gray[0])
        return out

    def remap_device(self, src, want_bgr: bool = True, want_gray: bool = False, row_stride: Optional[int] = None,
                     img_stride: Optional[int] = None, src_size: Optional[Tuple[int, int]] = None, offset_bytes: int = 0):
        """torch uint8 CUDA tensor (B,H,W,C) or (B,H,W) -> (remapped or None, gray or None) on the current stream.
        row_stride / img_stride / src_size / offset_bytes address a sub-image (one eye of a combined frame)."""
        import torch

        assert src.is_cuda and src.dtype == torch.uint8 and src.is_contiguous()
        ch = src.shape[-1] if src.dim() == 4 else 1
        b, sh, sw = src.shape[:3]
        if src_size is not None:
            sw, sh = src_size
        rs = row_stride if row_stride is not None else src.shape[2] * ch
        ims = img_stride if img_stride is not None else src.shape[1] * src.shape[2] * ch
        w, h = self.size
        dst = torch.empty((b, h, w, ch) if src.dim() == 4 else (b, h, w), dtype=torch.uint8, device=src.device) if want_bgr else None
        gray = torch.empty((b, h, w), dtype=torch.uint8, device=src.device) if want_gray else None
        stream = C.c_void_p(torch.cuda.current_stream(src.device).cuda_stream)
        N.check(N.lib.l3d_remap_apply_device(self._h, C.c_void_p(src.data_ptr() + offset_bytes), ch, sw, sh, rs, ims, b,
                                             C.c_void_p(dst.data_ptr()) if want_bgr else None,
                                             C.c_void_p(gray.data_ptr()) if want_gray else None, stream), "l3d_remap_apply_device")
        return dst, gray
