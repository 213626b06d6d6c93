"""ctypes/numpy front end of the CPU oracle (oracle/l3d_oracle.c).

TEST INFRASTRUCTURE ONLY.  Imported by tests/, by ``__graft_entry__.smoke()``
and by the ``cpu_baseline`` leg of ``bench.py`` -- never by the product
package.  Parity status against OpenCV: *unpinned* (see the header of
l3d_oracle.c); the Reconstructor functions are pinned by tests/golden/.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from dataclasses import dataclass

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "_build", "libl3d_oracle.so")
_lib = None


def build(force: bool = False) -> str:
    """Compile the oracle with gcc (no-op when the library is already there)."""
    if force or not os.path.exists(_LIB_PATH) or (
        os.path.getmtime(_LIB_PATH) < os.path.getmtime(os.path.join(_HERE, "l3d_oracle.c"))
    ):
        subprocess.check_call(["make", "-C", _HERE, "-B" if force else "-s"], stdout=subprocess.DEVNULL)
    return _LIB_PATH


class SgbmParams(C.Structure):
    """Mirror of cv2.StereoSGBM_create's keyword set
    (camera/single_usb_stereo_camera.py:252-274)."""

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


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = C.CDLL(_LIB_PATH)
        _lib.orc_version.restype = C.c_char_p
    return _lib


def _p(a, t):
    return a.ctypes.data_as(C.POINTER(t))


def sgbm_params(minDisparity=0, numDisparities=64, blockSize=5, P1=None, P2=None, disp12MaxDiff=1,
                preFilterCap=63, uniquenessRatio=10, speckleWindowSize=100, speckleRange=32, mode=2):
    """Defaults are the reference's left-matcher constructor arguments."""
    if P1 is None:
        P1 = 8 * 3 * blockSize ** 2
    if P2 is None:
        P2 = 32 * 3 * blockSize ** 2
    return SgbmParams(minDisparity, numDisparities, blockSize, P1, P2, disp12MaxDiff, preFilterCap,
                      uniquenessRatio, speckleWindowSize, speckleRange, mode)


def right_matcher_params(left: SgbmParams) -> SgbmParams:
    """cv2.ximgproc.createRightMatcher(left) for a StereoSGBM left matcher."""
    return SgbmParams(-(left.minDisparity + left.numDisparities) + 1, left.numDisparities, left.blockSize,
                      left.P1, left.P2, 1000000, left.preFilterCap, 0, 0, left.speckleRange, left.mode)


def wls_left_matcher_params(left: SgbmParams) -> SgbmParams:
    """State of the left matcher after cv2.ximgproc.createDisparityWLSFilter(left)
    has run: the factory switches off its LR check, speckle filter and
    uniqueness test."""
    return SgbmParams(left.minDisparity, left.numDisparities, left.blockSize, left.P1, left.P2, 1000000,
                      left.preFilterCap, 0, 0, left.speckleRange, left.mode)


def wls_params(left: SgbmParams, lambda_=8000.0, sigma_color=1.5) -> WlsParams:
    import math

    return WlsParams(lambda_, sigma_color, left.minDisparity, left.numDisparities, 24,
                     int(math.ceil(0.5 * left.blockSize)), 0.001, 3, 0.25)


def bgr2gray(bgr: np.ndarray) -> np.ndarray:
    bgr = np.ascontiguousarray(bgr, dtype=np.uint8)
    h, w, _ = bgr.shape
    out = np.empty((h, w), np.uint8)
    lib().orc_bgr2gray(_p(bgr, C.c_uint8), w, h, C.c_ssize_t(3 * w), _p(out, C.c_uint8))
    return out


def bgr2hsv(bgr: np.ndarray) -> np.ndarray:
    bgr = np.ascontiguousarray(bgr, dtype=np.uint8)
    h, w, _ = bgr.shape
    out = np.empty((h, w, 3), np.uint8)
    lib().orc_bgr2hsv(_p(bgr, C.c_uint8), w, h, C.c_ssize_t(3 * w), _p(out, C.c_uint8))
    return out


def sgbm_geometry(p: SgbmParams, w: int, h: int) -> dict:
    out = (C.c_int * 8)()
    rc = lib().orc_sgbm_geometry(C.byref(p), w, h, out)
    keys = ["minX1", "W1", "stripe_sz", "stripe_overlap", "invalid16", "ftzero", "P1", "P2"]
    d = dict(zip(keys, list(out)))
    d["rc"] = rc
    return d


def sgbm_compute(p: SgbmParams, left: np.ndarray, right: np.ndarray) -> np.ndarray:
    left = np.ascontiguousarray(left, np.uint8)
    right = np.ascontiguousarray(right, np.uint8)
    h, w = left.shape
    disp = np.empty((h, w), np.int16)
    rc = lib().orc_sgbm_compute(C.byref(p), _p(left, C.c_uint8), _p(right, C.c_uint8), w, h, _p(disp, C.c_int16))
    if rc:
        raise RuntimeError(f"orc_sgbm_compute failed: {rc}")
    return disp


@dataclass
class SgbmDebug:
    disp: np.ndarray
    C: np.ndarray
    T: np.ndarray
    S: np.ndarray
    raw: np.ndarray
    median: np.ndarray


def sgbm_compute_debug(p: SgbmParams, left: np.ndarray, right: np.ndarray) -> SgbmDebug:
    left = np.ascontiguousarray(left, np.uint8)
    right = np.ascontiguousarray(right, np.uint8)
    h, w = left.shape
    g = sgbm_geometry(p, w, h)
    w1, d = max(g["W1"], 0), p.numDisparities
    disp = np.empty((h, w), np.int16)
    cv = np.zeros((h, w1, d), np.int16)
    tv = np.zeros((h, w1, d), np.int16)
    sv = np.zeros((h, w1, d), np.int16)
    raw = np.empty((h, w), np.int16)
    med = np.empty((h, w), np.int16)
    rc = lib().orc_sgbm_compute_debug(C.byref(p), _p(left, C.c_uint8), _p(right, C.c_uint8), w, h,
                                      _p(disp, C.c_int16), _p(cv, C.c_int16), _p(tv, C.c_int16),
                                      _p(sv, C.c_int16), _p(raw, C.c_int16), _p(med, C.c_int16))
    if rc:
        raise RuntimeError(f"orc_sgbm_compute_debug failed: {rc}")
    return SgbmDebug(disp, cv, tv, sv, raw, med)


def sgbm_prefilter(p: SgbmParams, left: np.ndarray, right: np.ndarray) -> np.ndarray:
    left = np.ascontiguousarray(left, np.uint8)
    right = np.ascontiguousarray(right, np.uint8)
    h, w = left.shape
    out = np.empty((12, h, w), np.uint8)
    rc = lib().orc_sgbm_prefilter(C.byref(p), _p(left, C.c_uint8), _p(right, C.c_uint8), w, h, _p(out, C.c_uint8))
    if rc:
        raise RuntimeError("orc_sgbm_prefilter failed")
    return out


def median3_s16(a: np.ndarray) -> np.ndarray:
    a = np.ascontiguousarray(a, np.int16)
    h, w = a.shape
    out = np.empty_like(a)
    lib().orc_median3_s16(_p(a, C.c_int16), w, h, _p(out, C.c_int16))
    return out


def filter_speckles(a: np.ndarray, new_val: int, max_size: int, max_diff: int) -> np.ndarray:
    a = np.array(a, dtype=np.int16, order="C", copy=True)
    h, w = a.shape
    lib().orc_filter_speckles(_p(a, C.c_int16), w, h, int(new_val), int(max_size), int(max_diff))
    return a


def wls_filter(p: WlsParams, dl: np.ndarray, dr: np.ndarray, guide: np.ndarray, want_conf=False):
    dl = np.ascontiguousarray(dl, np.int16)
    dr = np.ascontiguousarray(dr, np.int16)
    guide = np.ascontiguousarray(guide, np.uint8)
    h, w = dl.shape
    out = np.empty((h, w), np.int16)
    conf = np.empty((h, w), np.float32) if want_conf else None
    rc = lib().orc_wls_filter(C.byref(p), _p(dl, C.c_int16), _p(dr, C.c_int16), _p(guide, C.c_uint8), w, h,
                              _p(out, C.c_int16), _p(conf, C.c_float) if want_conf else None)
    if rc:
        raise RuntimeError("orc_wls_filter failed")
    return (out, conf) if want_conf else out


def depth_from_disp16(disp16: np.ndarray, Q=None, want_xyz=False):
    disp16 = np.ascontiguousarray(disp16, np.int16)
    h, w = disp16.shape
    depth = np.empty((h, w), np.float32)
    xyz = np.empty((h, w, 3), np.float32) if want_xyz else None
    q = np.ascontiguousarray(Q, np.float64) if Q is not None else None
    lib().orc_depth_from_disp16(_p(disp16, C.c_int16), w, h, _p(q, C.c_double) if q is not None else None,
                                _p(depth, C.c_float), _p(xyz, C.c_float) if want_xyz else None)
    return (depth, xyz) if want_xyz else depth


def gauss_kernel(sigma: float) -> np.ndarray:
    taps = np.empty(512, np.float32)
    n = lib().orc_gauss_kernel(C.c_double(sigma), _p(taps, C.c_float), 512)
    if n < 0:
        raise ValueError("sigma too large")
    return taps[:n].copy()


def steger_extract(img: np.ndarray, sigma=3.0, threshold=200, roi=None, want_deriv=False):
    """Returns an (N,2) float32 array of (x, y) in raster order."""
    img = np.ascontiguousarray(img, np.uint8)
    if img.ndim == 3:
        h, w, ch = img.shape
    else:
        (h, w), ch = img.shape, 1
    cap = w * h
    xy = np.empty((cap, 2), np.float32)
    cnt = C.c_int(0)
    roi_arr = (C.c_int * 4)(*roi) if roi is not None else None
    deriv = None
    if want_deriv:
        if roi is not None:
            rw = max(0, min(w, roi[0] + roi[2]) - min(roi[0], w))
            rh = max(0, min(h, roi[1] + roi[3]) - min(roi[1], h))
        else:
            rw, rh = w, h
        deriv = np.zeros((5, rh, rw), np.float32)
    rc = lib().orc_steger_extract(_p(img, C.c_uint8), ch, w, h, C.c_ssize_t(w * ch), roi_arr, C.c_double(sigma),
                                  int(threshold), _p(xy, C.c_float), cap, C.byref(cnt),
                                  _p(deriv, C.c_float) if want_deriv else None)
    if rc:
        raise RuntimeError(f"orc_steger_extract failed: {rc}")
    pts = xy[: cnt.value].copy()
    return (pts, deriv) if want_deriv else pts


def reconstruct_from_depth(uv, depth: np.ndarray, K) -> np.ndarray:
    uv = np.ascontiguousarray(uv, np.float64).reshape(-1, 2)
    depth = np.ascontiguousarray(depth, np.float32)
    K = np.ascontiguousarray(K, np.float64)
    h, w = depth.shape
    out = np.empty((len(uv), 3), np.float64)
    n = C.c_int(0)
    lib().orc_reconstruct_from_depth(_p(uv, C.c_double), len(uv), _p(depth, C.c_float), w, h, _p(K, C.c_double),
                                     _p(out, C.c_double), C.byref(n))
    return out[: n.value].copy()


def reconstruct_plane(uv, K, plane, refraction=True, n_water=1.33) -> np.ndarray:
    uv = np.ascontiguousarray(uv, np.float64).reshape(-1, 2)
    kinv = np.ascontiguousarray(np.linalg.inv(np.asarray(K, np.float64)))
    plane = np.ascontiguousarray(plane, np.float64)
    out = np.empty((len(uv), 3), np.float64)
    lib().orc_reconstruct_plane(_p(uv, C.c_double), len(uv), _p(kinv, C.c_double), _p(plane, C.c_double),
                                int(bool(refraction)), C.c_double(n_water), _p(out, C.c_double))
    return out


def simple_extract(bgr: np.ndarray, hsv_lower=(40, 50, 100), hsv_upper=(80, 255, 255), brightness_threshold=100,
                   min_area=50, want_masks=False):
    """SimpleLaserExtractor.extract_centerline: (N,2) float64 array of (x, y), one row per image row."""
    bgr = np.ascontiguousarray(bgr, np.uint8)
    h, w, _ = bgr.shape
    lo = (C.c_int * 3)(*[int(v) for v in hsv_lower])
    hi = (C.c_int * 3)(*[int(v) for v in hsv_upper])
    xy = np.empty((h, 2), np.float64)
    n = C.c_int(0)
    masks = np.zeros((3, h, w), np.uint8) if want_masks else None
    rc = lib().orc_simple_extract(_p(bgr, C.c_uint8), w, h, C.c_ssize_t(3 * w), lo, hi, int(brightness_threshold),
                                  C.c_double(float(min_area)), _p(xy, C.c_double), C.byref(n),
                                  _p(masks, C.c_uint8) if want_masks else None)
    if rc:
        raise RuntimeError("orc_simple_extract failed")
    pts = xy[: n.value].copy()
    return (pts, masks) if want_masks else pts


def dist12(dist) -> np.ndarray:
    """Distortion coefficients in OpenCV order (k1 k2 p1 p2 [k3 [k4 k5 k6 [s1 s2 s3 s4]]]) padded to 12."""
    d = np.asarray(dist, np.float64).reshape(-1)
    if d.size not in (4, 5, 8, 12):
        raise ValueError(f"{d.size} distortion coefficients")
    out = np.zeros(12)
    out[: d.size] = d
    return out


def init_undistort_rectify_map(K, dist, R, P, size):
    """cv2.initUndistortRectifyMap(K, dist, R, P, (w, h), CV_32FC1) -> (map_x, map_y) float32 (h, w)."""
    w, h = size
    K = np.ascontiguousarray(K, np.float64).reshape(3, 3)
    iR = np.ascontiguousarray(np.linalg.inv(np.asarray(P, np.float64)[:3, :3] @ np.asarray(R, np.float64).reshape(3, 3)))
    d = dist12(dist)
    mx = np.empty((h, w), np.float32)
    my = np.empty((h, w), np.float32)
    lib().orc_init_undistort_rectify_map(_p(K, C.c_double), _p(d, C.c_double), _p(iR, C.c_double), w, h,
                                         _p(mx, C.c_float), _p(my, C.c_float))
    return mx, my


def remap_bilinear(src: np.ndarray, map_x: np.ndarray, map_y: np.ndarray) -> np.ndarray:
    """cv2.remap(src, map_x, map_y, cv2.INTER_LINEAR) for uint8 (H,W) or (H,W,C), constant border 0."""
    src = np.ascontiguousarray(src, np.uint8)
    ch = 1 if src.ndim == 2 else src.shape[2]
    sh, sw = src.shape[:2]
    mx = np.ascontiguousarray(map_x, np.float32)
    my = np.ascontiguousarray(map_y, np.float32)
    dh, dw = mx.shape
    dst = np.empty((dh, dw) + (() if src.ndim == 2 else (ch,)), np.uint8)
    lib().orc_remap_bilinear_u8(_p(src, C.c_uint8), sw, sh, ch, _p(mx, C.c_float), _p(my, C.c_float), dw, dh,
                                _p(dst, C.c_uint8))
    return dst


def steger_improved(img: np.ndarray, sigma: float = 3.0, threshold: int = 200, response_threshold: float = 0.5,
                    per_row: bool = False) -> np.ndarray:
    """ImprovedStegerExtractor.extract_centerline (per_row=False) / extract_centerline_optimized (per_row=True)
    (improved_steger.py:39-223) -> (N,2) float64 (x, y)."""
    img = np.ascontiguousarray(img, np.uint8)
    ch = 3 if img.ndim == 3 else 1
    h, w = img.shape[:2]
    xy = np.empty((w * h, 2), np.float64)
    n = C.c_int(0)
    rc = lib().orc_steger_improved(_p(img, C.c_uint8), ch, w, h, C.c_double(sigma), int(threshold), C.c_double(response_threshold),
                                   int(per_row), _p(xy, C.c_double), w * h, C.byref(n))
    assert rc == 0
    return xy[: n.value].copy()


def hybrid_extract(bgr: np.ndarray, hsv_lower=(50, 100, 180), hsv_upper=(70, 255, 255), threshold: int = 200, sigma: float = 2.0,
                   want_mask: bool = False):
    """HybridLaserExtractor.extract_centerline (improved_steger.py:250-344) -> (N,2) float64 (x, y)."""
    bgr = np.ascontiguousarray(bgr, np.uint8)
    h, w = bgr.shape[:2]
    xy = np.empty((h, 2), np.float64)
    n = C.c_int(0)
    lo = (C.c_int * 3)(*[int(v) for v in hsv_lower])
    hi = (C.c_int * 3)(*[int(v) for v in hsv_upper])
    mask = np.empty((h, w), np.uint8) if want_mask else None
    rc = lib().orc_hybrid_extract(_p(bgr, C.c_uint8), w, h, lo, hi, int(threshold), C.c_double(sigma), _p(xy, C.c_double), h, C.byref(n),
                                  _p(mask, C.c_uint8) if want_mask else None)
    assert rc == 0
    pts = xy[: n.value].copy()
    return (pts, mask) if want_mask else pts


def triangulate_disparity(uv, disp: np.ndarray, Q, window_size: int = 0, min_disparity: float = 1.0) -> np.ndarray:
    """ImprovedLaserReconstructor.reconstruct_from_disparity (window_size=0) / reconstruct_with_interpolation
    (improved_reconstruction.py:37-152) -> (M,3) float32."""
    uv = np.ascontiguousarray(uv, np.float64).reshape(-1, 2)
    is64 = disp.dtype == np.float64
    disp = np.ascontiguousarray(disp, np.float64 if is64 else np.float32)
    h, w = disp.shape
    q = np.ascontiguousarray(Q, np.float64)
    xyz = np.empty((max(len(uv), 1), 3), np.float32)
    n = C.c_int(0)
    lib().orc_triangulate_disparity(_p(uv, C.c_double), len(uv), disp.ctypes.data_as(C.c_void_p), int(is64), w, h, _p(q, C.c_double),
                                    int(window_size), C.c_double(min_disparity), _p(xyz, C.c_float), C.byref(n))
    return xyz[: n.value].copy()


def laser_depth_map(uv, disp: np.ndarray, Q) -> np.ndarray:
    """ImprovedLaserReconstructor.create_laser_depth_map (improved_reconstruction.py:154-186) with image_shape = disp.shape."""
    uv = np.ascontiguousarray(uv, np.float64).reshape(-1, 2)
    is64 = disp.dtype == np.float64
    disp = np.ascontiguousarray(disp, np.float64 if is64 else np.float32)
    h, w = disp.shape
    q = np.ascontiguousarray(Q, np.float64)
    out = np.empty((h, w), np.float32)
    lib().orc_laser_depth_map(_p(uv, C.c_double), len(uv), disp.ctypes.data_as(C.c_void_p), int(is64), w, h, _p(q, C.c_double), _p(out, C.c_float))
    return out


def voxel_downsample(points: np.ndarray, voxel_size: float) -> np.ndarray:
    """PointCloudProcessor._simple_voxel_downsample (utils/point_cloud.py:55-79)."""
    p = np.ascontiguousarray(points, np.float64).reshape(-1, 3)
    out = np.empty((max(len(p), 1), 3), np.float64)
    n = C.c_int(0)
    lib().orc_voxel_downsample(_p(p, C.c_double), len(p), C.c_double(voxel_size), _p(out, C.c_double), C.byref(n))
    return out[: n.value].copy()


def sor_simple(points: np.ndarray, nb_neighbors: int = 20, std_ratio: float = 2.0) -> np.ndarray:
    """PointCloudProcessor._simple_outlier_removal behind statistical_outlier_removal (utils/point_cloud.py:81-132)."""
    p = np.ascontiguousarray(points, np.float64).reshape(-1, 3)
    out = np.empty((max(len(p), 1), 3), np.float64)
    n = C.c_int(0)
    lib().orc_sor_simple(_p(p, C.c_double), len(p), int(nb_neighbors), C.c_double(std_ratio), _p(out, C.c_double), C.byref(n))
    return out[: n.value].copy()


def avg_point_spacing(points: np.ndarray) -> float:
    p = np.ascontiguousarray(points, np.float64).reshape(-1, 3)
    f = lib().orc_avg_point_spacing
    f.restype = C.c_double
    return float(f(_p(p, C.c_double), len(p)))
