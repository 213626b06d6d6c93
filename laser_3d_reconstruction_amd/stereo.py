"""Stereo matcher / disparity-filter objects with the cv2 call surface the
reference uses (camera/single_usb_stereo_camera.py:252-285, 324-332;
test_depth.py:68; test_improved_laser.py:151):

    matcher = StereoSGBM_create(minDisparity=0, numDisparities=64, blockSize=5, ...,
                                mode=STEREO_SGBM_MODE_SGBM_3WAY)
    right   = createRightMatcher(matcher)
    wls     = createDisparityWLSFilter(matcher); wls.setLambda(8000.0); wls.setSigmaColor(1.5)
    dl = matcher.compute(left_gray, right_gray)          # int16, disparity * 16
    dr = right.compute(right_gray, left_gray)
    df = wls.filter(dl, left_gray, disparity_map_right=dr)

Every ``compute`` / ``filter`` runs on the GPU through libl3d_hip.so.  NumPy
arrays in -> NumPy arrays out (host path); ``compute_device`` /
``filter_device`` take and return device pointers wrapped in torch tensors for
callers that keep frames resident in HBM.
"""
from __future__ import annotations

import ctypes as C
import math

import numpy as np

from . import _native as N

STEREO_SGBM_MODE_SGBM = 0
STEREO_SGBM_MODE_HH = 1
STEREO_SGBM_MODE_SGBM_3WAY = 2
STEREO_SGBM_MODE_HH4 = 3
DISP_SHIFT = 4
DISP_SCALE = 1 << DISP_SHIFT


def _torch_stream_ptr(t):
    import torch

    return C.c_void_p(torch.cuda.current_stream(t.device).cuda_stream)


class StereoSGBM:
    """cv2.StereoSGBM look-alike backed by l3d_sgbm_*: MODE_SGBM_3WAY (the reference's mode, the fast path), MODE_SGBM and
    MODE_HH (5 and 8 aggregation paths over the whole image); MODE_HH4 is rejected."""

    def __init__(self, minDisparity=0, numDisparities=16, blockSize=3, P1=0, P2=0, disp12MaxDiff=0,
                 preFilterCap=0, uniquenessRatio=0, speckleWindowSize=0, speckleRange=0,
                 mode=STEREO_SGBM_MODE_SGBM, device=0, max_batch=1):
        self._p = N.SgbmParams(int(minDisparity), int(numDisparities), int(blockSize), int(P1), int(P2),
                               int(disp12MaxDiff), int(preFilterCap), int(uniquenessRatio),
                               int(speckleWindowSize), int(speckleRange), int(mode))
        self.device = int(device)
        self.max_batch = int(max_batch)
        self._h = None
        self._shape = None

    # --- cv2-style accessors (createRightMatcher / createDisparityWLSFilter use them) ---
    def _set(self, name, v):
        setattr(self._p, name, int(v))
        self._release()

    def getMinDisparity(self): return self._p.minDisparity
    def setMinDisparity(self, v): self._set("minDisparity", v)
    def getNumDisparities(self): return self._p.numDisparities
    def setNumDisparities(self, v): self._set("numDisparities", v)
    def getBlockSize(self): return self._p.blockSize
    def setBlockSize(self, v): self._set("blockSize", v)
    def getP1(self): return self._p.P1
    def setP1(self, v): self._set("P1", v)
    def getP2(self): return self._p.P2
    def setP2(self, v): self._set("P2", v)
    def getDisp12MaxDiff(self): return self._p.disp12MaxDiff
    def setDisp12MaxDiff(self, v): self._set("disp12MaxDiff", v)
    def getPreFilterCap(self): return self._p.preFilterCap
    def setPreFilterCap(self, v): self._set("preFilterCap", v)
    def getUniquenessRatio(self): return self._p.uniquenessRatio
    def setUniquenessRatio(self, v): self._set("uniquenessRatio", v)
    def getSpeckleWindowSize(self): return self._p.speckleWindowSize
    def setSpeckleWindowSize(self, v): self._set("speckleWindowSize", v)
    def getSpeckleRange(self): return self._p.speckleRange
    def setSpeckleRange(self, v): self._set("speckleRange", v)
    def getMode(self): return self._p.mode
    def setMode(self, v): self._set("mode", v)

    @property
    def params(self) -> N.SgbmParams:
        return self._p

    def _release(self):
        if self._h is not None:
            N.lib.l3d_sgbm_destroy(self._h)
            self._h = None
            self._shape = None

    def __del__(self):
        try:
            self._release()
        except Exception:
            pass

    def _handle(self, w, h, batch):
        if self._h is None or self._shape != (w, h) or batch > self.max_batch:
            self._release()
            self.max_batch = max(self.max_batch, batch)
            hd = C.c_void_p()
            N.check(N.lib.l3d_sgbm_create(C.byref(self._p), w, h, self.max_batch, self.device, C.byref(hd)),
                    "l3d_sgbm_create")
            self._h = hd
            self._shape = (w, h)
        return self._h

    def geometry(self, w, h) -> dict:
        out = (C.c_int * 8)()
        N.check(N.lib.l3d_sgbm_geometry(C.byref(self._p), w, h, out), "l3d_sgbm_geometry")
        keys = ["minX1", "W1", "stripe_sz", "stripe_overlap", "invalid16", "ftzero", "P1", "P2"]
        return dict(zip(keys, list(out)))

    def compute(self, left: np.ndarray, right: np.ndarray) -> np.ndarray:
        """left/right: (H,W) or (B,H,W) uint8 -> int16 disparity * 16 of the same shape."""
        left = np.asarray(left)
        right = np.asarray(right)
        if left.dtype != np.uint8 or right.dtype != np.uint8:
            raise TypeError("StereoSGBM.compute expects uint8 images")
        if left.shape != right.shape or left.ndim not in (2, 3):
            raise ValueError(f"left/right must have the same (H,W) or (B,H,W) shape, got {left.shape} and {right.shape}")
        left = np.ascontiguousarray(left)
        right = np.ascontiguousarray(right)
        batched = left.ndim == 3
        b = left.shape[0] if batched else 1
        h, w = left.shape[-2:]
        hd = self._handle(w, h, b)
        disp = np.empty((b, h, w), np.int16)
        N.check(N.lib.l3d_sgbm_compute(hd, N.ptr(left), N.ptr(right), b, w, w * h, N.ptr(disp)), "l3d_sgbm_compute")
        return disp if batched else disp[0]

    def compute_depth_device(self, left_bgr, right_bgr, Q=None, disp_out=None, depth_out=None):
        """torch uint8 CUDA tensors (B,H,W,3) -> (int16 disparity, float32 depth) on the current stream: gray
        conversion of both views, the matcher and the disparity -> depth step in one library call (the reference's
        get_frames body, camera/single_usb_stereo_camera.py:319-346)."""
        import torch

        if left_bgr.dim() == 3:
            left_bgr, right_bgr = left_bgr[None], right_bgr[None]
        assert left_bgr.is_cuda and right_bgr.is_cuda and left_bgr.dtype == torch.uint8 and right_bgr.dtype == torch.uint8
        assert left_bgr.is_contiguous() and right_bgr.is_contiguous() and left_bgr.shape == right_bgr.shape and left_bgr.shape[-1] == 3
        b, h, w, _ = left_bgr.shape
        hd = self._handle(w, h, b)
        if disp_out is None:
            disp_out = torch.empty((b, h, w), dtype=torch.int16, device=left_bgr.device)
        if depth_out is None:
            depth_out = torch.empty((b, h, w), dtype=torch.float32, device=left_bgr.device)
        q = np.ascontiguousarray(Q, np.float64).reshape(16) if Q is not None else None
        N.check(N.lib.l3d_sgbm_compute_depth_device(hd, C.c_void_p(left_bgr.data_ptr()), C.c_void_p(right_bgr.data_ptr()), b,
                                                    N.ptr(q) if q is not None else None, C.c_void_p(disp_out.data_ptr()),
                                                    C.c_void_p(depth_out.data_ptr()), _torch_stream_ptr(left_bgr)),
                "l3d_sgbm_compute_depth_device")
        return disp_out, depth_out

    def compute_device(self, left, right, out=None):
        """torch uint8 CUDA tensors (B,H,W) -> torch int16 CUDA tensor; enqueued on the current stream."""
        import torch

        if left.dim() == 2:
            left, right = left[None], right[None]
        assert left.is_cuda and right.is_cuda and left.dtype == torch.uint8 and right.dtype == torch.uint8
        assert left.is_contiguous() and right.is_contiguous() and left.shape == right.shape
        b, h, w = left.shape
        hd = self._handle(w, h, b)
        if out is None:
            out = torch.empty((b, h, w), dtype=torch.int16, device=left.device)
        N.check(N.lib.l3d_sgbm_compute_device(hd, C.c_void_p(left.data_ptr()), C.c_void_p(right.data_ptr()), b,
                                              C.c_void_p(out.data_ptr()), _torch_stream_ptr(left)),
                "l3d_sgbm_compute_device")
        return out

    def profile(self, enable: bool = True):
        """Switch per-kernel HIP-event timing of this matcher's compute calls on/off (clears totals)."""
        if self._h is None:
            raise RuntimeError("profile() needs a compute call first (no handle yet)")
        N.check(N.lib.l3d_sgbm_profile(self._h, int(bool(enable))), "l3d_sgbm_profile")

    def profile_read(self) -> dict:
        """{kernel name: total milliseconds} plus 'calls' since profile(True)."""
        n = N.lib.l3d_sgbm_profile_kernel_count()
        ms = (C.c_double * n)()
        calls = C.c_long(0)
        N.check(N.lib.l3d_sgbm_profile_read(self._h, ms, n, C.byref(calls)), "l3d_sgbm_profile_read")
        out = {N.lib.l3d_sgbm_profile_kernel_name(k).decode(): ms[k] for k in range(n)}
        out["calls"] = calls.value
        return out

    def postfilter(self, raw: np.ndarray) -> np.ndarray:
        """The tail of compute() on its own: 3x3 median, then the speckle filter, of (H,W) or (B,H,W) int16 maps."""
        raw = np.ascontiguousarray(raw)
        if raw.dtype != np.int16 or raw.ndim not in (2, 3):
            raise TypeError("postfilter expects (H,W) or (B,H,W) int16 maps")
        batched = raw.ndim == 3
        b = raw.shape[0] if batched else 1
        h, w = raw.shape[-2:]
        hd = self._handle(w, h, b)
        out = np.empty((b, h, w), np.int16)
        N.check(N.lib.l3d_sgbm_postfilter(hd, N.ptr(raw), b, N.ptr(out)), "l3d_sgbm_postfilter")
        return out if batched else out[0]

    def fetch_stage(self, stage: int, batch: int, w: int, h: int) -> np.ndarray:
        if self._h is None:
            raise RuntimeError("fetch_stage() needs a compute call first (no handle yet)")
        g = self.geometry(w, h)
        if stage in (0, 1):
            arr = np.empty((batch, h, max(g["W1"], 0), self._p.numDisparities), np.int16)
        else:
            arr = np.empty((batch, h, w), np.int16)
        N.check(N.lib.l3d_sgbm_fetch_stage(self._h, stage, N.ptr(arr), arr.nbytes), "l3d_sgbm_fetch_stage")
        return arr


def StereoSGBM_create(minDisparity=0, numDisparities=16, blockSize=3, P1=0, P2=0, disp12MaxDiff=0,
                      preFilterCap=0, uniquenessRatio=0, speckleWindowSize=0, speckleRange=0,
                      mode=STEREO_SGBM_MODE_SGBM, **kw) -> StereoSGBM:
    """Same keyword set and defaults as cv2.StereoSGBM_create."""
    return StereoSGBM(minDisparity, numDisparities, blockSize, P1, P2, disp12MaxDiff, preFilterCap,
                      uniquenessRatio, speckleWindowSize, speckleRange, mode, **kw)


def createRightMatcher(matcher_left: StereoSGBM) -> StereoSGBM:
    """cv2.ximgproc.createRightMatcher for an SGBM left matcher: mirrored disparity
    range, same block/penalties/mode/prefilter, no uniqueness test, no LR check,
    no speckle filter."""
    min_disp = matcher_left.getMinDisparity()
    num_disp = matcher_left.getNumDisparities()
    r = StereoSGBM(-(min_disp + num_disp) + 1, num_disp, matcher_left.getBlockSize(),
                   device=matcher_left.device, max_batch=matcher_left.max_batch)
    r.setUniquenessRatio(0)
    r.setP1(matcher_left.getP1())
    r.setP2(matcher_left.getP2())
    r.setMode(matcher_left.getMode())
    r.setPreFilterCap(matcher_left.getPreFilterCap())
    r.setDisp12MaxDiff(1000000)
    r.setSpeckleWindowSize(0)
    return r


class DisparityWLSFilter:
    """cv2.ximgproc.DisparityWLSFilter look-alike backed by l3d_wls_*."""

    def __init__(self, min_disparity, num_disparities, disc_radius, device=0, max_batch=1):
        self._lambda = 8000.0
        self._sigma = 1.0
        self._lrc = 24
        self._radius = int(disc_radius)
        self._min_disp = int(min_disparity)
        self._num_disp = int(num_disparities)
        self.device = int(device)
        self.max_batch = int(max_batch)
        self._h = None
        self._shape = None
        self._conf_shape = None

    def getLambda(self): return self._lambda
    def setLambda(self, v): self._lambda = float(v); self._release()
    def getSigmaColor(self): return self._sigma
    def setSigmaColor(self, v): self._sigma = float(v); self._release()
    def getLRCthresh(self): return self._lrc
    def setLRCthresh(self, v): self._lrc = int(v); self._release()
    def getDepthDiscontinuityRadius(self): return self._radius
    def setDepthDiscontinuityRadius(self, v): self._radius = int(v); self._release()

    def params(self) -> N.WlsParams:
        return N.WlsParams(self._lambda, self._sigma, self._min_disp, self._num_disp, self._lrc, self._radius,
                           0.001, 3, 0.25)

    def _release(self):
        if self._h is not None:
            N.lib.l3d_wls_destroy(self._h)
            self._h = None
            self._shape = None

    def __del__(self):
        try:
            self._release()
        except Exception:
            pass

    def _handle(self, w, h, batch):
        if self._h is None or self._shape != (w, h) or batch > self.max_batch:
            self._release()
            self.max_batch = max(self.max_batch, batch)
            hd = C.c_void_p()
            p = self.params()
            N.check(N.lib.l3d_wls_create(C.byref(p), w, h, self.max_batch, self.device, C.byref(hd)), "l3d_wls_create")
            self._h = hd
            self._shape = (w, h)
        return self._h

    def filter(self, disparity_map_left, left_view, filtered_disparity_map=None, disparity_map_right=None,
               ROI=None, right_view=None) -> np.ndarray:
        """Same positional/keyword layout as cv2's DisparityFilter.filter."""
        if disparity_map_right is None:
            raise ValueError("DisparityWLSFilter (use_confidence=True) needs disparity_map_right")
        if ROI is not None:
            raise NotImplementedError("user ROI is not supported")
        dl = np.ascontiguousarray(disparity_map_left)
        dr = np.ascontiguousarray(disparity_map_right)
        if dl.dtype != np.int16 or dr.dtype != np.int16:
            raise TypeError("disparity maps must be int16 (disparity * 16)")
        g = np.asarray(left_view)
        if g.dtype != np.uint8:
            raise TypeError("left_view must be uint8")
        if g.ndim == dl.ndim + 1 and g.shape[-1] == 3:
            from .ops import bgr2gray  # 3-channel guide: weights from the gray value

            g = bgr2gray(g, device=self.device)
        g = np.ascontiguousarray(g)
        if dl.shape != dr.shape or g.shape != dl.shape or dl.ndim not in (2, 3):
            raise ValueError("disparity maps and guide must share one (H,W) or (B,H,W) shape")
        batched = dl.ndim == 3
        b = dl.shape[0] if batched else 1
        h, w = dl.shape[-2:]
        hd = self._handle(w, h, b)
        out = np.empty((b, h, w), np.int16)
        N.check(N.lib.l3d_wls_filter(hd, N.ptr(dl), N.ptr(dr), N.ptr(g), b, N.ptr(out)), "l3d_wls_filter")
        self._conf_shape = (b, h, w) if batched else (h, w)
        return out if batched else out[0]

    def filter_device(self, dl, guide, dr, out=None):
        import torch

        if dl.dim() == 2:
            dl, dr, guide = dl[None], dr[None], guide[None]
        assert dl.is_cuda and dl.dtype == torch.int16 and dr.dtype == torch.int16 and guide.dtype == torch.uint8
        assert dl.is_contiguous() and dr.is_contiguous() and guide.is_contiguous()
        b, h, w = dl.shape
        hd = self._handle(w, h, b)
        if out is None:
            out = torch.empty_like(dl)
        N.check(N.lib.l3d_wls_filter_device(hd, C.c_void_p(dl.data_ptr()), C.c_void_p(dr.data_ptr()),
                                            C.c_void_p(guide.data_ptr()), b, C.c_void_p(out.data_ptr()),
                                            _torch_stream_ptr(dl)), "l3d_wls_filter_device")
        self._conf_shape = (b, h, w)
        return out

    def getConfidenceMap(self) -> np.ndarray:
        if self._h is None or self._conf_shape is None:
            raise RuntimeError("getConfidenceMap before filter")
        conf = np.empty(self._conf_shape, np.float32)
        N.check(N.lib.l3d_wls_fetch_confidence(self._h, N.ptr(conf), conf.nbytes), "l3d_wls_fetch_confidence")
        return conf


def createDisparityWLSFilter(matcher_left: StereoSGBM) -> DisparityWLSFilter:
    """cv2.ximgproc.createDisparityWLSFilter for an SGBM matcher.  As in OpenCV,
    the factory reconfigures the matcher it is given: LR check off
    (disp12MaxDiff = 1000000), speckle filter off, uniqueness ratio 0."""
    matcher_left.setDisp12MaxDiff(1000000)
    matcher_left.setSpeckleWindowSize(0)
    matcher_left.setUniquenessRatio(0)
    wls = DisparityWLSFilter(matcher_left.getMinDisparity(), matcher_left.getNumDisparities(),
                             int(math.ceil(0.5 * matcher_left.getBlockSize())),
                             device=matcher_left.device, max_batch=matcher_left.max_batch)
    return wls
