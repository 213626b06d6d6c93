"""ImprovedStegerExtractor and HybridLaserExtractor with the reference's names and
call signatures (improved_steger.py:12-345), running on the GPU.

``extract_centerline`` returns ``[(x, y), ...]`` of Python floats like the reference;
``extract_points`` returns the same points as an (N, 2) float32 array.
"""
from __future__ import annotations

import ctypes as C
from typing import List, Tuple

import numpy as np

from . import _native as N


def _as_images(image: np.ndarray):
    img = np.ascontiguousarray(image)
    if img.dtype != np.uint8:
        raise TypeError("expected a uint8 image")
    if img.ndim == 2:
        return img, 1
    if img.ndim == 3 and img.shape[2] == 3:
        return img, 3
    raise ValueError(f"unsupported image shape {img.shape}")


class _StegerHandle:
    """Lazily (re)built l3d_steger handle of the improved kind."""

    def __init__(self, sigma, threshold, response_threshold, device):
        self.args = (float(sigma), int(threshold), float(response_threshold))
        self.device = device
        self.h = None
        self.cap = (0, 0)

    def get(self, w, h, args=None):
        """`args` = the owner's current (sigma, threshold, response_threshold): the reference reads its attributes on
        every call, so a changed value rebuilds the handle."""
        mw, mh = self.cap
        if args is not None:
            args = (float(args[0]), int(args[1]), float(args[2]))
        if self.h is None or w > mw or h > mh or (args is not None and args != self.args):
            self.release()
            if args is not None:
                self.args = args
            self.cap = (max(w, mw), max(h, mh))
            hd = C.c_void_p()
            N.check(N.lib.l3d_steger_create_improved(C.c_double(self.args[0]), self.args[1], C.c_double(self.args[2]), self.cap[0],
                                                     self.cap[1], 1, self.device, C.byref(hd)), "l3d_steger_create_improved")
            self.h = hd
        return self.h

    def release(self):
        if self.h is not None:
            N.lib.l3d_steger_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.release()
        except Exception:
            pass


class ImprovedStegerExtractor:
    """Steger extractor with Sobel-of-Sobel derivatives and a negative-eigenvalue test
    (improved_steger.py:12-223)."""

    def __init__(self, sigma=3.0, brightness_threshold=200, response_threshold=0.5, device: int = 0, verbose: bool = False):
        self.sigma = sigma
        self.brightness_threshold = brightness_threshold
        self.response_threshold = response_threshold
        self.device = device
        self._handle = _StegerHandle(sigma, brightness_threshold, response_threshold, device)
        if verbose:
            print(f"ImprovedStegerExtractor: sigma={sigma} threshold={brightness_threshold} response={response_threshold}")

    def extract_points(self, image: np.ndarray, per_row: bool = False) -> np.ndarray:
        img, ch = _as_images(image)
        h, w = img.shape[:2]
        hd = self._handle.get(w, h, (self.sigma, self.brightness_threshold, self.response_threshold))
        cap = h if per_row else w * h
        xy = np.empty((cap, 2), np.float32)
        n = C.c_int(0)
        N.check(N.lib.l3d_steger_extract_improved(hd, N.ptr(img), ch, w, h, 1, int(per_row), N.ptr(xy), cap, C.byref(n)),
                "l3d_steger_extract_improved")
        return xy[: n.value].copy()

    def extract_device(self, image, per_row: bool = False):
        """torch uint8 CUDA tensor (H,W,3) or (H,W) -> (xy (capacity,2) float32, count (1,) int32) on the device, on
        torch's current stream; the first count rows of xy are the points of extract_points."""
        import torch

        assert image.is_cuda and image.dtype == torch.uint8 and image.is_contiguous()
        assert image.dim() == 2 or (image.dim() == 3 and image.shape[2] == 3)
        ch = 3 if image.dim() == 3 else 1
        h, w = image.shape[:2]
        hd = self._handle.get(w, h, (self.sigma, self.brightness_threshold, self.response_threshold))
        cap = h if per_row else w * h
        xy = torch.empty((cap, 2), dtype=torch.float32, device=image.device)
        count = torch.empty((1,), dtype=torch.int32, device=image.device)
        N.check(N.lib.l3d_steger_extract_improved_device(hd, C.c_void_p(image.data_ptr()), ch, w, h, 1, int(per_row),
                                                         C.c_void_p(xy.data_ptr()), cap, C.c_void_p(count.data_ptr()),
                                                         C.c_void_p(torch.cuda.current_stream(image.device).cuda_stream)),
                "l3d_steger_extract_improved_device")
        return xy, count

    def extract_centerline(self, image: np.ndarray) -> List[Tuple[float, float]]:
        """Every accepted bright pixel, raster order (improved_steger.py:39-126)."""
        return [(float(x), float(y)) for x, y in self.extract_points(image, per_row=False)]

    def extract_centerline_optimized(self, image: np.ndarray) -> List[Tuple[float, float]]:
        """One point per image row: the accepted pixel with the strongest response (improved_steger.py:128-223)."""
        return [(float(x), float(y)) for x, y in self.extract_points(image, per_row=True)]


class HybridLaserExtractor:
    """HSV + brightness mask, then Steger on the masked image, one point per row (improved_steger.py:226-345)."""

    def __init__(self, hsv_lower=np.array([50, 100, 180]), hsv_upper=np.array([70, 255, 255]), brightness_threshold=200, sigma=2.0,
                 device: int = 0, verbose: bool = False):
        self.hsv_lower = hsv_lower
        self.hsv_upper = hsv_upper
        self.brightness_threshold = brightness_threshold
        self.sigma = sigma
        self.device = device
        self._handle = _StegerHandle(sigma, brightness_threshold, 0.5, device)
        self._mask_h = None
        self._mask_cap = (0, 0)
        if verbose:
            print(f"HybridLaserExtractor: HSV {hsv_lower} ~ {hsv_upper}, threshold {brightness_threshold}, sigma {sigma}")

    def __del__(self):
        try:
            if self._mask_h is not None:
                N.lib.l3d_simple_destroy(self._mask_h)
                self._mask_h = None
        except Exception:
            pass

    def _mask_maker(self, w, h):
        mw, mh = self._mask_cap
        if self._mask_h is None or w > mw or h > mh:
            if self._mask_h is not None:
                N.lib.l3d_simple_destroy(self._mask_h)
            self._mask_cap = (max(w, mw), max(h, mh))
            hd = C.c_void_p()
            N.check(N.lib.l3d_simple_create(self._mask_cap[0], self._mask_cap[1], self.device, C.byref(hd)), "l3d_simple_create")
            self._mask_h = hd
        return self._mask_h

    def extract_points(self, image: np.ndarray, want_mask: bool = False):
        img, ch = _as_images(image)
        if ch != 3:
            raise ValueError("HybridLaserExtractor expects a BGR image")
        h, w = img.shape[:2]
        st = self._handle.get(w, h, (self.sigma, self.brightness_threshold, 0.5))
        mk = self._mask_maker(w, h)
        lo = (C.c_int * 3)(*[int(v) for v in np.asarray(self.hsv_lower).reshape(3)])
        hi = (C.c_int * 3)(*[int(v) for v in np.asarray(self.hsv_upper).reshape(3)])
        xy = np.empty((h, 2), np.float32)
        n = C.c_int(0)
        mask = np.empty((h, w), np.uint8) if want_mask else None
        N.check(N.lib.l3d_hybrid_extract(mk, st, N.ptr(img), w, h, lo, hi, int(self.brightness_threshold), N.ptr(xy), h, C.byref(n),
                                         N.ptr(mask) if want_mask else None), "l3d_hybrid_extract")
        pts = xy[: n.value].copy()
        return (pts, mask) if want_mask else pts

    def extract_device(self, image):
        """torch uint8 CUDA tensor (H,W,3) -> (xy (H,2) float32, count (1,) int32) on the device, on torch's current
        stream; the first count rows of xy are the points of extract_points."""
        import torch

        assert image.is_cuda and image.dtype == torch.uint8 and image.is_contiguous() and image.dim() == 3 and image.shape[2] == 3
        h, w = image.shape[:2]
        st = self._handle.get(w, h, (self.sigma, self.brightness_threshold, 0.5))
        mk = self._mask_maker(w, h)
        lo = (C.c_int * 3)(*[int(v) for v in np.asarray(self.hsv_lower).reshape(3)])
        hi = (C.c_int * 3)(*[int(v) for v in np.asarray(self.hsv_upper).reshape(3)])
        xy = torch.empty((h, 2), dtype=torch.float32, device=image.device)
        count = torch.empty((1,), dtype=torch.int32, device=image.device)
        N.check(N.lib.l3d_hybrid_extract_device(mk, st, C.c_void_p(image.data_ptr()), w, h, lo, hi, int(self.brightness_threshold),
                                                C.c_void_p(xy.data_ptr()), h, C.c_void_p(count.data_ptr()),
                                                C.c_void_p(torch.cuda.current_stream(image.device).cuda_stream)),
                "l3d_hybrid_extract_device")
        return xy, count

    def extract_centerline(self, image: np.ndarray) -> List[Tuple[float, float]]:
        return [(float(x), float(y)) for x, y in self.extract_points(image)]
