"""Laser centre-line extractors with the reference's class names and call
signatures (core/laser_extractor.py:14-285), running on the GPU."""
from __future__ import annotations

import ctypes as C
from typing import List, Optional, Tuple

import numpy as np

from .. import _native as N


def _as_tuples(pts: np.ndarray) -> List[Tuple[float, float]]:
    """(N,2) array -> list of (x, y) Python floats (the float32 values widened exactly)."""
    return list(map(tuple, np.asarray(pts, np.float64).tolist()))


class SimpleLaserExtractor:
    """HSV + brightness threshold extractor (core/laser_extractor.py:14-100):
    ``extract_centerline(image BGR uint8) -> [(x, y), ...]``, one intensity-weighted
    centre per image row that holds laser pixels, ascending y; ``[]`` when no
    contour survives the area filter."""

    def __init__(self, hsv_lower=None, hsv_upper=None, brightness_threshold=100, min_area=50, device: int = 0,
                 verbose: bool = False):
        self.hsv_lower = np.asarray(hsv_lower if hsv_lower is not None else [40, 50, 100])
        self.hsv_upper = np.asarray(hsv_upper if hsv_upper is not None else [80, 255, 255])
        self.brightness_threshold = brightness_threshold
        self.min_area = min_area
        self.device = device
        self.verbose = verbose
        self._h = None
        self._cap = (0, 0, 0)
        if verbose:
            print(f"SimpleLaserExtractor: HSV {self.hsv_lower} ~ {self.hsv_upper}, threshold {brightness_threshold}, "
                  f"min area {min_area}")

    def _release(self):
        if self._h is not None:
            N.lib.l3d_simple_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self._release()
        except Exception:
            pass

    def _handle(self, w, h, b=1):
        mw, mh, mb = self._cap
        if self._h is None or w > mw or h > mh or b > mb:
            self._release()
            self._cap = (max(w, mw), max(h, mh), max(b, mb, 1))
            hd = C.c_void_p()
            N.check(N.lib.l3d_simple_create_batch(self._cap[0], self._cap[1], self._cap[2], self.device, C.byref(hd)),
                    "l3d_simple_create_batch")
            self._h = hd
        return self._h

    def _range(self):
        lo = (C.c_int * 3)(*[int(v) for v in np.asarray(self.hsv_lower).reshape(3)])
        hi = (C.c_int * 3)(*[int(v) for v in np.asarray(self.hsv_upper).reshape(3)])
        return lo, hi

    def extract_points(self, image: np.ndarray, want_masks: bool = False):
        """(N,2) float64 array of (x, y); with want_masks also the three stage masks (3,H,W) uint8."""
        img = np.ascontiguousarray(image)
        if img.dtype != np.uint8 or img.ndim != 3 or img.shape[2] != 3:
            raise ValueError("SimpleLaserExtractor expects a BGR uint8 image of shape (H, W, 3)")
        h, w = img.shape[:2]
        hd = self._handle(w, h)
        lo, hi = self._range()
        xy = np.empty((h, 2), np.float64)
        n = C.c_int(0)
        masks = np.empty((3, h, w), np.uint8) if want_masks else None
        N.check(N.lib.l3d_simple_extract(hd, N.ptr(img), w, h, lo, hi, int(self.brightness_threshold),
                                         C.c_double(float(self.min_area)), N.ptr(xy), C.byref(n),
                                         N.ptr(masks) if want_masks else None), "l3d_simple_extract")
        pts = xy[: n.value].copy()
        return (pts, masks) if want_masks else pts

    def extract_centerline(self, image: np.ndarray) -> List[Tuple[float, float]]:
        return _as_tuples(self.extract_points(image))

    def extract_batch(self, images) -> List[List[Tuple[float, float]]]:
        """A list (or (B,H,W,3) array) of BGR images of one shape -> one centre list per image, in one sequence of
        launches; images of different shapes are processed one by one."""
        if len(images) == 0:
            return []
        if len({np.asarray(im).shape for im in images}) != 1:
            return [self.extract_centerline(im) for im in images]
        img = np.ascontiguousarray(np.stack([np.asarray(im) for im in images]))
        if img.dtype != np.uint8 or img.ndim != 4 or img.shape[3] != 3:
            raise ValueError("SimpleLaserExtractor expects BGR uint8 images of shape (H, W, 3)")
        b, h, w = img.shape[:3]
        hd = self._handle(w, h, b)
        lo, hi = self._range()
        xy = np.empty((b, h, 2), np.float64)
        counts = np.zeros(b, np.int32)
        N.check(N.lib.l3d_simple_extract_batch(hd, N.ptr(img), w, h, b, lo, hi, int(self.brightness_threshold),
                                               C.c_double(float(self.min_area)), N.ptr(xy), N.ptr(counts)), "l3d_simple_extract_batch")
        return [_as_tuples(xy[f, : counts[f]]) for f in range(b)]

    def extract_batch_device(self, images) -> List[List[Tuple[float, float]]]:
        """torch uint8 CUDA tensor (B,H,W,3) -> one centre list per image; runs on torch's current stream."""
        import torch

        assert images.is_cuda and images.dtype == torch.uint8 and images.is_contiguous() and images.dim() == 4 and images.shape[3] == 3
        b, h, w = images.shape[:3]
        hd = self._handle(w, h, b)
        lo, hi = self._range()
        xy = np.empty((b, h, 2), np.float64)
        counts = np.zeros(b, np.int32)
        stream = C.c_void_p(torch.cuda.current_stream(images.device).cuda_stream)
        N.check(N.lib.l3d_simple_extract_batch_device(hd, C.c_void_p(images.data_ptr()), w, h, b, lo, hi, int(self.brightness_threshold),
                                                      C.c_double(float(self.min_area)), N.ptr(xy), N.ptr(counts), stream),
                "l3d_simple_extract_batch_device")
        return [_as_tuples(xy[f, : counts[f]]) for f in range(b)]

    def extract_device(self, image):
        """torch uint8 CUDA tensor (H,W,3) -> list of (x, y); runs on torch's current stream."""
        import torch

        assert image.is_cuda and image.dtype == torch.uint8 and image.is_contiguous() and image.dim() == 3
        h, w = image.shape[:2]
        hd = self._handle(w, h)
        lo, hi = self._range()
        xy = np.empty((h, 2), np.float64)
        n = C.c_int(0)
        stream = C.c_void_p(torch.cuda.current_stream(image.device).cuda_stream)
        N.check(N.lib.l3d_simple_extract_device(hd, C.c_void_p(image.data_ptr()), w, h, lo, hi,
                                                int(self.brightness_threshold), C.c_double(float(self.min_area)),
                                                N.ptr(xy), C.byref(n), stream), "l3d_simple_extract_device")
        return _as_tuples(xy[: n.value])


class FastStegerExtractor:
    """Steger ridge extractor.  ``extract_centerline(image, roi=None)`` returns the
    list of (x, y) sub-pixel centres in raster order, one entry per accepted
    bright pixel, exactly the reference's contract
    (core/laser_extractor.py:160-261)."""

    def __init__(self, sigma: float = 3.0, brightness_threshold: int = 200, use_lut: bool = True,
                 device: int = 0, verbose: bool = False):
        self.sigma = sigma
        self.brightness_threshold = brightness_threshold
        self.use_lut = use_lut          # accepted for signature compatibility; the reference never reads its tables
        self.lut = None
        self.kernel_size = int(2 * np.ceil(3 * sigma) + 1)  # what the reference prints; the blur uses round(8*sigma+1)|1 taps
        self.device = device
        self.verbose = verbose
        self._h = None
        self._cap = (0, 0, 0)
        self._key = None
        if verbose:
            print(f"FastStegerExtractor: sigma={sigma} threshold={brightness_threshold} kernel={self.kernel_size}")

    def _release(self):
        if self._h is not None:
            N.lib.l3d_steger_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self._release()
        except Exception:
            pass

    def _handle(self, w, h, b):
        mw, mh, mb = self._cap
        key = (float(self.sigma), int(self.brightness_threshold))
        # sigma / brightness_threshold are read on every call, as in the reference: a changed value rebuilds the handle
        if self._h is None or w > mw or h > mh or b > mb or key != self._key:
            self._release()
            self._key = key
            self._cap = (max(w, mw), max(h, mh), max(b, mb))
            hd = C.c_void_p()
            N.check(N.lib.l3d_steger_create(C.c_double(self.sigma), int(self.brightness_threshold), self._cap[0],
                                            self._cap[1], self._cap[2], self.device, C.byref(hd)), "l3d_steger_create")
            self._h = hd
        return self._h

    def extract_points(self, images: np.ndarray, roi=None) -> List[np.ndarray]:
        """images: (H,W), (H,W,3), (B,H,W) with channels=1 or (B,H,W,3) uint8 -> list of (N,2) float32 arrays."""
        img = np.ascontiguousarray(images)
        if img.dtype != np.uint8:
            raise TypeError("FastStegerExtractor expects uint8 images")
        if img.ndim == 2:
            img, ch = img[None], 1
        elif img.ndim == 3 and img.shape[-1] == 3:
            img, ch = img[None], 3
        elif img.ndim == 3:
            ch = 1
        elif img.ndim == 4 and img.shape[-1] == 3:
            ch = 3
        else:
            raise ValueError(f"unsupported image shape {img.shape}")
        b, h, w = img.shape[:3]
        hd = self._handle(w, h, b)
        cap = w * h
        xy = np.empty((b, cap, 2), np.float32)
        counts = np.zeros(b, np.int32)
        roi_arr = (C.c_int * 4)(*[int(v) for v in roi]) if roi is not None else None
        N.check(N.lib.l3d_steger_extract(hd, N.ptr(img), ch, w, h, b, roi_arr, N.ptr(xy), cap, N.ptr(counts)),
                "l3d_steger_extract")
        return [xy[i, : counts[i]].copy() for i in range(b)]

    def extract_centerline(self, image: np.ndarray,
                           roi: Optional[Tuple[int, int, int, int]] = None) -> List[Tuple[float, float]]:
        pts = self.extract_points(image, roi)[0]
        if self.verbose:
            print(f"  image {image.shape[1]}x{image.shape[0]}: {len(pts)} centre points")
        return _as_tuples(pts)

    def extract_batch(self, images: List[np.ndarray]) -> List[List[Tuple[float, float]]]:
        """One launch for the whole list when the images share a shape."""
        if len(images) == 0:
            return []
        shapes = {im.shape for im in images}
        if len(shapes) == 1:
            res = self.extract_points(np.stack(images))
        else:
            res = [self.extract_points(im)[0] for im in images]
        return [_as_tuples(p) for p in res]

    def extract_device(self, images, capacity: int, roi=None):
        """torch uint8 CUDA tensor (B,H,W,3) or (B,H,W) -> (xy (B,capacity,2) float32, counts (B,) int32) on device."""
        import torch

        assert images.is_cuda and images.dtype == torch.uint8 and images.is_contiguous()
        ch = 3 if (images.dim() == 4 and images.shape[-1] == 3) else 1
        b, h, w = images.shape[:3]
        hd = self._handle(w, h, b)
        xy = torch.empty((b, capacity, 2), dtype=torch.float32, device=images.device)
        counts = torch.empty((b,), dtype=torch.int32, device=images.device)
        roi_arr = (C.c_int * 4)(*[int(v) for v in roi]) if roi is not None else None
        stream = C.c_void_p(torch.cuda.current_stream(images.device).cuda_stream)
        N.check(N.lib.l3d_steger_extract_device(hd, C.c_void_p(images.data_ptr()), ch, w, h, b, roi_arr,
                                                C.c_void_p(xy.data_ptr()), capacity, C.c_void_p(counts.data_ptr()), stream),
                "l3d_steger_extract_device")
        return xy, counts
