"""ImprovedLaserReconstructor with the reference's signatures (improved_reconstruction.py:11-186): laser pixels
-> 3-D points straight from the disparity map, on the GPU, plus fix_roi_alignment (:189-227, array slicing)."""
from __future__ import annotations

import ctypes as C
from typing import List, Tuple

import numpy as np

from . import _native as N


class ImprovedLaserReconstructor:
    def __init__(self, Q_matrix: np.ndarray, device: int = 0, verbose: bool = False):
        self.Q = Q_matrix
        # the same four numbers the reference takes from Q (improved_reconstruction.py:26-30)
        self.fx = Q_matrix[2, 3]
        self.baseline = 1.0 / Q_matrix[3, 2]
        self.cx = -Q_matrix[0, 3]
        self.cy = -Q_matrix[1, 3]
        self.device = device
        self._q = np.ascontiguousarray(Q_matrix, np.float64)
        if verbose:
            print(f"ImprovedLaserReconstructor: fx={self.fx:.2f} baseline={self.baseline:.4f} m centre=({self.cx:.2f}, {self.cy:.2f})")

    @staticmethod
    def _prep(laser_points, disparity_map):
        uv = np.ascontiguousarray(np.asarray(laser_points, np.float64).reshape(-1, 2))
        d = np.asarray(disparity_map)
        if d.ndim != 2:
            raise ValueError("disparity_map must be a 2-D array")
        is64 = d.dtype == np.float64
        d = np.ascontiguousarray(d, np.float64 if is64 else np.float32)
        return uv, d, int(is64)

    def _triangulate(self, laser_points, disparity_map, window_size, min_disparity):
        if len(laser_points) == 0:
            return np.array([])
        uv, d, is64 = self._prep(laser_points, disparity_map)
        h, w = d.shape
        xyz = np.empty((len(uv), 3), np.float32)
        n = C.c_int(0)
        N.check(N.lib.l3d_triangulate_disparity(N.ptr(uv), len(uv), N.ptr(d), is64, w, h, N.ptr(self._q), int(window_size),
                                                C.c_double(float(min_disparity)), N.ptr(xyz), C.byref(n), self.device),
                "l3d_triangulate_disparity")
        return xyz[: n.value].copy() if n.value else np.array([])

    def reconstruct_from_disparity(self, laser_points: List[Tuple[float, float]], disparity_map: np.ndarray,
                                   min_disparity: float = 1.0) -> np.ndarray:
        """Nearest-pixel disparity -> (N,3) float32 [X, Y, Z]; np.array([]) when nothing survives."""
        return self._triangulate(laser_points, disparity_map, 0, min_disparity)

    def reconstruct_with_interpolation(self, laser_points: List[Tuple[float, float]], disparity_map: np.ndarray,
                                       window_size: int = 3, min_disparity: float = 1.0) -> np.ndarray:
        """Median of the valid disparities in a window around each laser pixel."""
        # window_size // 2 defines the window: an even size covers the next odd one, as in the reference's slicing
        return self._triangulate(laser_points, disparity_map, 2 * (int(window_size) // 2) + 1, min_disparity)

    def triangulate_device(self, uv, disparity, window_size: int = 3, min_disparity: float = 1.0):
        """Device-resident form of reconstruct_with_interpolation (window_size >= 1) and reconstruct_from_disparity
        (window_size = 0): uv (n,2) float64 and disparity (H,W) float32 CUDA tensors -> (xyz (n,3) float32, keep (n,)
        int32); row i is a point of the host method's result where keep is 1.  Runs on torch's current stream."""
        import torch

        assert uv.is_cuda and uv.dtype == torch.float64 and uv.is_contiguous() and uv.dim() == 2 and uv.shape[1] == 2
        assert disparity.is_cuda and disparity.dtype == torch.float32 and disparity.is_contiguous() and disparity.dim() == 2
        n = uv.shape[0]
        h, w = disparity.shape
        xyz = torch.empty((n, 3), dtype=torch.float32, device=uv.device)
        keep = torch.empty((n,), dtype=torch.int32, device=uv.device)
        if n:
            ws = 2 * (int(window_size) // 2) + 1 if window_size else 0
            N.check(N.lib.l3d_triangulate_disparity_device(C.c_void_p(uv.data_ptr()), n, C.c_void_p(disparity.data_ptr()), w, h,
                                                           N.ptr(self._q), ws, C.c_double(float(min_disparity)),
                                                           C.c_void_p(xyz.data_ptr()), C.c_void_p(keep.data_ptr()),
                                                           C.c_void_p(torch.cuda.current_stream(uv.device).cuda_stream)),
                    "l3d_triangulate_disparity_device")
        return xyz, keep

    def create_laser_depth_map(self, laser_points: List[Tuple[float, float]], disparity_map: np.ndarray,
                               image_shape: Tuple[int, int]) -> np.ndarray:
        h, w = image_shape
        uv, d, is64 = self._prep(laser_points, disparity_map) if len(laser_points) else (np.zeros((0, 2)), np.asarray(disparity_map, np.float32), 0)
        if d.shape != (h, w):
            raise ValueError("create_laser_depth_map: image_shape must equal the disparity map's shape")
        out = np.empty((h, w), np.float32)
        d = np.ascontiguousarray(d)
        N.check(N.lib.l3d_laser_depth_map(N.ptr(uv) if len(uv) else None, len(uv), N.ptr(d), is64, w, h, N.ptr(self._q), N.ptr(out), self.device),
                "l3d_laser_depth_map")
        return out


def fix_roi_alignment(left_rect: np.ndarray, right_rect: np.ndarray, roi_left, roi_right):
    """Crops both rectified images to the intersection of their valid ROIs (improved_reconstruction.py:189-227)."""
    x1, y1 = max(roi_left[0], roi_right[0]), max(roi_left[1], roi_right[1])
    x2 = min(roi_left[0] + roi_left[2], roi_right[0] + roi_right[2])
    y2 = min(roi_left[1] + roi_left[3], roi_right[1] + roi_right[3])
    if x2 - x1 > 0 and y2 - y1 > 0:
        return left_rect[y1:y2, x1:x2], right_rect[y1:y2, x1:x2]
    return left_rect, right_rect
