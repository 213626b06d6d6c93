"""Reconstructor: laser pixels -> 3-D points, with the reference's class and
method signatures (core/reconstruction.py:11-262).  reconstruct_from_depth,
reconstruct_point and reconstruct_laser_line run on the GPU; the three
post-processing helpers (filter_outliers / transform_points /
merge_point_clouds) operate on a few hundred points and stay NumPy."""
from __future__ import annotations

from typing import List, Tuple

import numpy as np

from .. import ops


class Reconstructor:
    def __init__(self, camera_intrinsic: np.ndarray, laser_plane: np.ndarray,
                 use_refraction_correction: bool = True, device: int = 0):
        self.K = np.asarray(camera_intrinsic, dtype=np.float64)
        self.K_inv = np.linalg.inv(self.K)
        self.laser_plane = np.asarray(laser_plane, dtype=np.float64)
        self.use_refraction = use_refraction_correction
        # hard-coded in the reference (core/reconstruction.py:28); Config.WATER_REFRACTION_INDEX is not consulted there
        self.water_refraction_index = 1.33
        self.device = device

    def reconstruct_point(self, u: float, v: float) -> np.ndarray:
        """3-D point of pixel (u, v) on the laser plane; [nan, nan, nan] when the ray is
        parallel to the plane or the intersection lies behind the camera.  One kernel launch and one
        round trip through host memory per call (tens of microseconds): use reconstruct_laser_line
        for a list of pixels, as the reference's own frame loop does."""
        return ops.reconstruct_plane([[u, v]], self.K_inv, self.laser_plane, self.use_refraction,
                                     self.water_refraction_index, self.device)[0]

    def reconstruct_laser_line(self, laser_points: List[Tuple[float, float]]) -> np.ndarray:
        if len(laser_points) == 0:
            return np.array([])
        pts = ops.reconstruct_plane(laser_points, self.K_inv, self.laser_plane, self.use_refraction,
                                    self.water_refraction_index, self.device)
        pts = pts[~np.isnan(pts).any(axis=1)]
        return pts if len(pts) else np.array([])

    def reconstruct_from_depth(self, laser_points: List[Tuple[float, float]], depth_image: np.ndarray) -> np.ndarray:
        """Back-projects laser pixels through the depth map.  As in the reference the depth
        value is divided by 1000 (core/reconstruction.py:173)."""
        if len(laser_points) == 0:
            return np.array([])
        pts = ops.reconstruct_from_depth(laser_points, depth_image, self.K, self.device)
        return pts if len(pts) else np.array([])

    # --- device-resident forms: the extractor's output stays in HBM (torch tensors on the current stream) ---
    def reconstruct_from_depth_device(self, xy, counts, depth):
        """xy (B,cap,2) float32, counts (B,) int32, depth (B,H,W) float32 CUDA tensors ->
        (xyz (B,cap,3) float64, keep (B,cap) int32): row i of frame b is valid where keep is 1, i.e. where
        reconstruct_from_depth would append the point."""
        import ctypes as C
        import torch
        from .. import _native as N

        assert xy.is_cuda and xy.dtype == torch.float32 and xy.is_contiguous() and xy.dim() == 3 and xy.shape[2] == 2
        assert counts.dtype == torch.int32 and depth.dtype == torch.float32 and depth.is_contiguous() and depth.dim() == 3
        b, cap, _ = xy.shape
        assert depth.shape[0] == b and counts.shape[0] == b
        h, w = depth.shape[1:]
        xyz = torch.empty((b, cap, 3), dtype=torch.float64, device=xy.device)
        keep = torch.empty((b, cap), dtype=torch.int32, device=xy.device)
        k = np.ascontiguousarray(self.K, np.float64)
        N.check(N.lib.l3d_reconstruct_from_depth_device(C.c_void_p(xy.data_ptr()), C.c_void_p(counts.data_ptr()), cap, b,
                                                        C.c_void_p(depth.data_ptr()), w, h, N.ptr(k), C.c_void_p(xyz.data_ptr()),
                                                        C.c_void_p(keep.data_ptr()),
                                                        C.c_void_p(torch.cuda.current_stream(xy.device).cuda_stream)),
                "l3d_reconstruct_from_depth_device")
        return xyz, keep

    def reconstruct_from_depth_uv64_device(self, uv, depth):
        """uv (n,2) float64 and depth (H,W) float32 CUDA tensors -> (xyz (n,3) float64, keep (n,) int32): the form for
        laser pixels that are doubles (SimpleLaserExtractor's row centroids) and a depth map that is still on the device."""
        import ctypes as C
        import torch
        from .. import _native as N

        assert uv.is_cuda and uv.dtype == torch.float64 and uv.is_contiguous() and uv.dim() == 2 and uv.shape[1] == 2
        assert depth.is_cuda and depth.dtype == torch.float32 and depth.is_contiguous() and depth.dim() == 2
        n = uv.shape[0]
        h, w = depth.shape
        xyz = torch.empty((n, 3), dtype=torch.float64, device=uv.device)
        keep = torch.empty((n,), dtype=torch.int32, device=uv.device)
        k = np.ascontiguousarray(self.K, np.float64)
        N.check(N.lib.l3d_reconstruct_from_depth_uv64_device(C.c_void_p(uv.data_ptr()), n, C.c_void_p(depth.data_ptr()), w, h, N.ptr(k),
                                                             C.c_void_p(xyz.data_ptr()), C.c_void_p(keep.data_ptr()),
                                                             C.c_void_p(torch.cuda.current_stream(uv.device).cuda_stream)),
                "l3d_reconstruct_from_depth_uv64_device")
        return xyz, keep

    def reconstruct_laser_line_device(self, xy, counts):
        """xy (B,cap,2) float32, counts (B,) int32 CUDA tensors -> xyz (B,cap,3) float64: reconstruct_point of every
        laser pixel (refraction as configured); rows of NaN where the reference drops the point."""
        import ctypes as C
        import torch
        from .. import _native as N

        assert xy.is_cuda and xy.dtype == torch.float32 and xy.is_contiguous() and xy.dim() == 3 and xy.shape[2] == 2
        b, cap, _ = xy.shape
        xyz = torch.empty((b, cap, 3), dtype=torch.float64, device=xy.device)
        kinv = np.ascontiguousarray(self.K_inv, np.float64)
        plane = np.ascontiguousarray(self.laser_plane, np.float64)
        N.check(N.lib.l3d_reconstruct_plane_device(C.c_void_p(xy.data_ptr()), C.c_void_p(counts.data_ptr()), cap, b, N.ptr(kinv),
                                                   N.ptr(plane), int(bool(self.use_refraction)), C.c_double(self.water_refraction_index),
                                                   C.c_void_p(xyz.data_ptr()),
                                                   C.c_void_p(torch.cuda.current_stream(xy.device).cuda_stream)),
                "l3d_reconstruct_plane_device")
        return xyz

    def filter_outliers(self, points_3d: np.ndarray, threshold: float = 0.01) -> np.ndarray:
        """Keeps points whose distance to both sequence neighbours is below threshold
        (core/reconstruction.py:184-219)."""
        if len(points_3d) < 3:
            return points_3d
        p = np.asarray(points_3d)
        d = np.linalg.norm(p[1:] - p[:-1], axis=1)
        keep = np.empty(len(p), bool)
        keep[0] = d[0] < threshold
        keep[-1] = d[-1] < threshold
        keep[1:-1] = (d[:-1] < threshold) & (d[1:] < threshold)
        out = p[keep]
        return out if len(out) else np.array([])

    def transform_points(self, points_3d: np.ndarray, R: np.ndarray, t: np.ndarray) -> np.ndarray:
        if len(points_3d) == 0:
            return points_3d
        return (R @ points_3d.T).T + t

    def merge_point_clouds(self, clouds: List[np.ndarray]) -> np.ndarray:
        if not clouds:
            return np.array([])
        valid = [c for c in clouds if len(c) > 0]
        if not valid:
            return np.array([])
        return np.vstack(valid)
