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
        parallel to the plane or the intersection lies behind the camera."""
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
