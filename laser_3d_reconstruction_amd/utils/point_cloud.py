"""PointCloudProcessor with the reference's method names and return conventions
(utils/point_cloud.py:12-349), as it behaves without Open3D: voxel-grid downsampling,
the statistical outlier filter and the point-spacing metric run on the GPU; PLY writing is
plain file I/O.  Open3D-only features of the reference (PCD files, normals, the viewer) are
not provided: save_pcd falls back to PLY and estimate_normals returns zeros, exactly as the
reference does when Open3D is missing."""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional

import numpy as np

from .. import _native as N


def _np_pairwise_mean(a: np.ndarray) -> float:
    return float(np.add.reduce(np.ascontiguousarray(a, np.float64)) / len(a))


class PointCloudProcessor:
    def __init__(self, device: int = 0):
        self.try_open3d = False
        self.device = device

    @staticmethod
    def _pts(points):
        return np.ascontiguousarray(points, np.float64).reshape(-1, 3)

    def voxel_downsample(self, points: np.ndarray, voxel_size: float = 0.002) -> np.ndarray:
        if len(points) == 0:
            return points
        p = self._pts(points)
        out = np.empty_like(p)
        n = C.c_int(0)
        N.check(N.lib.l3d_voxel_downsample(N.ptr(p), len(p), C.c_double(float(voxel_size)), N.ptr(out), C.byref(n), self.device),
                "l3d_voxel_downsample")
        return out[: n.value].copy()

    def statistical_outlier_removal(self, points: np.ndarray, nb_neighbors: int = 20, std_ratio: float = 2.0) -> np.ndarray:
        if len(points) < nb_neighbors:
            return points
        p = self._pts(points)
        out = np.empty_like(p)
        n = C.c_int(0)
        N.check(N.lib.l3d_sor_simple(N.ptr(p), len(p), int(nb_neighbors), C.c_double(float(std_ratio)), N.ptr(out), C.byref(n), self.device),
                "l3d_sor_simple")
        # np.array([]) of an empty list in the reference
        return out[: n.value].copy() if n.value else np.array([])

    def save_ply(self, points: np.ndarray, filename: str, colors: Optional[np.ndarray] = None):
        """ASCII PLY, `float` x y z with six decimals (+ uchar colours) -- byte for byte the reference's writer
        (utils/point_cloud.py:134-181)."""
        if len(points) == 0:
            return
        if not filename.endswith('.ply'):
            filename += '.ply'
        os.makedirs(os.path.dirname(filename) if os.path.dirname(filename) else '.', exist_ok=True)
        with_colors = colors is not None and len(colors) == len(points)
        with open(filename, 'w') as f:
            f.write("ply\nformat ascii 1.0\n")
            f.write(f"element vertex {len(points)}\n")
            f.write("property float x\nproperty float y\nproperty float z\n")
            if with_colors:
                f.write("property uchar red\nproperty uchar green\nproperty uchar blue\n")
            f.write("end_header\n")
            for i in range(len(points)):
                f.write(f"{points[i, 0]:.6f} {points[i, 1]:.6f} {points[i, 2]:.6f}")
                if with_colors:
                    c = colors[i]
                    f.write(f" {int(c[0])} {int(c[1])} {int(c[2])}")
                f.write("\n")

    def save_ply_binary(self, points: np.ndarray, filename: str, colors: Optional[np.ndarray] = None):
        """Binary little-endian PLY with double coordinates and uchar colours: the layout of the files the
        reference's collector writes through Open3D (point_cloud_collector.py:137-140; header of
        output/pointcloud_20260115_202445.ply)."""
        if len(points) == 0:
            return
        if not filename.endswith('.ply'):
            filename += '.ply'
        os.makedirs(os.path.dirname(filename) if os.path.dirname(filename) else '.', exist_ok=True)
        p = self._pts(points)
        with_colors = colors is not None and len(colors) == len(p)
        fields = [("x", "<f8"), ("y", "<f8"), ("z", "<f8")] + ([("r", "u1"), ("g", "u1"), ("b", "u1")] if with_colors else [])
        rec = np.empty(len(p), np.dtype(fields))
        rec["x"], rec["y"], rec["z"] = p[:, 0], p[:, 1], p[:, 2]
        if with_colors:
            c = np.asarray(colors)
            rec["r"], rec["g"], rec["b"] = c[:, 0].astype(np.uint8), c[:, 1].astype(np.uint8), c[:, 2].astype(np.uint8)
        with open(filename, 'wb') as f:
            f.write(b"ply\nformat binary_little_endian 1.0\n")
            f.write(f"element vertex {len(p)}\n".encode())
            f.write(b"property double x\nproperty double y\nproperty double z\n")
            if with_colors:
                f.write(b"property uchar red\nproperty uchar green\nproperty uchar blue\n")
            f.write(b"end_header\n")
            f.write(rec.tobytes())

    def save_pcd(self, points: np.ndarray, filename: str, colors: Optional[np.ndarray] = None):
        self.save_ply(points, filename.replace('.pcd', '.ply'), colors)

    def estimate_normals(self, points: np.ndarray, radius: float = 0.01) -> np.ndarray:
        return np.zeros_like(points)

    def compute_point_cloud_metrics(self, points: np.ndarray) -> dict:
        if len(points) == 0:
            return {'num_points': 0, 'bbox': None, 'center': None, 'dimensions': None}
        metrics = {
            'num_points': len(points),
            'center': np.mean(points, axis=0),
            'std': np.std(points, axis=0),
            'min': np.min(points, axis=0),
            'max': np.max(points, axis=0),
        }
        dimensions = metrics['max'] - metrics['min']
        metrics['dimensions'] = dimensions
        metrics['volume'] = np.prod(dimensions)
        if len(points) > 1:
            p = self._pts(points)
            nn = np.empty(len(p), np.float64)
            N.check(N.lib.l3d_nn_distances(N.ptr(p), len(p), N.ptr(nn), self.device), "l3d_nn_distances")
            metrics['avg_point_spacing'] = np.mean(nn)
        return metrics

    def visualize(self, points: np.ndarray, colors: Optional[np.ndarray] = None, window_name: str = "3D Point Cloud"):
        print(f"visualisation needs Open3D; the cloud has {len(points)} points")

    def merge_and_clean(self, point_clouds: list, voxel_size: float = 0.002) -> np.ndarray:
        if not point_clouds:
            return np.array([])
        merged = np.vstack([pc for pc in point_clouds if len(pc) > 0])
        if len(merged) == 0:
            return merged
        merged = self.voxel_downsample(merged, voxel_size)
        merged = self.statistical_outlier_removal(merged)
        return merged
