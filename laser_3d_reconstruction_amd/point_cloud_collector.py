"""PointCloudCollector / RealTimePointCloudCollector with the reference's names and bookkeeping
(point_cloud_collector.py:14-306): per-frame 3-D points and colours accumulate in lists, the statistics and the
automatic saves follow the same counters.

The reference hands the merged cloud to Open3D for voxel downsampling, statistical outlier removal, the viewer and the
file writer.  Open3D is not part of this package: ``save`` writes the merged cloud in the layout Open3D gives the
reference's files (binary little-endian PLY, double coordinates, uchar colours: PointCloudProcessor.save_ply_binary) and,
when asked to downsample or to remove outliers, uses PointCloudProcessor's voxel mean and neighbour statistics -- the
algorithms of the reference's own Open3D-free path (utils/point_cloud.py:31-132), which are not Open3D's; colours are
dropped by those two steps.  ``get_point_cloud`` returns the merged (points, colours) arrays, ``visualize`` is not
mirrored.
"""
from __future__ import annotations

import json
import os
from datetime import datetime
from typing import Optional

import numpy as np

from .utils.point_cloud import PointCloudProcessor


class PointCloudCollector:
    def __init__(self, output_dir: str = "output", device: int = 0, verbose: bool = False):
        self.output_dir = output_dir
        os.makedirs(output_dir, exist_ok=True)
        self.points = []
        self.colors = []
        self.frame_count = 0
        self.total_points = 0
        self.verbose = verbose
        self._processor = PointCloudProcessor(device=device)

    def add_points(self, points_3d: np.ndarray, colors: Optional[np.ndarray] = None):
        """One frame's (N,3) points and optional (N,3) RGB colours in 0..1; green when none are given (:41-67)."""
        if len(points_3d) == 0:
            return
        self.points.append(points_3d)
        self.colors.append(colors if colors is not None else np.ones((len(points_3d), 3)) * [0, 1, 0])
        self.frame_count += 1
        self.total_points += len(points_3d)

    def get_point_cloud(self):
        """(points (N,3), colours (N,3)) of everything collected; two empty arrays when nothing is."""
        if len(self.points) == 0:
            return np.zeros((0, 3)), np.zeros((0, 3))
        return np.vstack(self.points), np.vstack(self.colors)

    def save(self, filename: Optional[str] = None, format: str = 'ply', downsample: bool = True, voxel_size: float = 0.002,
             remove_outliers: bool = True) -> str:
        """Writes `<output_dir>/<filename>.ply` and `<filename>_metadata.json` with the reference's keys (:90-166);
        returns the path, or "" when there is nothing to save or the format is unknown."""
        if len(self.points) == 0 or format not in ('ply', 'pcd'):
            return ""
        points, colors = self.get_point_cloud()
        if downsample and len(points) > 1000:
            points, colors = self._processor.voxel_downsample(points, voxel_size), None
        if remove_outliers and len(points) > 100:
            points, colors = self._processor.statistical_outlier_removal(points, nb_neighbors=20, std_ratio=2.0), None
        if filename is None:
            filename = f"pointcloud_{datetime.now().strftime('%Y%m%d_%H%M%S')}"
        filepath = os.path.join(self.output_dir, f"{filename}.{format}")
        # (the reference's 'pcd' goes through Open3D's writer; here both names get the PLY layout)
        self._processor.save_ply_binary(points, filepath if format == 'ply' else filepath[:-4] + '.ply',
                                        None if colors is None else np.round(np.clip(colors, 0, 1) * 255))
        metadata = {
            'filename': filepath, 'format': format, 'timestamp': datetime.now().isoformat(),
            'total_points': self.total_points, 'frame_count': self.frame_count, 'final_points': len(points),
            'downsample': downsample, 'voxel_size': voxel_size if downsample else None, 'remove_outliers': remove_outliers,
        }
        with open(os.path.join(self.output_dir, f"{filename}_metadata.json"), 'w') as f:
            json.dump(metadata, f, indent=2)
        return filepath

    def clear(self):
        self.points = []
        self.colors = []
        self.frame_count = 0
        self.total_points = 0

    def get_statistics(self) -> dict:
        if len(self.points) == 0:
            return {'frame_count': 0, 'total_points': 0, 'avg_points_per_frame': 0}
        all_points = np.vstack(self.points)
        return {
            'frame_count': self.frame_count, 'total_points': self.total_points, 'unique_points': len(all_points),
            'avg_points_per_frame': self.total_points / self.frame_count if self.frame_count > 0 else 0,
            'bounds_min': all_points.min(axis=0).tolist() if len(all_points) > 0 else [0, 0, 0],
            'bounds_max': all_points.max(axis=0).tolist() if len(all_points) > 0 else [0, 0, 0],
        }

    def export_numpy(self, filename: Optional[str] = None) -> str:
        if len(self.points) == 0:
            return ""
        points, colors = self.get_point_cloud()
        if filename is None:
            filename = f"pointcloud_{datetime.now().strftime('%Y%m%d_%H%M%S')}"
        points_file = os.path.join(self.output_dir, f"{filename}_points.npy")
        np.save(points_file, points)
        np.save(os.path.join(self.output_dir, f"{filename}_colors.npy"), colors)
        return points_file


class RealTimePointCloudCollector(PointCloudCollector):
    """Saves by itself every `auto_save_interval` frames (0: never) and when `max_points` is reached, after which it
    starts again from empty (:250-306)."""

    def __init__(self, output_dir: str = "output", auto_save_interval: int = 100, max_points: int = 100000, device: int = 0,
                 verbose: bool = False):
        super().__init__(output_dir, device=device, verbose=verbose)
        self.auto_save_interval = auto_save_interval
        self.max_points = max_points
        self.frames_since_save = 0
        self.save_count = 0

    def add_points(self, points_3d: np.ndarray, colors: Optional[np.ndarray] = None):
        super().add_points(points_3d, colors)
        self.frames_since_save += 1
        if self.auto_save_interval > 0 and self.frames_since_save >= self.auto_save_interval:
            self.auto_save()
        if self.total_points >= self.max_points:
            self.auto_save()
            self.clear()

    def auto_save(self):
        if len(self.points) == 0:
            return
        self.save_count += 1
        self.save(filename=f"auto_save_{self.save_count}_{datetime.now().strftime('%Y%m%d_%H%M%S')}", downsample=True,
                  remove_outliers=True)
        self.frames_since_save = 0
