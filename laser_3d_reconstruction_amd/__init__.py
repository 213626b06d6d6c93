"""MI355X-native stereo-depth + laser centre-line hot path.

Exports the same names as the reference package (__init__.py:11-24).  The
modules the reference keeps at its top level are mirrored as submodules:
``improved_steger`` (ImprovedStegerExtractor, HybridLaserExtractor),
``improved_reconstruction`` (ImprovedLaserReconstructor, fix_roi_alignment),
``camera.rectify`` (stereo_rectify, Rectifier); ``system.LaserReconstructionSystem`` is the frame loop of the
reference's main.py (initialize / process_frame / save_point_cloud), ``system.ImprovedLaserSystem`` the one of
test_improved_laser.py, ``point_cloud_collector`` its collector classes.
"""
__version__ = "0.1.0"

from .config import Config
from .core.laser_extractor import FastStegerExtractor, SimpleLaserExtractor
from .core.reconstruction import Reconstructor
from .camera.single_usb_stereo_camera import SingleUSBStereoCameraManager
from .utils.point_cloud import PointCloudProcessor
from .system import LaserReconstructionSystem

# the reference's north-star text calls the extractor "StegerExtractor"
StegerExtractor = FastStegerExtractor

__all__ = [
    'SimpleLaserExtractor',
    'FastStegerExtractor',
    'StegerExtractor',
    'Reconstructor',
    'SingleUSBStereoCameraManager',
    'PointCloudProcessor',
    'Config',
    'LaserReconstructionSystem',
]
