"""MI355X-native stereo-depth + laser centre-line hot path.

Exports the same five names as the reference package (__init__.py:11-24);
PointCloudProcessor (post-hoc point-cloud IO) is outside the accelerated path
and is not provided.
"""
__version__ = "0.1.0"

from .config import Config
from .core.laser_extractor import FastStegerExtractor, SimpleLaserExtractor
from .core.reconstruction import Reconstructor
from .camera.single_usb_stereo_camera import SingleUSBStereoCameraManager

# the reference's north-star text calls the extractor "StegerExtractor"
StegerExtractor = FastStegerExtractor

__all__ = [
    'SimpleLaserExtractor',
    'FastStegerExtractor',
    'StegerExtractor',
    'Reconstructor',
    'SingleUSBStereoCameraManager',
    'Config',
]
