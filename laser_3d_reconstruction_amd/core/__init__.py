from .laser_extractor import SimpleLaserExtractor, FastStegerExtractor
from .reconstruction import Reconstructor

__all__ = ['SimpleLaserExtractor', 'FastStegerExtractor', 'Reconstructor']
