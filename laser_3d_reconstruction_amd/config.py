"""Config: the reference's knob names and defaults (config.py:9-99).  Unlike the
reference, where the stereo knobs are only printed, SingleUSBStereoCameraManager
here honours STEREO_NUM_DISPARITIES / STEREO_BLOCK_SIZE / STEREO_USE_WLS_FILTER /
STEREO_LAMBDA / STEREO_SIGMA when a Config is passed to it."""
import numpy as np


class Config:
    # camera
    SINGLE_USB_CAMERA_ID = 1
    CAMERA_WIDTH = 640
    CAMERA_HEIGHT = 240
    CAMERA_FPS = 30
    SPLIT_MODE = 'horizontal'
    STEREO_CALIBRATION_FILE = "stereo_calibration.json"

    # stereo matching
    STEREO_NUM_DISPARITIES = 64
    STEREO_BLOCK_SIZE = 5
    STEREO_USE_WLS_FILTER = True
    STEREO_LAMBDA = 8000.0
    STEREO_SIGMA = 1.5

    # laser extractor
    LASER_EXTRACTOR_TYPE = 'simple'
    SIMPLE_LASER_HSV_LOWER = np.array([50, 100, 180])
    SIMPLE_LASER_HSV_UPPER = np.array([70, 255, 255])
    SIMPLE_LASER_BRIGHTNESS_THRESHOLD = 200
    SIMPLE_LASER_MIN_AREA = 50
    STEGER_SIGMA = 3.0
    STEGER_BRIGHTNESS_THRESHOLD = 200
    STEGER_USE_LUT = True

    # laser plane ax + by + cz + d = 0
    LASER_PLANE_COEFFICIENTS = np.array([0, 0, 1, 0], dtype=np.float64)

    # 3-D reconstruction
    USE_REFRACTION_CORRECTION = False
    WATER_REFRACTION_INDEX = 1.33

    # point cloud
    VOXEL_SIZE = 0.002
    OUTLIER_REMOVAL_NEIGHBORS = 20
    OUTLIER_REMOVAL_STD_RATIO = 2.0
    SAVE_FORMAT = 'ply'

    # system
    DEBUG_MODE = False
    OUTPUT_DIR = 'output'
    AUTO_SAVE_INTERVAL = 60
    MIN_POINT_CLOUD_SIZE = 100
    JETSON_OPTIMIZED = False
    USE_CUDA = False
    NUM_THREADS = 4

    @classmethod
    def to_dict(cls):
        out = {}
        for key in dir(cls):
            if not key.startswith('_') and key.isupper():
                value = getattr(cls, key)
                out[key] = value.tolist() if isinstance(value, np.ndarray) else value
        return out

    @classmethod
    def print_config(cls):
        print("\n" + "=" * 60)
        print("system configuration - single USB stereo camera")
        print("=" * 60)
        for key, value in cls.to_dict().items():
            print(f"  {key}: {value}")
        print("=" * 60 + "\n")
