"""Development aid: single-frame latency of the drop-in host entry points (numpy in, numpy out)."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from laser_3d_reconstruction_amd.camera.single_usb_stereo_camera import SingleUSBStereoCameraManager
from laser_3d_reconstruction_amd.core.laser_extractor import FastStegerExtractor
from laser_3d_reconstruction_amd.config import Config
from laser_3d_reconstruction_amd.utils import synthetic as S

W, H, D = 640, 360, 64
cfg = Config()
cfg.STEREO_NUM_DISPARITIES, cfg.STEREO_BLOCK_SIZE = D, 5
for wls in (False, True):
    cfg.STEREO_USE_WLS_FILTER = wls
    m = SingleUSBStereoCameraManager(width=2 * W, height=H, config=cfg, calibration_file="/nonexistent.json")
    m.initialize_offline()
    lb, rb, _ = S.stereo_pair(W, H, D, 3)
    m.compute_depth(lb, rb)
    t0 = time.perf_counter()
    n = 30
    for _ in range(n):
        d = m.compute_depth(lb, rb)
    dt = (time.perf_counter() - t0) / n
    print(f"compute_depth wls={wls}: {dt * 1e3:.2f} ms per pair ({1 / dt:.0f} pairs/s), valid {float((d > 0).mean()):.2f}")
ex = FastStegerExtractor(3.0, 200)
ex.extract_centerline(lb)
t0 = time.perf_counter()
for _ in range(30):
    pts = ex.extract_centerline(lb)
print(f"extract_centerline: {(time.perf_counter() - t0) / 30 * 1e3:.2f} ms, {len(pts)} points")
