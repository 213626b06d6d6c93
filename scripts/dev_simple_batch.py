"""Throughput of the Simple extractor on a stack of frames (device-resident input): frames per second by batch size."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from laser_3d_reconstruction_amd import SimpleLaserExtractor
from laser_3d_reconstruction_amd.utils import synthetic as S
frames = np.stack([S.stereo_pair(640, 360, 64, 100 + k)[0] for k in range(16)])
for B in (1, 16, 64, 256):
    L = torch.from_numpy(np.concatenate([frames] * max(1, B // 16))[:B].copy()).cuda()
    ex = SimpleLaserExtractor(np.array([50, 100, 180]), np.array([70, 255, 255]), 200, 50)
    for _ in range(3): ex.extract_batch_device(L)
    t0 = time.perf_counter(); n = max(3, 256 // B)
    for _ in range(n): ex.extract_batch_device(L)
    dt = (time.perf_counter() - t0) / n
    print(f"batch {B}: {dt * 1e3:.3f} ms per call, {B / dt:.0f} frames/s", flush=True)
