"""Development aid: Steger extraction time on frames with and without a laser line."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from laser_3d_reconstruction_amd.core.laser_extractor import FastStegerExtractor
from laser_3d_reconstruction_amd.utils import synthetic as S

W, H, D, B = 640, 360, 64, 256
dev = torch.device("cuda", 0)
ex = FastStegerExtractor(3.0, 200)
for laser in (True, False):
    frames = [S.stereo_pair(W, H, D, k, laser=laser)[0] for k in range(8)]
    x = torch.from_numpy(np.stack([frames[i % 8] for i in range(B)])).to(dev)
    ex.extract_device(x, 4 * H)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(20):
        xy, cnt = ex.extract_device(x, 4 * H)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 20
    print(f"laser={laser}: {dt * 1e3:.3f} ms per {B} frames, {B * W * H / dt / 1e9:.0f} Gpix/s, points/frame {float(cnt.float().mean()):.0f}")
