import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from laser_3d_reconstruction_amd import stereo, ops
from laser_3d_reconstruction_amd.utils import synthetic as S
W, H, D, b = 1920, 1080, 128, 11
lb, rb, _ = S.stereo_pair(W, H, D, 11)
dev = torch.device("cuda", 0)
L = torch.from_numpy(np.stack([lb, lb[::-1].copy()])).to(dev)
R = torch.from_numpy(np.stack([rb, rb[::-1].copy()])).to(dev)
gl, gr = ops.bgr2gray_device(L), ops.bgr2gray_device(R)
kw = dict(minDisparity=0, numDisparities=D, blockSize=b, P1=8*3*b*b, P2=32*3*b*b, disp12MaxDiff=1, uniquenessRatio=10, speckleWindowSize=0, speckleRange=32, preFilterCap=63, mode=2)
m = stereo.StereoSGBM_create(**kw, max_batch=2)
d1 = m.compute_device(gl, gr).cpu().numpy()
print("device path valid", (d1 > -16).mean(), flush=True)
a, c = gl[0].cpu().numpy(), gr[0].cpu().numpy()
print(a.flags['C_CONTIGUOUS'], a.shape, a.dtype, a.mean(), flush=True)
m2 = stereo.StereoSGBM_create(**kw)
single = m2.compute(a, c)
raw = m2.fetch_stage(2, 1, W, H)[0]
print("host path: nonzero", (single != 0).mean(), "raw nonzero", (raw != 0).mean(), "equal", np.array_equal(single, d1[0]), flush=True)
single = m2.compute(a, c)
print("host path again: nonzero", (single != 0).mean(), "equal", np.array_equal(single, d1[0]), flush=True)
m3 = stereo.StereoSGBM_create(**kw)
single = m3.compute(a.copy(), c.copy())
print("host path m3: nonzero", (single != 0).mean(), "equal", np.array_equal(single, d1[0]), flush=True)
