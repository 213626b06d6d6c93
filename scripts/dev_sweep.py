"""Per-kernel SGBM times for a sweep of batch sizes and lane-group widths (development aid).

usage: python scripts/dev_sweep.py [W H D blockSize] ; env L3D_SGBM_LPP is set per run by this script
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from laser_3d_reconstruction_amd import stereo  # noqa: E402
from laser_3d_reconstruction_amd.utils import synthetic as S  # noqa: E402

W, H, D, b = (int(v) for v in (sys.argv[1:5] if len(sys.argv) >= 5 else (640, 360, 64, 5)))
batches = [int(v) for v in os.environ.get("SWEEP_BATCHES", "64,128,256").split(",")]
lpps = [int(v) if v else 0 for v in os.environ.get("SWEEP_LPP", "0").split(",")]  # 0 = library default
dev = torch.device("cuda", 0)
frames = [S.stereo_pair(W, H, D, k, laser=False) for k in range(4)]
gl = np.stack([f[0][:, :, 0] for f in frames])
gr = np.stack([f[1][:, :, 0] for f in frames])
kw = dict(minDisparity=0, numDisparities=D, blockSize=b, P1=8 * 3 * b * b, P2=32 * 3 * b * b, disp12MaxDiff=1,
          uniquenessRatio=10, speckleWindowSize=100, speckleRange=32, preFilterCap=63, mode=2)
ref = None
for lpp in lpps:
    if lpp:
        os.environ["L3D_SGBM_LPPV"] = os.environ["L3D_SGBM_LPPH"] = str(lpp)
    for B in batches:
        reps = (B + 3) // 4
        L = torch.from_numpy(np.concatenate([gl] * reps)[:B]).to(dev)
        R = torch.from_numpy(np.concatenate([gr] * reps)[:B]).to(dev)
        m = stereo.StereoSGBM_create(**kw, max_batch=B)
        out = m.compute_device(L, R)
        torch.cuda.synchronize()
        first = out[:4].cpu().numpy()
        if ref is None:
            ref = first
        same = np.array_equal(ref, first)
        m.profile(True)
        n = 5
        for _ in range(n):
            m.compute_device(L, R, out=out)
        torch.cuda.synchronize()
        p = m.profile_read()
        calls = p.pop("calls")
        tot = sum(p.values()) / calls
        print(f"LPP={lpp} B={B} same={same} total {tot:.3f} ms ({B / tot * 1e3:.0f} pairs/s) " +
              " ".join(f"{k.replace('k_', '')}={v / calls:.3f}" for k, v in p.items()), flush=True)
        del m, L, R, out
        torch.cuda.empty_cache()
