"""Development aid: where the time of a single-frame extract_centerline call goes (host side)."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import ctypes as C
from laser_3d_reconstruction_amd import _native as N
from laser_3d_reconstruction_amd.core.laser_extractor import FastStegerExtractor
from laser_3d_reconstruction_amd.utils import synthetic as S

lb, rb, _ = S.stereo_pair(640, 360, 64, 3)
ex = FastStegerExtractor(3.0, 200)
ex.extract_centerline(lb)
def t(f, n=50):
    f(); t0 = time.perf_counter()
    for _ in range(n): r = f()
    return (time.perf_counter() - t0) / n * 1e3, r
ms, pts = t(lambda: ex.extract_points(lb)); print(f"extract_points {ms:.3f} ms")
ms, _ = t(lambda: ex.extract_centerline(lb)); print(f"extract_centerline {ms:.3f} ms")
p = pts[0]
ms, _ = t(lambda: [(float(x), float(y)) for x, y in p]); print(f"list of tuples {ms:.3f} ms for {len(p)} points")
ms, _ = t(lambda: list(map(tuple, p.tolist()))); print(f"tolist variant {ms:.3f} ms")
img = np.ascontiguousarray(lb)[None]
hd = ex._handle(640, 360, 1)
cap = 640 * 360
xy = np.empty((1, cap, 2), np.float32); counts = np.zeros(1, np.int32)
ms, _ = t(lambda: N.lib.l3d_steger_extract(hd, N.ptr(img), 3, 640, 360, 1, None, N.ptr(xy), cap, N.ptr(counts))); print(f"l3d_steger_extract alone {ms:.3f} ms")
ms, _ = t(lambda: np.empty((1, cap, 2), np.float32)); print(f"np.empty {ms:.4f} ms")
