"""Developer check (not collected by pytest): every HIP stage vs the CPU oracle."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import oracle as O  # noqa: E402
from laser_3d_reconstruction_amd import ops, stereo  # noqa: E402
from laser_3d_reconstruction_amd.core.laser_extractor import FastStegerExtractor  # noqa: E402
from laser_3d_reconstruction_amd.utils import synthetic as S  # noqa: E402

ok = True


def report(name, a, r, tol=0):
    global ok
    a = np.asarray(a); r = np.asarray(r)
    if a.shape != r.shape:
        print(f"{name}: SHAPE {a.shape} vs {r.shape}"); ok = False; return
    if tol == 0:
        bad = np.argwhere(a != r)
        if len(bad):
            ok = False
            print(f"{name}: {len(bad)}/{a.size} mismatches; first {bad[0]} got {a[tuple(bad[0])]} want {r[tuple(bad[0])]}")
        else:
            print(f"{name}: exact ({a.size})")
    else:
        d = np.abs(a.astype(np.float64) - r.astype(np.float64))
        d = d[np.isfinite(d)]
        m = d.max() if d.size else 0.0
        print(f"{name}: max|diff|={m:.3g} (tol {tol})")
        if m > tol or (np.isnan(a) != np.isnan(r)).any():
            ok = False; print("   FAIL")


W, H, D = 640, 360, 64
lb, rb, _ = S.stereo_pair(W, H, D, 3)
gl = ops.bgr2gray(lb); gr = ops.bgr2gray(rb)
report("gray", gl, O.bgr2gray(lb))

# steger
for img, roi, tag in ((lb, None, "bgr"), (gl, None, "gray"), (lb, (200, 40, 300, 250), "roi"), (lb[:67, :130], None, "small")):
    ex = FastStegerExtractor()
    t = time.time(); got = ex.extract_points(img, roi)[0]; dt = time.time() - t
    want = O.steger_extract(img, 3.0, 200, roi)
    print(f"steger {tag}: {len(got)} vs {len(want)} points ({dt*1e3:.1f} ms)")
    if len(got) == len(want):
        report(f"steger {tag} xy", got, want, 1e-3)
        print("   bit-identical:", np.array_equal(got, want))
    else:
        ok = False
bt = FastStegerExtractor().extract_batch([lb, lb[:, ::-1].copy(), S.stereo_pair(W, H, D, 5)[0]])
w2 = [O.steger_extract(x) for x in (lb, lb[:, ::-1].copy(), S.stereo_pair(W, H, D, 5)[0])]
for i, (a, b) in enumerate(zip(bt, w2)):
    report(f"steger batch[{i}]", np.array(a, np.float32).reshape(-1, 2), b, 1e-3)

# full WLS path
left = stereo.StereoSGBM_create(minDisparity=0, numDisparities=D, blockSize=5, P1=600, P2=2400, disp12MaxDiff=1, uniquenessRatio=10,
                                speckleWindowSize=100, speckleRange=32, preFilterCap=63, mode=stereo.STEREO_SGBM_MODE_SGBM_3WAY)
right = stereo.createRightMatcher(left)
wls = stereo.createDisparityWLSFilter(left)
wls.setLambda(8000.0); wls.setSigmaColor(1.5)
dl = left.compute(gl, gr); dr = right.compute(gr, gl)
pl = O.wls_left_matcher_params(O.sgbm_params())
report("dl", dl, O.sgbm_compute(pl, gl, gr))
report("dr", dr, O.sgbm_compute(O.right_matcher_params(O.sgbm_params()), gr, gl))
t = time.time(); df = wls.filter(dl, gl, disparity_map_right=dr); dt = time.time() - t
wf, wconf = O.wls_filter(O.wls_params(O.sgbm_params(), 8000.0, 1.5), dl, dr, gl, want_conf=True)
report("wls conf", wls.getConfidenceMap(), wconf, 1e-4)
report("wls out", df, wf)
print(f"   wls {dt*1e3:.1f} ms; |diff|>0: {(df != wf).sum()} of {df.size}; max {np.abs(df.astype(int)-wf.astype(int)).max()}")

Q = np.array([[1, 0, 0, -319.5], [0, 1, 0, -179.5], [0, 0, 0, 418.05], [0, 0, 15.94, 0]], np.float64)
dep, xyz = ops.depth_from_disparity16(df, Q, want_xyz=True)
od, oxyz = O.depth_from_disp16(df, Q, want_xyz=True)
report("depth", dep, od); 
fin = np.isfinite(oxyz)
report("xyz finite", np.where(fin, xyz, 0), np.where(fin, oxyz, 0))
report("depth uncal", ops.depth_from_disparity16(df, None), O.depth_from_disp16(df, None))

# reconstruct
K = np.array([[209.025, 0, 159.5], [0, 209.72, 119.5], [0, 0, 1.0]])
uv = np.random.default_rng(0).uniform([-5, -5], [330, 250], (500, 2))
depth = np.random.default_rng(1).uniform(-100, 3000, (240, 320)).astype(np.float32)
report("from_depth", ops.reconstruct_from_depth(uv, depth, K), O.reconstruct_from_depth(uv, depth, K), 1e-12)
for refr in (False, True):
    a = ops.reconstruct_plane(uv, np.linalg.inv(K), [0.1, -0.05, 1.0, -0.8], refr)
    b = O.reconstruct_plane(uv, K, [0.1, -0.05, 1.0, -0.8], refr)
    report(f"plane refr={refr}", a, b, 1e-12)
print("ALL OK" if ok else "SOME MISMATCH")
sys.exit(0 if ok else 1)
