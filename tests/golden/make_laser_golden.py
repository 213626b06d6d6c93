"""Generates tests/golden/laser_golden.npz by running the reference's own laser extractors
(core/laser_extractor.py: FastStegerExtractor.extract_centerline :160-261, SimpleLaserExtractor.extract_centerline
:45-100) on small images.  Run once in the build container, where /root/reference exists:

    python tests/golden/make_laser_golden.py

What this pins and what it does not.  The reference module does ``import cv2`` and calls it for the image operators
(cvtColor, GaussianBlur, filter2D, inRange, threshold, morphologyEx, findContours, contourArea, drawContours).  OpenCV
is not installed here, so a stand-in module is registered under that name whose functions call the CPU oracle's
restatements of those operators (oracle/l3d_oracle.c) or, for the trivial ones, plain numpy.  Everything else is the
reference's own Python + NumPy + LAPACK running unmodified: the float32 ``np.linalg.eig`` per bright pixel, the choice
of the eigenvector by ``abs(eigenvalues[0]) >= abs(eigenvalues[1])``, the denominator guard, the accept test
``abs(t*nx) <= 0.5 and abs(t*ny) <= 0.5``, the raster order of the list, the ROI offsets, and for the Simple extractor
the area filter, the mask composition and the per-row ``np.average`` centroid.  The vectors therefore pin those parts
of the oracle and of the GPU path against the reference itself, conditional on the operators behind ``cv2`` -- which
stay unpinned (no OpenCV, no fixtures of it in the reference).

NumPy: the reference declares numpy < 2.  This container has NumPy 2.x, where ``python_int + np.float32`` stays
float32 (NEP 50); under NumPy 1.x the same expression is float64.  The file records the version; the tests compare
to 1e-3 px, far above that difference, and report any point that falls on the other side of an accept test.

``numba`` (imported by the reference, never used) is registered as an empty module with the two imported names.
Only numeric inputs and outputs are written; no reference source is copied.
"""
import contextlib
import importlib.util
import io
import os
import sys

sys.dont_write_bytecode = True   # importing the reference must not leave a bytecode cache beside its sources
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
REF = "/root/reference"
OUT = os.path.join(HERE, "laser_golden.npz")

from oracle import oracle as O  # noqa: E402


from _cv2_standin import make_cv2_standin  # noqa: E402


def load_reference_extractors():
    sys.modules["cv2"] = make_cv2_standin()
    nb = types.ModuleType("numba")
    nb.jit = lambda *a, **k: (lambda f: f)
    nb.prange = range
    sys.modules.setdefault("numba", nb)
    spec = importlib.util.spec_from_file_location("ref_laser_extractor", os.path.join(REF, "core", "laser_extractor.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def steger_images():
    """Small images around the decisions of the per-pixel loop."""
    rng = np.random.default_rng(20261112)
    out = {}
    yy, xx = np.mgrid[0:48, 0:72].astype(np.float64)

    def ridge(cx, s, amp):
        return amp * np.exp(-0.5 * ((xx - cx) / s) ** 2)

    base = 30 + 10 * rng.random((48, 72))
    # a slanted Gaussian ridge
    out["ridge_slanted"] = np.clip(base + ridge(30.3 + 0.21 * yy, 2.0, 225), 0, 255).astype(np.uint8)
    # a perfectly vertical ridge on a flat background: every row is the same, so dxy == 0 exactly
    out["ridge_vertical_dxy0"] = np.clip(40 + ridge(36.4, 2.2, 215), 0, 255).astype(np.uint8)
    # a perfectly horizontal one (dxy == 0, normal along y)
    out["ridge_horizontal_dxy0"] = np.clip(40 + 215 * np.exp(-0.5 * ((yy - 20.6) / 2.2) ** 2), 0, 255).astype(np.uint8)
    # two ridges that cross
    out["ridges_crossing"] = np.clip(base + ridge(20 + 0.6 * yy, 1.8, 220) + ridge(52 - 0.5 * yy, 2.4, 220), 0, 255).astype(np.uint8)
    # near-isotropic blobs: |lambda0| ~ |lambda1| around the centres, and an exactly symmetric one (integer centre)
    blob = np.zeros_like(xx)
    for cx, cy, s in ((18.0, 14.0, 3.0), (45.3, 30.7, 3.5), (60.0, 12.0, 2.5)):
        blob += 230 * np.exp(-0.5 * (((xx - cx) / s) ** 2 + ((yy - cy) / s) ** 2))
    out["blobs_isotropic"] = np.clip(25 + blob, 0, 255).astype(np.uint8)
    # saturated plateau (flat top: tiny derivatives, the denominator guard and large offsets) and bright noise
    plat = np.full((48, 72), 35.0)
    plat[10:30, 20:40] = 255
    plat[34:44, 50:68] = 210 + 45 * rng.random((10, 18))
    out["plateau_and_noise"] = plat.astype(np.uint8)
    # bright smooth noise: a third of the pixels are candidates, with Hessians of every shape and many offsets near 0.5
    f = rng.random((56, 88))
    for _ in range(3):
        f = (f + np.roll(f, 1, 0) + np.roll(f, -1, 0) + np.roll(f, 1, 1) + np.roll(f, -1, 1)) / 5.0
    f = (f - f.min()) / (f.max() - f.min())
    out["bright_noise"] = np.clip(255 * f + 60, 0, 255).astype(np.uint8)
    return out


def main():
    ref = load_reference_extractors()
    store = {"numpy_version": np.array(np.__version__)}
    names = []

    def run_steger(name, img, sigma, thr, roi):
        with contextlib.redirect_stdout(io.StringIO()):
            ex = ref.FastStegerExtractor(sigma=sigma, brightness_threshold=thr)
            pts = ex.extract_centerline(img, roi=roi) if roi is not None else ex.extract_centerline(img)
        pts = np.array([(float(a), float(b)) for a, b in pts], np.float64).reshape(-1, 2)
        store[f"steger/{name}/image"] = img
        store[f"steger/{name}/params"] = np.array([sigma, thr] + (list(roi) if roi is not None else [-1, -1, -1, -1]), np.float64)
        store[f"steger/{name}/points"] = pts
        names.append(f"steger/{name}")
        print(f"steger/{name}: {len(pts)} points", file=sys.stderr)

    for name, img in steger_images().items():
        run_steger(name, img, 3.0, 200, None)
    imgs = steger_images()
    run_steger("ridge_slanted_sigma1.5", imgs["ridge_slanted"], 1.5, 180, None)
    run_steger("ridges_crossing_roi", imgs["ridges_crossing"], 3.0, 200, (9, 5, 50, 31))
    run_steger("ridge_slanted_roi_past_edge", imgs["ridge_slanted"], 3.0, 200, (25, 30, 100, 100))
    # BGR input: a green laser line over texture (the gray conversion is the stand-in's)
    rng = np.random.default_rng(7)
    tex = rng.integers(20, 120, (40, 64, 3)).astype(np.uint8)
    yy, xx = np.mgrid[0:40, 0:64].astype(np.float64)
    line = np.exp(-0.5 * ((xx - (22.7 + 0.3 * yy)) / 2.0) ** 2)
    bgr = np.clip(tex + line[..., None] * np.array([60.0, 255.0, 60.0]), 0, 255).astype(np.uint8)
    run_steger("bgr_green_line", bgr, 3.0, 200, None)
    run_steger("bgr_green_line_roi", bgr, 3.0, 200, (10, 4, 40, 30))
    # a crop of one of the reference's own camera frames (its brightest region; threshold lowered to reach it)
    real = np.load(os.path.join(HERE, "real_pairs.npz"))["left_bgr"][0]
    g = O.bgr2gray(real)
    thr = int(min(200, max(120, np.percentile(g, 99.0) - 1)))
    ys, xs = np.nonzero(g > thr)
    y0, x0 = max(0, int(ys.mean()) - 24), max(0, int(xs.mean()) - 36)
    run_steger("real_frame_crop", np.ascontiguousarray(real[y0:y0 + 48, x0:x0 + 72]), 3.0, thr, None)

    def run_simple(name, img, lo, hi, thr, min_area):
        with contextlib.redirect_stdout(io.StringIO()):
            ex = ref.SimpleLaserExtractor(hsv_lower=np.array(lo), hsv_upper=np.array(hi), brightness_threshold=thr, min_area=min_area)
            pts = ex.extract_centerline(img)
        pts = np.array([(float(a), float(b)) for a, b in pts], np.float64).reshape(-1, 2)
        store[f"simple/{name}/image"] = img
        store[f"simple/{name}/params"] = np.array(list(lo) + list(hi) + [thr, min_area], np.float64)
        store[f"simple/{name}/points"] = pts
        names.append(f"simple/{name}")
        print(f"simple/{name}: {len(pts)} points", file=sys.stderr)

    run_simple("green_line_defaults", bgr, (40, 50, 100), (80, 255, 255), 100, 50)
    wide = np.exp(-0.5 * ((xx - (22.7 + 0.3 * yy)) / 4.0) ** 2)
    dark = np.clip(tex // 4 + wide[..., None] * np.array([40.0, 255.0, 40.0]), 0, 255).astype(np.uint8)
    run_simple("green_line_config", dark, (50, 100, 180), (70, 255, 255), 140, 50)    # the HSV range config.py passes (config.py:45-46)
    # blobs of several sizes around the area threshold, one with a hole, two in one row, one touching the border
    img = rng.integers(0, 50, (60, 90, 3)).astype(np.uint8)
    green = np.array([60, 255, 60], np.uint8)

    def put(y0, y1, x0, x1):
        img[y0:y1, x0:x1] = green - rng.integers(0, 30, (y1 - y0, x1 - x0, 3)).astype(np.uint8)

    put(5, 20, 8, 22); img[10:14, 12:17] = 10          # a hole: the filled contour covers it
    put(8, 16, 40, 49)                                  # 8 x 9: contour area 56 > 50
    put(30, 37, 30, 38)                                 # 7 x 8: contour area 42, dropped at min_area 50
    put(28, 45, 60, 66); put(33, 36, 66, 80)            # an L shape
    put(50, 60, 0, 12)                                  # touches two borders
    put(52, 58, 70, 71)                                 # a one-pixel-wide line: removed by the opening
    run_simple("blobs_min_area_50", img, (40, 50, 100), (80, 255, 255), 100, 50)
    run_simple("blobs_min_area_10", img, (40, 50, 100), (80, 255, 255), 100, 10)
    run_simple("nothing_found", np.zeros((20, 30, 3), np.uint8), (40, 50, 100), (80, 255, 255), 100, 50)

    store["names"] = np.array(names)
    np.savez_compressed(OUT, **store)
    print("wrote", OUT, os.path.getsize(OUT), "bytes", file=sys.stderr)


if __name__ == "__main__":
    main()
