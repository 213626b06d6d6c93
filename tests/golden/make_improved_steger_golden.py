"""Generates tests/golden/improved_steger_golden.npz by running the reference's own Steger variants
(improved_steger.py: ImprovedStegerExtractor.extract_centerline :39-126, .extract_centerline_optimized :128-223,
HybridLaserExtractor.extract_centerline :250-344) on small images.  Run once in the build container:

    python tests/golden/make_improved_steger_golden.py

As for make_laser_golden.py: the image operators behind the reference's `cv2` calls (cvtColor, GaussianBlur, Sobel,
inRange, morphologyEx, ...) come from the CPU oracle through tests/golden/_cv2_standin.py, because OpenCV is not
installed; everything else is the reference's own Python + NumPy + LAPACK: the float32 `np.linalg.eig`, the choice of
the eigenvalue by `np.argmax(np.abs(...))`, the negative-eigenvalue test, the denominator guard `< 1e-6`, the
`response_threshold` test, the bounds test, the per-row choice of the strongest response in the optimized / hybrid
forms and the hybrid's mask composition.  The vectors pin those parts against the reference itself, conditional on
the operators, whose parity with OpenCV stays unpinned.  Only numeric inputs and outputs are written.
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
sys.path.insert(0, HERE)
REF = "/root/reference"
OUT = os.path.join(HERE, "improved_steger_golden.npz")

from _cv2_standin import make_cv2_standin  # noqa: E402
from make_laser_golden import steger_images  # noqa: E402


def main():
    sys.modules["cv2"] = make_cv2_standin()
    spec = importlib.util.spec_from_file_location("ref_improved_steger", os.path.join(REF, "improved_steger.py"))
    ref = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(ref)
    store, names = {"numpy_version": np.array(np.__version__)}, []

    def keep(kind, name, img, params, pts):
        pts = np.array([(float(a), float(b)) for a, b in pts], np.float64).reshape(-1, 2)
        key = f"{kind}/{name}"
        store[key + "/image"], store[key + "/params"], store[key + "/points"] = img, np.array(params, np.float64), pts
        names.append(key)
        print(f"{key}: {len(pts)} points", file=sys.stderr)

    imgs = steger_images()
    rng = np.random.default_rng(11)
    tex = rng.integers(20, 120, (40, 64, 3)).astype(np.uint8)
    yy, xx = np.mgrid[0:40, 0:64].astype(np.float64)
    line = np.exp(-0.5 * ((xx - (22.7 + 0.3 * yy)) / 2.5) ** 2)
    bgr = np.clip(tex // 3 + line[..., None] * np.array([50.0, 255.0, 50.0]), 0, 255).astype(np.uint8)
    cases = [("ridge_slanted", imgs["ridge_slanted"], 3.0, 200, 0.5), ("ridges_crossing", imgs["ridges_crossing"], 3.0, 200, 0.5),
             ("blobs_isotropic", imgs["blobs_isotropic"], 3.0, 200, 0.5), ("plateau_and_noise", imgs["plateau_and_noise"], 2.0, 200, 0.5),
             ("bright_noise", imgs["bright_noise"], 3.0, 200, 0.5), ("bright_noise_loose", imgs["bright_noise"], 1.5, 180, 1.0),
             ("ridge_vertical_dxy0", imgs["ridge_vertical_dxy0"], 3.0, 200, 0.25), ("bgr_green_line", bgr, 2.0, 150, 0.5)]
    for name, img, sigma, thr, resp in cases:
        with contextlib.redirect_stdout(io.StringIO()):
            ex = ref.ImprovedStegerExtractor(sigma=sigma, brightness_threshold=thr, response_threshold=resp)
            a = ex.extract_centerline(img)
            b = ex.extract_centerline_optimized(img)
        keep("all_points", name, img, [sigma, thr, resp], a)
        keep("per_row", name, img, [sigma, thr, resp], b)
    # hybrid: HSV / brightness mask, then the per-row Steger inside it
    for name, lo, hi, thr, sigma in [("green_line_config", (50, 100, 180), (70, 255, 255), 140, 2.0),
                                     ("green_line_wide_range", (40, 50, 100), (80, 255, 255), 100, 3.0)]:
        with contextlib.redirect_stdout(io.StringIO()):
            ex = ref.HybridLaserExtractor(hsv_lower=np.array(lo), hsv_upper=np.array(hi), brightness_threshold=thr, sigma=sigma)
            pts = ex.extract_centerline(bgr)
        keep("hybrid", name, bgr, list(lo) + list(hi) + [thr, sigma], pts)
    with contextlib.redirect_stdout(io.StringIO()):
        ex = ref.HybridLaserExtractor()
        pts = ex.extract_centerline(np.zeros((24, 32, 3), np.uint8))
    keep("hybrid", "nothing_found", np.zeros((24, 32, 3), np.uint8), [50, 100, 180, 70, 255, 255, 200, 2.0], pts)
    store["names"] = np.array(names)
    np.savez_compressed(OUT, **store)
    print("wrote", OUT, os.path.getsize(OUT), "bytes", file=sys.stderr)


if __name__ == "__main__":
    main()
