"""Generates tests/golden/reconstruction_golden.json by running the reference's
own Reconstructor / ImprovedLaserReconstructor (pure numpy code) on seeded
inputs.  Run once in the build container, where /root/reference exists:

    python tests/golden/make_reconstruction_golden.py

The two reference modules start with ``import cv2`` but never call it on the
paths used here (core/reconstruction.py:8 is an unused import;
improved_reconstruction.py only uses cv2 in visualize_laser_depth), so an empty
module object is registered under that name before importing them.  Only the
numeric inputs and outputs are written; no reference source is copied.
"""
import contextlib
import importlib.util
import io
import json
import os
import sys

sys.dont_write_bytecode = True   # importing the reference must not leave a bytecode cache beside its sources
import types

import numpy as np

REF = "/root/reference"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "reconstruction_golden.json")


def _load(name, path):
    spec = importlib.util.spec_from_file_location(name, path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def main():
    sys.modules.setdefault("cv2", types.ModuleType("cv2"))
    rec = _load("ref_reconstruction", os.path.join(REF, "core", "reconstruction.py"))
    imp = _load("ref_improved_reconstruction", os.path.join(REF, "improved_reconstruction.py"))
    rng = np.random.default_rng(20260128)
    K = np.array([[209.025, 0, 159.5], [0, 209.72, 119.5], [0, 0, 1.0]])
    cases = {"K": K.tolist(), "numpy_version": np.__version__, "plane_cases": [], "depth_cases": [],
             "disparity_cases": []}

    planes = [[0, 0, 1, -0.5], [0.1, -0.05, 1.0, -0.8], [1.0, 0.0, 0.0, -0.2], [0, 0, 1, 0.5], [0.0, 1.0, 0.0, 0.0]]
    uv = np.concatenate([
        rng.uniform([0, 0], [320, 240], size=(24, 2)),
        np.array([[159.5, 119.5], [200.25, 100.5], [0.0, 0.0], [319.0, 239.0], [-20.5, 300.0], [159.5, 0.0]]),
    ])
    for plane in planes:
        for refr in (False, True):
            r = rec.Reconstructor(K, np.array(plane, dtype=np.float64), use_refraction_correction=refr)
            with np.errstate(all="ignore"):
                pts = np.array([r.reconstruct_point(float(u), float(v)) for u, v in uv])
                line = r.reconstruct_laser_line([(float(u), float(v)) for u, v in uv])
            cases["plane_cases"].append({
                "plane": plane, "refraction": refr, "uv": uv.tolist(),
                "points": [[None if np.isnan(c) else float(c) for c in p] for p in pts],
                "laser_line": np.asarray(line, dtype=np.float64).reshape(-1, 3).tolist(),
            })

    # reconstruct_from_depth: depth maps with zeros / negatives, points in and out of bounds
    for seed, (h, w) in enumerate([(24, 32), (17, 29)]):
        depth = rng.uniform(-200.0, 3000.0, size=(h, w)).astype(np.float32)
        depth[rng.uniform(size=(h, w)) < 0.2] = 0.0
        pts = rng.uniform([-3, -3], [w + 3, h + 3], size=(40, 2))
        pts = np.concatenate([pts, [[0.0, 0.0], [w - 1e-9, h - 1e-9], [-0.5, 5.0], [5.0, -0.99], [w, 3.0]]])
        r = rec.Reconstructor(K, np.array([0, 0, 1, 0.0]), use_refraction_correction=False)
        out = r.reconstruct_from_depth([(float(a), float(b)) for a, b in pts], depth)
        cases["depth_cases"].append({"depth": depth.tolist(), "uv": pts.tolist(),
                                     "points": np.asarray(out, np.float64).reshape(-1, 3).tolist()})
    r = rec.Reconstructor(K, np.array([0, 0, 1, 0.0]))
    cases["empty_from_depth_shape"] = list(np.shape(r.reconstruct_from_depth([], np.zeros((4, 4), np.float32))))
    cases["empty_laser_line_shape"] = list(np.shape(r.reconstruct_laser_line([])))

    # disparity-domain triangulation (improved_reconstruction.py:37-186)
    Q = np.array([[1, 0, 0, -159.5], [0, 1, 0, -119.5], [0, 0, 0, 209.02533991591997], [0, 0, -15.941451590053976, 0]])
    Qp = Q.copy()
    Qp[3, 2] = 15.941451590053976
    for q in (Q, Qp):
        with contextlib.redirect_stdout(io.StringIO()):
            ir = imp.ImprovedLaserReconstructor(q)
        h, w = 30, 40
        disp = rng.uniform(-2.0, 40.0, size=(h, w)).astype(np.float32)
        disp[rng.uniform(size=(h, w)) < 0.15] = -1.0
        disp[3, 4] = np.nan
        disp[5, 6] = np.inf
        pts = rng.uniform([-2, -2], [w + 2, h + 2], size=(60, 2))
        pts = np.concatenate([pts, [[4.2, 3.1], [6.0, 5.0], [0.5, 0.5], [w - 1.49, h - 1.49], [2.5, 3.5]]])
        lp = [(float(a), float(b)) for a, b in pts]
        with np.errstate(all="ignore"):
            a = ir.reconstruct_from_disparity(lp, disp)
            b = ir.reconstruct_with_interpolation(lp, disp)
            c = ir.create_laser_depth_map(lp, disp, (h, w))
        cases["disparity_cases"].append({
            "Q": q.tolist(), "disparity": [[None if not np.isfinite(v) else float(v) for v in row] for row in disp],
            "nan_at": [3, 4], "inf_at": [5, 6], "uv": pts.tolist(),
            "nearest": np.asarray(a, np.float64).reshape(-1, 3).tolist(),
            "median3": np.asarray(b, np.float64).reshape(-1, 3).tolist(),
            "laser_depth": np.asarray(c, np.float64).tolist(),
        })
    # more of the same with other windows / thresholds / map dtypes, and half-integer pixel coordinates
    # (Python's round() goes to the even neighbour) on a map with many exact ties for the median
    cases["disparity_cases_ext"] = []
    for k, (win, mind, dt) in enumerate([(3, 1.0, np.float32), (5, 1.0, np.float32), (5, 0.5, np.float64), (1, 2.0, np.float32),
                                         (7, 1.0, np.float32)]):
        with contextlib.redirect_stdout(io.StringIO()):
            ir = imp.ImprovedLaserReconstructor(Qp)
        h, w = 24, 32
        disp = np.round(rng.uniform(-2.0, 30.0, size=(h, w)) * (4 if k % 2 else 16)) / (4 if k % 2 else 16)
        disp[rng.uniform(size=(h, w)) < 0.3] = -1.0
        disp[2, 3] = np.nan
        disp[7, 9] = np.inf
        disp[8, 9] = -np.inf
        disp[10, 10] = 0.05          # a depth beyond 10 m
        disp = disp.astype(dt)
        pts = rng.uniform([-1, -1], [w + 1, h + 1], size=(80, 2))
        grid = np.array([[x + 0.5, y + 0.5] for x in range(0, w, 3) for y in range(0, h, 5)])
        pts = np.concatenate([pts, grid, [[3.0, 2.0], [9.0, 7.0], [9.0, 8.0], [10.0, 10.0], [-0.5, 0.5], [w - 0.5, h - 0.5]]])
        lp = [(float(a), float(b)) for a, b in pts]
        with np.errstate(all="ignore"):
            a = ir.reconstruct_from_disparity(lp, disp, min_disparity=mind)
            b = ir.reconstruct_with_interpolation(lp, disp, window_size=win, min_disparity=mind)
            c = ir.create_laser_depth_map(lp, disp, (h, w))
        enc = lambda v: None if np.isnan(v) else ("inf" if v == np.inf else ("-inf" if v == -np.inf else float(v)))
        cases["disparity_cases_ext"].append({
            "Q": Qp.tolist(), "dtype": np.dtype(dt).name, "window_size": win, "min_disparity": mind,
            "disparity": [[enc(v) for v in row] for row in disp], "uv": pts.tolist(),
            "nearest": np.asarray(a, np.float64).reshape(-1, 3).tolist(),
            "median": np.asarray(b, np.float64).reshape(-1, 3).tolist(),
            "laser_depth": np.asarray(c, np.float64).tolist(),
        })
    with open(OUT, "w") as f:
        json.dump(cases, f)
    print("wrote", OUT, os.path.getsize(OUT), "bytes")


if __name__ == "__main__":
    with contextlib.redirect_stdout(sys.stderr):
        pass
    main()
