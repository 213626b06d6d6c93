"""Differential stress of the Steger extractor on the GPU against the CPU restatement (development aid, not part of the test
suite): random image sizes (widths that are not multiples of the 16-pixel marking groups or the 64-pixel tiles), batches, one or
three channels, regions of interest, thresholds, sigmas, and images with lines, blobs, noise or nothing above the threshold.
The same handle serves every case of a (sigma, threshold) pair, so flags and counts left over from one call would show in the next.
usage: python scripts/gpu_stress_steger.py [cases] [seed]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from laser_3d_reconstruction_amd import FastStegerExtractor  # noqa: E402
from oracle import oracle as O  # noqa: E402


def image(rng, W, H, ch, kind):
    img = rng.integers(0, 120, (H, W)).astype(np.float32)
    yy, xx = np.mgrid[0:H, 0:W].astype(np.float32)
    if kind == 0:      # slanted line
        x0, k, w = rng.uniform(0, W), rng.uniform(-0.7, 0.7), rng.uniform(1.0, 5.0)
        img += 230 * np.exp(-((xx - x0 - k * yy) ** 2) / (2 * w * w))
    elif kind == 1:    # horizontal line
        y0, w = rng.uniform(0, H), rng.uniform(1.0, 4.0)
        img += 230 * np.exp(-((yy - y0) ** 2) / (2 * w * w))
    elif kind == 2:    # blobs
        for _ in range(int(rng.integers(1, 6))):
            cx, cy, w = rng.uniform(0, W), rng.uniform(0, H), rng.uniform(1.5, 8.0)
            img += 240 * np.exp(-((xx - cx) ** 2 + (yy - cy) ** 2) / (2 * w * w))
    elif kind == 3:    # bright noise everywhere
        img = rng.integers(150, 256, (H, W)).astype(np.float32)
    # kind 4: nothing above the threshold
    g = np.clip(img, 0, 255).astype(np.uint8)
    if ch == 1:
        return g
    return np.ascontiguousarray(np.stack([g, np.clip(g.astype(int) + rng.integers(-3, 4, g.shape), 0, 255).astype(np.uint8), g], -1))


def write_summary(name, record):
    """Adds this run's result to the JSON file named by L3D_STRESS_SUMMARY (one entry per script and seed)."""
    import json
    path = os.environ.get("L3D_STRESS_SUMMARY")
    if not path:
        return
    data = json.load(open(path)) if os.path.exists(path) else {}
    data.setdefault(name, []).append(record)
    json.dump(data, open(path, "w"), indent=1)


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 200
    rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
    t0, bad, pts = time.time(), 0, 0
    extractors = {}
    for case in range(n):
        sigma, thr = float(rng.choice([3.0, 3.0, 1.5, 2.0])), int(rng.choice([200, 200, 120, 180]))
        ex = extractors.setdefault((sigma, thr), FastStegerExtractor(sigma, thr))
        W, H = int(rng.integers(5, 400)), int(rng.integers(5, 200))
        B, ch = int(rng.integers(1, 6)), int(rng.choice([1, 3]))
        imgs = np.stack([image(rng, W, H, ch, int(rng.integers(0, 5))) for _ in range(B)])
        roi = None
        if rng.random() < 0.3:
            roi = (int(rng.integers(0, W)), int(rng.integers(0, H)), int(rng.integers(1, W + 20)), int(rng.integers(1, H + 20)))
        got = ex.extract_points(imgs, roi)
        for b in range(B):
            want = O.steger_extract(imgs[b], sigma, thr, roi)
            pts += len(want)
            ok = len(got[b]) == len(want) and (len(want) == 0 or np.abs(got[b] - want).max() <= 1e-3)
            if not ok:
                bad += 1
                print(f"MISMATCH case {case} image {b}: W={W} H={H} B={B} ch={ch} roi={roi} sigma={sigma} thr={thr}: {len(got[b])} vs {len(want)} points", flush=True)
        if case % 50 == 49:
            print(f"{case + 1} cases, {pts} points, {bad} mismatches, {time.time() - t0:.0f} s", flush=True)
    print(f"done: {n} cases, {pts} points, {bad} mismatches")
    write_summary("steger_extractor", {"script": "scripts/gpu_stress_steger.py", "seed": int(sys.argv[2]) if len(sys.argv) > 2 else 1, "cases": n,
                                       "points": int(pts), "mismatches": bad, "compared": "point count and coordinates (1e-3 px) against the CPU "
                                       "restatement, every image", "seconds": round(time.time() - t0)})
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
