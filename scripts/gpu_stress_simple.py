"""Differential stress of the Simple (HSV + brightness) extractor and of the hybrid extractor's mask stage on the GPU
against the CPU restatement (development aid, not part of the test suite): random image sizes (widths on both sides of
the 64-pixel groups the row-run labelling works in), random blob fields with holes, islands inside holes, diagonally
touching squares, components on the image border, thin lines and speckle, random area thresholds.  Compared: the three
stage masks (combined, after morphology, filled contours) and the centre list, bit for bit; every fifth case also as a
stack of two to five images through the batch entry points.  One handle serves every case, so labels left over from a
larger image would show in the next.
usage: python scripts/gpu_stress_simple.py [cases] [seed]"""
import os
import sys
import time

import numpy as np
from scipy import ndimage as ndi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from laser_3d_reconstruction_amd import SimpleLaserExtractor  # noqa: E402
from oracle import oracle as O  # noqa: E402

LO, HI, THR = (50, 100, 180), (70, 255, 255), 200


def image(rng, W, H):
    kind = int(rng.integers(0, 5))
    if kind == 0:      # smooth random field: blobs with holes
        z = ndi.gaussian_filter(rng.normal(size=(H, W)), rng.uniform(0.7, 3.0))
        m = z > rng.uniform(-0.5, 0.8) * z.std()
    elif kind == 1:    # mostly foreground: background islands, border contact everywhere
        z = ndi.gaussian_filter(rng.normal(size=(H, W)), rng.uniform(0.7, 2.0))
        m = z > -rng.uniform(0.3, 1.2) * z.std()
    elif kind == 2:    # salt and pepper (the morphology stage decides)
        m = rng.random((H, W)) < rng.uniform(0.2, 0.8)
    elif kind == 3:    # a laser-like line, with gaps
        yy, xx = np.mgrid[0:H, 0:W]
        m = np.abs(xx - (rng.uniform(0, W) + rng.uniform(-0.8, 0.8) * yy)) < rng.uniform(1.5, 5.0)
        m &= rng.random((H, 1)) < 0.9
    else:              # nested rings and diagonal squares
        m = np.zeros((H, W), bool)
        for _ in range(int(rng.integers(1, 6))):
            s = int(rng.integers(5, max(6, min(H, W) // 2 + 1)))
            y, x = int(rng.integers(-2, max(1, H - s + 3))), int(rng.integers(-2, max(1, W - s + 3)))
            for k, v in ((0, True), (2, False), (4, True), (6, False)):
                if s - 2 * k > 0:
                    m[max(y + k, 0):max(y + s - k, 0), max(x + k, 0):max(x + s - k, 0)] = v
        for _ in range(int(rng.integers(0, 5))):
            y, x = int(rng.integers(0, max(1, H - 6))), int(rng.integers(0, max(1, W - 6)))
            m[y:y + 3, x:x + 3] = True
            m[y + 3:y + 6, x + 3:x + 6] = True
    img = np.zeros((H, W, 3), np.uint8)
    img[...] = (30, 40, 30)
    img[m] = (140, 255, 140)
    img[..., 1] = np.where(m, rng.integers(245, 256, (H, W)), img[..., 1])
    return img


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
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 300
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    rng = np.random.default_rng(seed)
    t0, bad, rows = time.time(), 0, 0
    ex = SimpleLaserExtractor(LO, HI, THR, 50)
    for case in range(n):
        W, H = int(rng.integers(3, 420)), int(rng.integers(3, 260))
        if case % 10 == 0:
            W = int(rng.choice([63, 64, 65, 127, 128, 129, 192, 320, 640]))
        img = image(rng, W, H)
        ex.min_area = float(rng.choice([0, 1, 3, 10, 50, 200]))
        got, gm = ex.extract_points(img, want_masks=True)
        want, wm = O.simple_extract(img, LO, HI, THR, ex.min_area, want_masks=True)
        ok = np.array_equal(gm, wm) and np.array_equal(got, want)
        if ok and case % 7 == 0:        # the device entry point on the same image
            import torch
            dev = ex.extract_device(torch.from_numpy(img).cuda())
            ok = np.array_equal(np.asarray(dev, np.float64).reshape(-1, 2), want)
        if ok and case % 5 == 0:        # a stack of images of this size through the batch entry points, the handle reused
            stack = [img] + [image(rng, W, H) for _ in range(int(rng.integers(1, 5)))]
            wants = [O.simple_extract(f, LO, HI, THR, ex.min_area) for f in stack]
            gots = ex.extract_batch(stack) if case % 10 else ex.extract_batch_device(__import__("torch").from_numpy(np.stack(stack)).cuda())
            ok = all(np.array_equal(np.asarray(g, np.float64).reshape(-1, 2), w) for g, w in zip(gots, wants))
        rows += len(want)
        if not ok:
            bad += 1
            print(f"MISMATCH case {case}: {W}x{H} min_area {ex.min_area}: masks equal {[bool(np.array_equal(gm[k], wm[k])) for k in range(3)]}, "
                  f"{len(got)} / {len(want)} centres", flush=True)
        if (case + 1) % 100 == 0:
            print(f"{case + 1} cases, {rows} centres, {bad} mismatches, {time.time() - t0:.0f} s", flush=True)
    print(f"done: {n} cases, {rows} centres, {bad} mismatches")
    write_summary("simple_extractor", {"script": "scripts/gpu_stress_simple.py", "seed": seed, "cases": n, "centres": rows, "mismatches": bad,
                                       "compared": "three stage masks and the centre list against the CPU restatement, bit for bit",
                                       "seconds": round(time.time() - t0)})
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
