"""Deterministic synthetic rectified stereo pairs with a green laser ridge.

Used by the benchmark and the parity tests in place of camera frames (the
reference only ever runs on a live UVC device,
camera/single_usb_stereo_camera.py:294-309).  Frame ``k`` is a pure function
of ``(width, height, num_disparities, k)``.
"""
from __future__ import annotations

import numpy as np


def _box5(a: np.ndarray) -> np.ndarray:
    p = np.pad(a, 2, mode="reflect")
    c = np.cumsum(np.cumsum(p, axis=0), axis=1)
    c = np.pad(c, ((1, 0), (1, 0)))
    h, w = a.shape
    return (c[5:5 + h, 5:5 + w] - c[0:h, 5:5 + w] - c[5:5 + h, 0:w] + c[0:h, 0:w]) / 25.0


def disparity_scene(width: int, height: int, num_disparities: int, rng) -> np.ndarray:
    """Right-view disparity: slanted background plus 3-5 fronto-parallel rectangles,
    values in [4, num_disparities - 8]."""
    lo, hi = 4.0, float(num_disparities - 8)
    yy, xx = np.mgrid[0:height, 0:width].astype(np.float64)
    d = lo + (hi - lo) * 0.25 * (0.3 + 0.5 * xx / width + 0.2 * yy / height)
    for _ in range(int(rng.integers(3, 6))):
        rw = int(rng.integers(width // 8, width // 3))
        rh = int(rng.integers(height // 8, height // 3))
        x0 = int(rng.integers(0, width - rw))
        y0 = int(rng.integers(0, height - rh))
        d[y0:y0 + rh, x0:x0 + rw] = float(rng.uniform(lo + 0.3 * (hi - lo), hi))
    return d


def stereo_pair(width: int = 640, height: int = 360, num_disparities: int = 64, k: int = 0,
                laser: bool = True, noise: float = 1.5):
    """Returns (left_bgr, right_bgr, disparity_right_view) as (H,W,3) u8, (H,W,3) u8, (H,W) f64."""
    rng = np.random.default_rng(1234 + k)
    tex = _box5(rng.uniform(0.0, 1.0, (height, width)))
    tex = (tex - tex.min()) / max(tex.max() - tex.min(), 1e-9)
    left = 20.0 + 160.0 * tex
    d = disparity_scene(width, height, num_disparities, rng)
    # right(x, y) = left(x + d(x, y), y), linear interpolation, clamped at the border
    xs = np.arange(width)[None, :] + d
    x0 = np.clip(np.floor(xs).astype(np.int64), 0, width - 1)
    x1 = np.clip(x0 + 1, 0, width - 1)
    fr = np.clip(xs - np.floor(xs), 0.0, 1.0)
    rows = np.arange(height)[:, None]
    right = left[rows, x0] * (1.0 - fr) + left[rows, x1] * fr
    right = right + rng.normal(0.0, noise, right.shape)
    left_u8 = np.clip(np.rint(left), 0, 255).astype(np.uint8)
    right_u8 = np.clip(np.rint(right), 0, 255).astype(np.uint8)
    left_bgr = np.repeat(left_u8[:, :, None], 3, axis=2)
    right_bgr = np.repeat(right_u8[:, :, None], 3, axis=2)
    if laser:
        left_bgr = add_laser(left_bgr)
    return left_bgr, right_bgr, d


def laser_centre(width: int, height: int) -> np.ndarray:
    y = np.arange(height, dtype=np.float64)
    return 0.5 * width + 0.15 * width * np.sin(2.0 * np.pi * y / height) + 0.1 * y * (width / 640.0) * (360.0 / height)


def add_laser(bgr: np.ndarray, s: float = 2.0) -> np.ndarray:
    """Blends a green ridge into the frame: flat-topped Gaussian profile (about 4 px at full
    strength) towards B,G,R = (140, 255, 140).  That colour has gray 208 (> the extractors'
    brightness threshold 200, which the texture, capped at 180, never reaches) and HSV
    (60, 115, 255), inside Config's SIMPLE_LASER_HSV range."""
    h, w, _ = bgr.shape
    xc = laser_centre(w, h)[:, None]
    prof = np.minimum(1.0, 1.6 * np.exp(-0.5 * ((np.arange(w)[None, :] - xc) / s) ** 2))
    out = bgr.astype(np.float64)
    for c, target in enumerate((140.0, 255.0, 140.0)):
        out[:, :, c] += (target - out[:, :, c]) * prof
    return np.clip(np.rint(out), 0, 255).astype(np.uint8)


def shifted_pair(width: int, height: int, shift: int, seed: int = 7):
    """Known-answer pair: right(x) = left(x + shift) exactly (gray, no noise)."""
    rng = np.random.default_rng(seed)
    tex = _box5(rng.uniform(0.0, 1.0, (height, width + shift)))
    tex = (tex - tex.min()) / max(tex.max() - tex.min(), 1e-9)
    big = np.clip(np.rint(20.0 + 200.0 * tex), 0, 255).astype(np.uint8)
    left = big[:, :width].copy()
    right = big[:, shift:shift + width].copy()
    return left, right
