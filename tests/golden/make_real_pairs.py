#!/usr/bin/env python3
"""Packs four of the reference's own stereo pairs (calibration_images/{left,right}/*.png, 320x240 camera frames of a
checkerboard: sensor noise, saturated highlights, low-texture walls) and the K / distortion / R / T of its
stereo_calibration.json into tests/golden/real_pairs.npz.  Data only: pixels as BGR uint8 arrays (the channel order
cv2.VideoCapture delivers and the reference's code works on) and the calibration numbers.  No expected outputs ship
with the reference for these images; the tests run the GPU path against the CPU oracle on them.

    python tests/golden/make_real_pairs.py        # needs /root/reference and Pillow
"""
import json
import os

import numpy as np
from PIL import Image

REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))


def main():
    left_dir = os.path.join(REF, "calibration_images", "left")
    names = sorted(os.listdir(left_dir))
    picks = [names[i] for i in (0, 9, 18, 27)]
    left, right = [], []
    for n in picks:
        rn = n.replace("left_", "right_", 1)
        l = np.asarray(Image.open(os.path.join(left_dir, n)).convert("RGB"))
        r = np.asarray(Image.open(os.path.join(REF, "calibration_images", "right", rn)).convert("RGB"))
        left.append(l[:, :, ::-1])      # RGB file order -> BGR
        right.append(r[:, :, ::-1])
    c = json.load(open(os.path.join(REF, "stereo_calibration.json")))
    out = dict(left_bgr=np.ascontiguousarray(np.stack(left)), right_bgr=np.ascontiguousarray(np.stack(right)),
               names=np.array(picks),
               K1=np.array(c["camera_matrix_left"], np.float64), D1=np.array(c["dist_coeffs_left"], np.float64),
               K2=np.array(c["camera_matrix_right"], np.float64), D2=np.array(c["dist_coeffs_right"], np.float64),
               R=np.array(c["R"], np.float64), T=np.array(c["T"], np.float64))
    path = os.path.join(HERE, "real_pairs.npz")
    np.savez_compressed(path, **out)
    print(path, os.path.getsize(path), out["left_bgr"].shape)


if __name__ == "__main__":
    main()
