"""Generates tests/golden/depth_golden.npz by running the reference's own per-frame depth code
(camera/single_usb_stereo_camera.py: _initialize_stereo_matcher_optimized :233-292 and get_frames :294-359) on
stored frames.  Run once in the build container, where /root/reference exists:

    python tests/golden/make_depth_golden.py

The reference's method bodies run unmodified: the width rule that picks numDisparities / blockSize (:239-248), the
matcher keyword block (:252-274), the right matcher and the disparity filter with lambda 8000 / sigma 1.5 (:277-282),
the frame split (:143-150), remap when maps are loaded (:312-317), gray conversion, the three matcher / filter calls,
`astype(float32) / 16` (:335), `reprojectImageTo3D(...)[:, :, 2]` and the three validity masks in the reference's order
(:340-346), or the uncalibrated fallback (:349-357).  The `cv2` operators behind those lines come from the CPU oracle
(tests/golden/_cv2_standin.py: OpenCV is not installed), so the vectors pin the reference's own glue -- parameters, call
order, scaling, masks, return values -- conditional on operators whose parity with OpenCV is unpinned.  The camera
is an object with isOpened() / read() that hands out the stored frame; capture itself is not part of the path.
Only numeric inputs and outputs are written.
"""
import contextlib
import importlib.util
import io
import os
import sys

sys.dont_write_bytecode = True   # importing the reference must not leave a bytecode cache beside its sources

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)
REF = "/root/reference"
OUT = os.path.join(HERE, "depth_golden.npz")

from _cv2_standin import O, add_stereo_names, make_cv2_standin  # noqa: E402


class StoredFrame:
    def __init__(self, frame):
        self.frame = frame

    def isOpened(self):
        return True

    def read(self):
        return True, self.frame.copy()


def main():
    sys.modules["cv2"] = add_stereo_names(make_cv2_standin())
    spec = importlib.util.spec_from_file_location("ref_camera", os.path.join(REF, "camera", "single_usb_stereo_camera.py"))
    ref = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(ref)
    from laser_3d_reconstruction_amd.camera import rectify       # host-side numpy: supplies R1 / P1 / Q as INPUT data
    from laser_3d_reconstruction_amd.utils import synthetic as S

    store, names = {}, []

    def run(name, frame, calib=None, with_maps=False):
        h, w2 = frame.shape[:2]
        with contextlib.redirect_stdout(io.StringIO()):
            cam = ref.SingleUSBStereoCameraManager(width=w2, height=h)
            cam.cap = StoredFrame(frame)
            if calib is not None:
                cam.Q = calib["Q"]
                if with_maps:
                    size = (w2 // 2, h)
                    cam.map_left_x, cam.map_left_y = O.init_undistort_rectify_map(calib["K1"], calib["D1"], calib["R1"], calib["P1"], size)
                    cam.map_right_x, cam.map_right_y = O.init_undistort_rectify_map(calib["K2"], calib["D2"], calib["R2"], calib["P2"], size)
            cam._initialize_stereo_matcher_optimized()
            left, depth = cam.get_frames()
        store[f"{name}/frame"] = frame
        store[f"{name}/left"] = np.ascontiguousarray(left)
        store[f"{name}/depth"] = np.ascontiguousarray(depth, np.float32)
        store[f"{name}/with_maps"] = np.array(int(with_maps))
        p = cam.stereo_matcher.params
        store[f"{name}/matcher"] = np.array([cam.wls_filter.wls_base.numDisparities, cam.wls_filter.wls_base.blockSize, p.P1, p.P2,
                                             cam.wls_filter.wls_base.disp12MaxDiff, cam.wls_filter.wls_base.uniquenessRatio,
                                             cam.wls_filter.wls_base.speckleWindowSize, cam.wls_filter.wls_base.speckleRange,
                                             p.preFilterCap, p.mode], np.int64)
        store[f"{name}/wls"] = np.array([cam.wls_filter.lam, cam.wls_filter.sigma])
        if calib is not None:
            for k, v in calib.items():
                store[f"{name}/calib/{k}"] = np.asarray(v, np.float64)
        names.append(name)
        print(f"{name}: eye {w2 // 2}x{h}, D={cam.wls_filter.wls_base.numDisparities} b={cam.wls_filter.wls_base.blockSize}, "
              f"{int((depth > 0).sum())} valid depths, max {float(depth.max()):.3f}", file=sys.stderr)

    # 1. narrow eyes (width rule: 64 / 5), no calibration: the fallback depth 0.06 * 350 / d
    lb, rb, _ = S.stereo_pair(320, 120, 64, 11)
    run("narrow_uncalibrated", np.concatenate([lb, rb], axis=1))
    # 2. the reference's own camera frame and calibration: rectification maps, Q, masks
    real = np.load(os.path.join(HERE, "real_pairs.npz"))
    K1, D1, K2, D2, R, T = (real[k] for k in ("K1", "D1", "K2", "D2", "R", "T"))
    R1, R2, P1, P2, Q, _, _ = rectify.stereo_rectify(K1, D1, K2, D2, (320, 240), R, T, alpha=0)
    calib = dict(K1=K1, D1=D1, K2=K2, D2=D2, R1=R1, R2=R2, P1=P1, P2=P2, Q=Q)
    run("real_pair_calibrated", np.concatenate([real["left_bgr"][1], real["right_bgr"][1]], axis=1), calib, with_maps=True)
    # 3. wide eyes (width rule: 96 / 7) with a Q and no maps: frames that are rectified already
    lb, rb, _ = S.stereo_pair(448, 96, 96, 12)
    f = 209.02533991591997 * (448 / 320.0)
    Qw = np.array([[1, 0, 0, -223.5], [0, 1, 0, -47.5], [0, 0, 0, f], [0, 0, 15.941451590053976, 0]], np.float64)
    run("wide_q_only", np.concatenate([lb, rb], axis=1), dict(Q=Qw))
    store["names"] = np.array(names)
    np.savez_compressed(OUT, **store)
    print("wrote", OUT, os.path.getsize(OUT), "bytes", file=sys.stderr)


if __name__ == "__main__":
    main()
