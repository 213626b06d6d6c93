"""A stand-in for the `cv2` module, for the golden-vector generators only (the make_*_golden.py scripts beside it).

OpenCV is not installed in the build container.  The reference's modules call it for their image operators; to run
the reference's OWN Python around those calls, this module object answers the same names with the CPU oracle's
restatements of the operators (oracle/l3d_oracle.c; the trivial ones in numpy).  Vectors made through it pin the
reference's own code paths, conditional on these operators, whose parity with OpenCV is unpinned.  Never imported by
the product package or by a test.
"""
import ctypes as C
import os
import sys
import types

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from oracle import oracle as O  # noqa: E402


def _reflect101(i, n):
    if n == 1:
        return np.zeros_like(i)
    p = 2 * (n - 1)
    i = np.mod(i, p)
    return np.where(i >= n, p - i, i)


def make_cv2_standin():
    """A module object with the cv2 names core/laser_extractor.py uses, backed by the oracle's restatements."""
    cv2 = types.ModuleType("cv2")
    cv2.__doc__ = "stand-in for OpenCV: operators restated by oracle/l3d_oracle.c (parity with OpenCV unpinned)"
    cv2.COLOR_BGR2GRAY, cv2.COLOR_BGR2HSV = 6, 40
    cv2.THRESH_BINARY = 0
    cv2.MORPH_OPEN, cv2.MORPH_CLOSE = 2, 3
    cv2.RETR_EXTERNAL, cv2.CHAIN_APPROX_SIMPLE = 0, 2
    cv2.CV_32F = 5
    lib = O.lib()

    def cvtColor(img, code):
        if code == cv2.COLOR_BGR2GRAY:
            return O.bgr2gray(img)
        if code == cv2.COLOR_BGR2HSV:
            return O.bgr2hsv(img)
        raise NotImplementedError(code)

    def inRange(src, lo, hi):
        lo, hi = np.asarray(lo).reshape(1, 1, -1), np.asarray(hi).reshape(1, 1, -1)
        return (np.all((src >= lo) & (src <= hi), axis=2) * 255).astype(np.uint8)

    def threshold(src, thresh, maxval, kind):
        assert kind == cv2.THRESH_BINARY
        return thresh, np.where(src > thresh, maxval, 0).astype(np.uint8)

    def bitwise_and(a, b):
        return a & b

    def _morph(src, dilate):
        src = np.ascontiguousarray(src, np.uint8)
        dst = np.empty_like(src)
        h, w = src.shape
        lib.orc_morph3x3(O._p(src, C.c_uint8), w, h, int(dilate), O._p(dst, C.c_uint8))
        return dst

    def morphologyEx(src, op, kernel):
        assert kernel.shape == (3, 3) and np.all(kernel == 1)
        if op == cv2.MORPH_CLOSE:
            return _morph(_morph(src, 1), 0)
        if op == cv2.MORPH_OPEN:
            return _morph(_morph(src, 0), 1)
        raise NotImplementedError(op)

    def findContours(mask, mode, method):
        assert mode == cv2.RETR_EXTERNAL
        mask = np.ascontiguousarray(mask, np.uint8)
        h, w = mask.shape
        cap = 4 * (w + 2) * (h + 2)
        xy = np.empty((cap, 2), np.int32)
        lens = np.empty(w * h + 1, np.int32)
        n = lib.orc_find_external_contours(O._p(mask, C.c_uint8), w, h, O._p(xy, C.c_int), cap, O._p(lens, C.c_int), len(lens))
        out, o = [], 0
        for k in range(n):
            out.append(xy[o:o + lens[k]].reshape(-1, 1, 2).copy())   # cv2's (N, 1, 2) int32 layout, every traced point kept
            o += lens[k]
        return tuple(out), None

    def contourArea(cnt):
        p = np.asarray(cnt, np.float64).reshape(-1, 2)
        q = np.roll(p, 1, axis=0)
        a = 0.0
        for (px, py), (qx, qy) in zip(q, p):      # same summation order as the restatement
            a += px * qy - py * qx
        return abs(a * 0.5)

    def drawContours(img, contours, idx, color, thickness):
        assert idx == -1 and thickness == -1 and img.dtype == np.uint8 and img.flags.c_contiguous
        h, w = img.shape
        for c in contours:
            pts = np.ascontiguousarray(np.asarray(c, np.int32).reshape(-1, 2))
            lib.orc_fill_contour(O._p(pts, C.c_int), len(pts), w, h, O._p(img, C.c_uint8), int(color))
        return img

    def GaussianBlur(src, ksize, sx, sy=None):
        sy = sx if sy is None else sy
        assert tuple(ksize) == (0, 0) and sx == sy and src.dtype == np.float32
        src = np.ascontiguousarray(src)
        h, w = src.shape
        dst = np.empty_like(src)
        assert lib.orc_gauss_blur_f32(O._p(src, C.c_float), w, h, C.c_double(sx), O._p(dst, C.c_float)) == 0
        return dst

    def filter2D(src, ddepth, kernel):
        """Correlation with the anchor at (kw // 2, kh // 2), BORDER_REFLECT_101, float32 accumulation over the taps in
        raster order starting from zero."""
        assert ddepth == cv2.CV_32F and src.dtype == np.float32
        k = np.asarray(kernel, np.float32)
        kh, kw = k.shape
        ay, ax = kh // 2, kw // 2
        h, w = src.shape
        ys, xs = np.arange(h), np.arange(w)
        acc = np.zeros((h, w), np.float32)
        for r in range(kh):
            yy = _reflect101(ys + r - ay, h)
            for c in range(kw):
                xx = _reflect101(xs + c - ax, w)
                acc = (acc + k[r, c] * src[np.ix_(yy, xx)]).astype(np.float32)
        return acc

    def Sobel(src, ddepth, dx, dy, ksize=3):
        assert ddepth == cv2.CV_32F and ksize == 3 and (dx, dy) in ((1, 0), (0, 1)) and src.dtype == np.float32
        src = np.ascontiguousarray(src)
        h, w = src.shape
        dst = np.empty_like(src)
        assert lib.orc_sobel3_f32(O._p(src, C.c_float), w, h, int(dx == 1), O._p(dst, C.c_float)) == 0
        return dst

    for f in (cvtColor, inRange, threshold, bitwise_and, morphologyEx, findContours, contourArea, drawContours, GaussianBlur, filter2D, Sobel):
        setattr(cv2, f.__name__, f)
    return cv2


def add_stereo_names(cv2):
    """The names camera/single_usb_stereo_camera.py uses: StereoSGBM_create, ximgproc.createRightMatcher,
    ximgproc.createDisparityWLSFilter, reprojectImageTo3D, remap (reference :252-282, 313-340)."""
    cv2.STEREO_SGBM_MODE_SGBM_3WAY = 2
    cv2.INTER_LINEAR = 1
    cv2.COLORMAP_JET = 2          # a default argument of the reference's visualize_depth (:394), never called here

    class StereoSGBM:
        def __init__(self, params):
            self.params = params

        def compute(self, left, right):
            return O.sgbm_compute(self.params, left, right)

    def StereoSGBM_create(minDisparity=0, numDisparities=16, blockSize=3, P1=0, P2=0, disp12MaxDiff=0, preFilterCap=0,
                          uniquenessRatio=0, speckleWindowSize=0, speckleRange=0, mode=0):
        return StereoSGBM(O.sgbm_params(minDisparity, numDisparities, blockSize, P1, P2, disp12MaxDiff, preFilterCap,
                                        uniquenessRatio, speckleWindowSize, speckleRange, mode))

    class DisparityWLSFilter:
        def __init__(self, left):
            self.left, self.lam, self.sigma = left, 8000.0, 1.0
            # the factory reconfigures the left matcher: no left-right check, no speckle filter, no uniqueness test
            self.wls_base = O.sgbm_params(**{f: getattr(left.params, f) for f, _ in left.params._fields_})
            left.params = O.wls_left_matcher_params(left.params)

        def setLambda(self, v):
            self.lam = float(v)

        def setSigmaColor(self, v):
            self.sigma = float(v)

        def filter(self, disparity_map_left, left_view, disparity_map_right=None):
            return O.wls_filter(O.wls_params(self.wls_base, self.lam, self.sigma), disparity_map_left, disparity_map_right, left_view)

    xi = types.ModuleType("cv2.ximgproc")
    xi.createRightMatcher = lambda left: StereoSGBM(O.right_matcher_params(left.params))
    xi.createDisparityWLSFilter = lambda left: DisparityWLSFilter(left)
    cv2.ximgproc = xi
    cv2.StereoSGBM_create = StereoSGBM_create

    def reprojectImageTo3D(disparity, Q):
        d16 = np.asarray(disparity, np.float32) * np.float32(16.0)
        assert np.array_equal(d16, np.round(d16)) and np.abs(d16).max() < 32768      # the reference passes int16 / 16
        _, xyz = O.depth_from_disp16(d16.astype(np.int16), np.asarray(Q, np.float64), want_xyz=True)
        return xyz

    def remap(src, map_x, map_y, interpolation):
        assert interpolation == cv2.INTER_LINEAR
        return O.remap_bilinear(src, map_x, map_y)

    cv2.reprojectImageTo3D = reprojectImageTo3D
    cv2.remap = remap
    return cv2


def add_capture_names(cv2, frames, stereo_rectify):
    """The names SingleUSBStereoCameraManager.initialize / _load_calibration use (reference :93-113, 176-206):
    VideoCapture (hands out `frames` in turn, round and round, and counts the reads in VideoCapture.reads),
    stereoRectify (`stereo_rectify`: a host-side function supplied by the caller -- its results are INPUT data of the
    vectors, stored with them), initUndistortRectifyMap (the oracle's restatement)."""
    cv2.CAP_PROP_FRAME_WIDTH, cv2.CAP_PROP_FRAME_HEIGHT, cv2.CAP_PROP_FPS = 3, 4, 5
    cv2.CALIB_ZERO_DISPARITY = 1024
    cv2.CV_32FC1 = 5

    class VideoCapture:
        reads = 0
        rectified = None      # what stereoRectify returned last, for the generator to store

        def __init__(self, camera_id):
            self.camera_id, self.props = camera_id, {}

        def isOpened(self):
            return True

        def set(self, prop, value):
            self.props[prop] = value
            return True

        def get(self, prop):
            return self.props.get(prop, 0)

        def read(self):
            frame = frames[VideoCapture.reads % len(frames)]
            VideoCapture.reads += 1
            return True, frame.copy()

        def release(self):
            pass

    def stereoRectify(K1, D1, K2, D2, size, R, T, flags=0, alpha=-1):
        assert flags == cv2.CALIB_ZERO_DISPARITY
        out = stereo_rectify(K1, D1, K2, D2, size, R, T, alpha=alpha)
        VideoCapture.rectified = out
        return out

    def initUndistortRectifyMap(K, D, R, P, size, m1type):
        assert m1type == cv2.CV_32FC1
        return O.init_undistort_rectify_map(K, D, R, P, size)

    cv2.VideoCapture, cv2.stereoRectify, cv2.initUndistortRectifyMap = VideoCapture, stereoRectify, initUndistortRectifyMap
    return cv2
