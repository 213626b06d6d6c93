"""SingleUSBStereoCameraManager with the reference's constructor, attributes and
methods (camera/single_usb_stereo_camera.py:13-389), computing depth on the GPU.

Differences a caller can see:
  * ``compute_depth(left_img, right_img)`` is the frame-pair entry point the
    reference's README sketches (readme.md:367); ``get_frames`` delegates to it.
  * frames come from ``frame_source`` (any object with ``read() -> (ok, frame)``,
    ``isOpened()`` and ``release()``) instead of a cv2.VideoCapture, because
    camera capture is outside the accelerated path.
  * when a ``Config`` is given, its STEREO_* knobs select the matcher parameters
    (the reference prints them but always uses its width rule); without one the
    reference's width rule applies (:239-248).
  * the rectification maps live on the GPU (``rectifier_left`` / ``rectifier_right``);
    ``map_left_x`` ... are host copies fetched on first use.  R1, R2, P1, P2, Q come
    from ``rectify.stereo_rectify`` as in the reference (:176-187); a calibration
    file that carries its own "Q" but no distortion model is still honoured when
    ``use_file_rectification=True``.
"""
from __future__ import annotations

import json
import os
from typing import Dict, Optional, Tuple

import numpy as np

from .. import ops, stereo
from . import rectify


class SingleUSBStereoCameraManager:
    def __init__(self, camera_id: int = 0, width: int = 640, height: int = 240, fps: int = 30,
                 split_mode: str = 'horizontal', calibration_file: str = 'stereo_calibration.json',
                 config=None, device: int = 0, frame_source=None, verbose: bool = False, max_batch: int = 1,
                 use_file_rectification: bool = False):
        self.camera_id = camera_id
        self.width = width
        self.height = height
        self.fps = fps
        self.split_mode = split_mode
        self.calibration_file = calibration_file
        self.config = config
        self.device = device
        self.verbose = verbose
        self.max_batch = max_batch
        self.use_file_rectification = use_file_rectification

        self.cap = frame_source
        self.camera_matrix_left = None
        self.dist_coeffs_left = None
        self.camera_matrix_right = None
        self.dist_coeffs_right = None
        self.R = None
        self.T = None
        self.R1 = self.R2 = self.P1 = self.P2 = self.Q = None
        # as in the reference, where stereoRectify's valid rectangles go to local names (:176) and these two stay None
        # (callers test them: test_improved_laser.py:137); the rectangles themselves are kept in valid_roi_left / _right
        self.roi_left = self.roi_right = None
        self.valid_roi_left = self.valid_roi_right = None
        self.rectifier_left = self.rectifier_right = None
        self.stereo_matcher = None
        self.right_matcher = None
        self.wls_filter = None
        self.use_wls = True
        self._side_stream = None       # the right matcher's stream (see _depth_on_device)

        if split_mode == 'horizontal':
            self.single_width = width // 2
            self.single_height = height
        else:
            self.single_width = width
            self.single_height = height // 2
        self._log(f"stereo camera manager: total {width}x{height}, per eye {self.single_width}x{self.single_height}, {split_mode}")

    def _log(self, msg):
        if self.verbose:
            print(msg)

    # ------------------------------------------------------------------ init
    def initialize(self) -> bool:
        if self.cap is None or not self.cap.isOpened():
            self._log(f"cannot open camera {self.camera_id}: no frame source")
            return False
        ret, frame = self.cap.read()
        if not ret or frame is None:
            self._log("cannot read a frame")
            return False
        if not self._load_calibration():
            self._use_default_calibration()
        self._initialize_stereo_matcher_optimized()
        for _ in range(10):
            self.cap.read()
        return True

    def initialize_offline(self) -> bool:
        """Calibration + matcher set-up without a frame source (for compute_depth on caller-supplied pairs)."""
        if not self._load_calibration():
            self._use_default_calibration()
        self._initialize_stereo_matcher_optimized()
        return True

    def _split_frame(self, frame: np.ndarray) -> Tuple[np.ndarray, np.ndarray]:
        if self.split_mode == 'horizontal':
            mid = frame.shape[1] // 2
            return frame[:, :mid], frame[:, mid:]
        mid = frame.shape[0] // 2
        return frame[:mid, :], frame[mid:, :]

    def _load_calibration(self) -> bool:
        if not self.calibration_file or not os.path.exists(self.calibration_file):
            self._log(f"calibration file not found: {self.calibration_file}")
            return False
        try:
            with open(self.calibration_file, 'r', encoding='utf-8') as f:
                calib = json.load(f)
            self.camera_matrix_left = np.array(calib['camera_matrix_left'])
            self.dist_coeffs_left = np.array(calib['dist_coeffs_left'])
            self.camera_matrix_right = np.array(calib['camera_matrix_right'])
            self.dist_coeffs_right = np.array(calib['dist_coeffs_right'])
            self.R = np.array(calib['R'])
            self.T = np.array(calib['T'])
            if self.use_file_rectification and all(k in calib for k in ('R1', 'R2', 'P1', 'P2', 'Q')):
                for key in ('R1', 'R2', 'P1', 'P2', 'Q'):
                    setattr(self, key, np.array(calib[key], dtype=np.float64))
            else:
                image_size = (self.single_width, self.single_height)
                self.R1, self.R2, self.P1, self.P2, self.Q, self.valid_roi_left, self.valid_roi_right = rectify.stereo_rectify(
                    self.camera_matrix_left, self.dist_coeffs_left, self.camera_matrix_right, self.dist_coeffs_right,
                    image_size, self.R, self.T, flags=rectify.CALIB_ZERO_DISPARITY, alpha=0)
            self.rectifier_left = self.rectifier_right = None  # built on the device at first use
            return True
        except Exception as e:  # same convention as the reference: report and fall back
            self._log(f"calibration load failed: {e}")
            return False

    def _ensure_rectifiers(self) -> bool:
        """cv2.initUndistortRectifyMap for both eyes (reference :190-206), on the device."""
        if self.R1 is None or self.P1 is None:
            return False
        if self.rectifier_left is None:
            size = (self.single_width, self.single_height)
            self.rectifier_left = rectify.Rectifier.from_calibration(self.camera_matrix_left, self.dist_coeffs_left, self.R1, self.P1,
                                                                    size, device=self.device)
            self.rectifier_right = rectify.Rectifier.from_calibration(self.camera_matrix_right, self.dist_coeffs_right, self.R2, self.P2,
                                                                     size, device=self.device)
        return True

    @property
    def map_left_x(self):
        return self.rectifier_left.maps[0] if self._ensure_rectifiers() else None

    @property
    def map_left_y(self):
        return self.rectifier_left.maps[1] if self._ensure_rectifiers() else None

    @property
    def map_right_x(self):
        return self.rectifier_right.maps[0] if self._ensure_rectifiers() else None

    @property
    def map_right_y(self):
        return self.rectifier_right.maps[1] if self._ensure_rectifiers() else None

    def rectify_pair(self, left_img: np.ndarray, right_img: np.ndarray) -> Tuple[np.ndarray, np.ndarray]:
        """cv2.remap of both eyes (reference :312-317); the images pass through unchanged without a calibration."""
        if not self._ensure_rectifiers():
            return np.ascontiguousarray(left_img), np.ascontiguousarray(right_img)
        return self.rectifier_left.remap(left_img), self.rectifier_right.remap(right_img)

    def _use_default_calibration(self):
        fx = fy = 350.0
        cx = self.single_width / 2.0
        cy = self.single_height / 2.0
        self.camera_matrix_left = np.array([[fx, 0, cx], [0, fy, cy], [0, 0, 1]])
        self.camera_matrix_right = self.camera_matrix_left.copy()
        self.dist_coeffs_left = np.zeros((1, 5))
        self.dist_coeffs_right = np.zeros((1, 5))

    def _matcher_knobs(self):
        cfg = self.config
        if cfg is not None:
            return (int(cfg.STEREO_NUM_DISPARITIES), int(cfg.STEREO_BLOCK_SIZE), bool(cfg.STEREO_USE_WLS_FILTER),
                    float(cfg.STEREO_LAMBDA), float(cfg.STEREO_SIGMA))
        if self.single_width < 400:
            return 64, 5, True, 8000.0, 1.5
        return 96, 7, True, 8000.0, 1.5

    def _initialize_stereo_matcher_optimized(self):
        num_disparities, block_size, use_wls, lam, sigma = self._matcher_knobs()
        self.use_wls = use_wls
        self.stereo_matcher = stereo.StereoSGBM_create(
            minDisparity=0, numDisparities=num_disparities, blockSize=block_size,
            P1=8 * 3 * block_size ** 2, P2=32 * 3 * block_size ** 2,
            disp12MaxDiff=1, uniquenessRatio=10, speckleWindowSize=100, speckleRange=32,
            preFilterCap=63, mode=stereo.STEREO_SGBM_MODE_SGBM_3WAY,
            device=self.device, max_batch=self.max_batch)
        if use_wls:
            self.right_matcher = stereo.createRightMatcher(self.stereo_matcher)
            self.wls_filter = stereo.createDisparityWLSFilter(self.stereo_matcher)
            self.wls_filter.setLambda(lam)
            self.wls_filter.setSigmaColor(sigma)
        self._log(f"SGBM_3WAY D={num_disparities} block={block_size} wls={use_wls}")

    # ------------------------------------------------------------- per frame
    def _to_gray(self, img: np.ndarray) -> np.ndarray:
        if img.ndim == 3 and img.shape[-1] == 3:
            return ops.bgr2gray(img, device=self.device)
        return np.ascontiguousarray(img)

    def compute_disparity(self, left_img: np.ndarray, right_img: np.ndarray) -> np.ndarray:
        """int16 disparity*16 of a rectified pair: SGBM (+ right SGBM + WLS when enabled)."""
        if self.stereo_matcher is None:
            self._initialize_stereo_matcher_optimized()
        left_gray = self._to_gray(left_img)
        right_gray = self._to_gray(right_img)
        disparity_left = self.stereo_matcher.compute(left_gray, right_gray)
        if self.use_wls and self.wls_filter is not None:
            disparity_right = self.right_matcher.compute(right_gray, left_gray)
            return self.wls_filter.filter(disparity_left, left_gray, disparity_map_right=disparity_right)
        return disparity_left

    def _depth_on_device(self, gl, gr):
        """(B,H,W) uint8 gray tensors on the GPU -> float32 depth tensor; nothing leaves the device in between."""
        if self.stereo_matcher is None:
            self._initialize_stereo_matcher_optimized()
        if self.use_wls and self.wls_filter is not None:
            # The two matchers are independent, and one frame leaves most of the chip idle: the right matcher runs on
            # a stream of its own beside the left one, and the filter waits for both.
            import torch

            cur = torch.cuda.current_stream(gl.device)
            if self._side_stream is None:
                self._side_stream = torch.cuda.Stream(device=gl.device)
            self._side_stream.wait_stream(cur)
            with torch.cuda.stream(self._side_stream):
                dr = self.right_matcher.compute_device(gr, gl)
            d = self.stereo_matcher.compute_device(gl, gr)
            cur.wait_stream(self._side_stream)
            dr.record_stream(cur)
            d = self.wls_filter.filter_device(d, gl, dr)
        else:
            d = self.stereo_matcher.compute_device(gl, gr)
        return ops.depth_from_disparity16_device(d, self.Q)

    def _upload(self, img: np.ndarray):
        import torch

        return torch.from_numpy(np.ascontiguousarray(img)).to(torch.device("cuda", self.device))

    def compute_depth(self, left_img: np.ndarray, right_img: np.ndarray) -> np.ndarray:
        """Depth map (float32, Q units, 0 = invalid) of a rectified BGR or gray pair; the body of
        the reference's get_frames from the gray conversion on (:319-357).  One upload of the pair, one download
        of the depth map: the stages in between hand device buffers to each other."""
        if left_img.shape != right_img.shape or left_img.dtype != np.uint8 or right_img.dtype != np.uint8:
            raise ValueError("compute_depth expects two uint8 images of one shape")
        L, R = self._upload(left_img)[None], self._upload(right_img)[None]
        gl = ops.bgr2gray_device(L) if L.dim() == 4 else L
        gr = ops.bgr2gray_device(R) if R.dim() == 4 else R
        return self._depth_on_device(gl, gr)[0].cpu().numpy()

    def get_frames_device(self, rectified_event=None):
        """The next frame as CUDA tensors: (rectified left image (1,H,W,3) or (1,H,W) uint8, depth map (1,H,W) float32),
        or (None, None) when no frame can be read.  What get_frames downloads; a caller that goes on working on the
        device (laser extractor, back-projection) starts from here.  `rectified_event` (a torch.cuda.Event) is recorded
        on the current stream once the rectified left image is complete, ahead of the matchers: work that needs only
        that image can wait for it on another stream and run beside them."""
        if self.cap is None or not self.cap.isOpened():
            return None, None
        ret, frame = self.cap.read()
        if not ret:
            return None, None
        if (not self._ensure_rectifiers() or frame.ndim != 3 or frame.shape[2] != 3 or frame.dtype != np.uint8
                or (frame.shape[1] // 2 if self.split_mode == 'horizontal' else frame.shape[1]) != self.single_width
                or (frame.shape[0] if self.split_mode == 'horizontal' else frame.shape[0] // 2) != self.single_height):
            left_img, right_img = self._split_frame(frame)
            left_rectified, right_rectified = self.rectify_pair(left_img, right_img)
            if left_rectified.shape != right_rectified.shape or left_rectified.dtype != np.uint8 or right_rectified.dtype != np.uint8:
                raise ValueError("get_frames expects the two eyes as uint8 images of one shape")
            L, R = self._upload(left_rectified)[None], self._upload(right_rectified)[None]
            if rectified_event is not None:
                rectified_event.record()
            gl = ops.bgr2gray_device(L) if L.dim() == 4 else L
            gr = ops.bgr2gray_device(R) if R.dim() == 4 else R
            return L, self._depth_on_device(gl, gr)
        # The combined frame is uploaded once; each eye is remapped straight out of it (the right eye starts half a
        # row, or half a frame, further on), the gray planes come out of the remap, and only the rectified left
        # image and the depth map are kept (reference :312-357).
        F = self._upload(frame)[None]
        fh, fw, ch = frame.shape
        w, h = self.single_width, self.single_height
        right_off = w * ch if self.split_mode == 'horizontal' else h * fw * ch
        view = dict(row_stride=fw * ch, img_stride=fh * fw * ch, src_size=(w, h))
        left_rect, gl = self.rectifier_left.remap_device(F, want_bgr=True, want_gray=True, **view)
        if rectified_event is not None:
            rectified_event.record()
        _, gr = self.rectifier_right.remap_device(F, want_bgr=False, want_gray=True, offset_bytes=right_off, **view)
        return left_rect, self._depth_on_device(gl, gr)

    def get_frames(self) -> Tuple[Optional[np.ndarray], Optional[np.ndarray]]:
        left, depth = self.get_frames_device()
        if left is None:
            return None, None
        return left[0].cpu().numpy(), depth[0].cpu().numpy()

    def get_camera_intrinsics(self) -> Optional[Dict]:
        if self.camera_matrix_left is None:
            return None
        K = self.P1[:3, :3] if self.P1 is not None else self.camera_matrix_left
        baseline = np.linalg.norm(self.T) if self.T is not None else 0.06
        return {
            'width': self.single_width, 'height': self.single_height,
            'fx': float(K[0, 0]), 'fy': float(K[1, 1]), 'cx': float(K[0, 2]), 'cy': float(K[1, 2]),
            'baseline': float(baseline),
        }

    def stop(self):
        if self.cap is not None:
            self.cap.release()
