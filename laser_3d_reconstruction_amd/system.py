"""The reference's two frame loops on the GPU: LaserReconstructionSystem (main.py:37-233: depth map, then laser pixels
through it) and ImprovedLaserSystem (test_improved_laser.py:23-207: left matcher only, laser pixels triangulated from
the disparity map).

LaserReconstructionSystem: ``initialize`` builds the same four components from the same ``Config`` knobs
(main.py:51-162): the camera manager (without the Config, so the matcher follows the reference's width rule), the laser
extractor chosen by ``LASER_EXTRACTOR_TYPE``, a ``Reconstructor`` on the rectified intrinsics with
``LASER_PLANE_COEFFICIENTS`` / ``USE_REFRACTION_CORRECTION``, and the point-cloud processor.  ``process_frame`` is
main.py:164-189: frame -> rectified left image + depth map -> laser centre line -> 3-D points through the depth map ->
rows without NaN appended to ``point_cloud``.  Between those stages nothing leaves the device here: the combined camera
frame is uploaded once, the extractor reads the rectified left image where the remap kernel left it (the Simple
extractor on a stream of its own, beside the matchers), the back-projection reads the depth map where the depth kernel
left it, and the three return values plus the new 3-D points are downloaded at the end.

Differences a caller can see: frames come from ``frame_source`` (an object with ``isOpened() / read() / release()``;
camera capture is outside the accelerated path); the console window of ``run_realtime`` (cv2.imshow, key handling:
main.py:235-343) is not mirrored -- ``run`` below is its loop without the display.
"""
from __future__ import annotations

import os
import time
from typing import List, Optional

import numpy as np

from .camera.single_usb_stereo_camera import SingleUSBStereoCameraManager
from .config import Config
from .core.laser_extractor import FastStegerExtractor, SimpleLaserExtractor, _as_tuples
from .core.reconstruction import Reconstructor
from .improved_reconstruction import ImprovedLaserReconstructor
from .improved_steger import HybridLaserExtractor, ImprovedStegerExtractor
from .point_cloud_collector import RealTimePointCloudCollector
from .utils.point_cloud import PointCloudProcessor


class LaserReconstructionSystem:
    def __init__(self, config=None, frame_source=None, device: int = 0, verbose: bool = False):
        self.config = config or Config()
        self.frame_source = frame_source
        self.device = device
        self.verbose = verbose
        self.camera: Optional[SingleUSBStereoCameraManager] = None
        self.laser_extractor = None
        self.reconstructor: Optional[Reconstructor] = None
        self.point_cloud_processor: Optional[PointCloudProcessor] = None
        self.point_cloud: List[List[float]] = []
        self.frame_count = 0
        self.start_time = None
        self.show_depth = True
        self._side_stream = None       # the laser extractor's stream (see process_frame)

    def _log(self, msg):
        if self.verbose:
            print(msg)

    def initialize(self, camera_id=None, width=None, height=None) -> bool:
        cfg = self.config
        camera_id = camera_id if camera_id is not None else cfg.SINGLE_USB_CAMERA_ID
        width = width or cfg.CAMERA_WIDTH
        height = height or cfg.CAMERA_HEIGHT
        try:
            self.camera = SingleUSBStereoCameraManager(camera_id=camera_id, width=width, height=height, fps=cfg.CAMERA_FPS,
                                                       split_mode=cfg.SPLIT_MODE, calibration_file=cfg.STEREO_CALIBRATION_FILE,
                                                       device=self.device, frame_source=self.frame_source, verbose=self.verbose)
            if not self.camera.initialize():
                self._log("camera initialisation failed")
                return False
        except Exception as e:
            self._log(f"camera initialisation error: {e}")
            return False
        try:
            if cfg.LASER_EXTRACTOR_TYPE == 'simple':
                self.laser_extractor = SimpleLaserExtractor(hsv_lower=cfg.SIMPLE_LASER_HSV_LOWER, hsv_upper=cfg.SIMPLE_LASER_HSV_UPPER,
                                                            brightness_threshold=cfg.SIMPLE_LASER_BRIGHTNESS_THRESHOLD,
                                                            min_area=cfg.SIMPLE_LASER_MIN_AREA, device=self.device, verbose=self.verbose)
            else:
                self.laser_extractor = FastStegerExtractor(sigma=cfg.STEGER_SIGMA, brightness_threshold=cfg.STEGER_BRIGHTNESS_THRESHOLD,
                                                           use_lut=cfg.STEGER_USE_LUT, device=self.device, verbose=self.verbose)
        except Exception as e:
            self._log(f"laser extractor initialisation error: {e}")
            return False
        try:
            intr = self.camera.get_camera_intrinsics()
            if intr is None:
                self._log("no camera intrinsics")
                return False
            K = np.array([[intr['fx'], 0, intr['cx']], [0, intr['fy'], intr['cy']], [0, 0, 1]], dtype=np.float64)
            self.reconstructor = Reconstructor(camera_intrinsic=K, laser_plane=cfg.LASER_PLANE_COEFFICIENTS,
                                               use_refraction_correction=cfg.USE_REFRACTION_CORRECTION, device=self.device)
        except Exception as e:
            self._log(f"reconstructor initialisation error: {e}")
            return False
        try:
            self.point_cloud_processor = PointCloudProcessor(device=self.device)
        except Exception as e:
            self._log(f"point cloud processor initialisation error: {e}")
            return False
        return True

    def process_frame(self):
        """(color_image, depth_image, laser_points) of the next frame, or (None, None, None); the frame's valid 3-D
        points are appended to ``point_cloud`` (main.py:164-189)."""
        import torch

        rectified = torch.cuda.Event()
        left, depth = self.camera.get_frames_device(rectified_event=rectified)
        if left is None:
            return None, None, None
        steger_on_device = isinstance(self.laser_extractor, FastStegerExtractor) and left.dim() == 4
        simple_on_device = isinstance(self.laser_extractor, SimpleLaserExtractor) and left.dim() == 4
        if steger_on_device:
            # (three short kernels behind the depth map's: nothing to gain from a stream of their own)
            h, w = left.shape[1:3]
            xy, counts = self.laser_extractor.extract_device(left, w * h)
            n = int(counts[0].item())
            laser_points = _as_tuples(xy[0, :n].cpu().numpy())
            points_3d = np.empty((0, 3))
            if n > 0:
                xyz, keep = self.reconstructor.reconstruct_from_depth_device(xy, counts, depth)
                points_3d = xyz[0, :n][keep[0, :n] != 0].cpu().numpy()
        elif simple_on_device:
            # The Simple extractor needs only the rectified left image and ends with a host synchronisation: it waits for
            # that image on a stream of its own and runs beside the matchers and the disparity filter, which leave most
            # of the chip idle on a single frame.
            cur = torch.cuda.current_stream(left.device)
            if self._side_stream is None:
                self._side_stream = torch.cuda.Stream(device=left.device)
            self._side_stream.wait_event(rectified)
            with torch.cuda.stream(self._side_stream):
                laser_points = self.laser_extractor.extract_device(left[0])
            cur.wait_stream(self._side_stream)
            points_3d = []
            if len(laser_points) > 0:
                # the centroids are doubles and go back up as they are; the depth map is read where it lies
                uv = torch.tensor(np.asarray(laser_points, np.float64).reshape(-1, 2), dtype=torch.float64, device=left.device)
                xyz, keep = self.reconstructor.reconstruct_from_depth_uv64_device(uv, depth[0])
                points_3d = xyz[keep != 0].cpu().numpy()
        color_image = left[0].cpu().numpy()
        depth_image = depth[0].cpu().numpy()
        if not steger_on_device and not simple_on_device:
            laser_points = self.laser_extractor.extract_centerline(color_image)
            points_3d = self.reconstructor.reconstruct_from_depth(laser_points, depth_image) if len(laser_points) > 0 else []
        if len(points_3d) > 0:
            points_3d = points_3d[~np.isnan(points_3d).any(axis=1)]
            self.point_cloud.extend(points_3d.tolist())
        self.frame_count += 1
        return color_image, depth_image, laser_points

    def save_point_cloud(self, filename=None) -> bool:
        """Voxel downsampling, outlier removal and the PLY / PCD writer with the Config's knobs (main.py:191-233)."""
        cfg = self.config
        if len(self.point_cloud) < cfg.MIN_POINT_CLOUD_SIZE:
            self._log(f"point cloud too small ({len(self.point_cloud)} < {cfg.MIN_POINT_CLOUD_SIZE}): not saved")
            return False
        os.makedirs(cfg.OUTPUT_DIR, exist_ok=True)
        if filename is None:
            filename = f"point_cloud_{time.strftime('%Y%m%d_%H%M%S')}.{cfg.SAVE_FORMAT}"
        path = os.path.join(cfg.OUTPUT_DIR, filename)
        points = np.array(self.point_cloud, dtype=np.float32)
        points = self.point_cloud_processor.voxel_downsample(points, voxel_size=cfg.VOXEL_SIZE)
        points = self.point_cloud_processor.statistical_outlier_removal(points, nb_neighbors=cfg.OUTLIER_REMOVAL_NEIGHBORS,
                                                                        std_ratio=cfg.OUTLIER_REMOVAL_STD_RATIO)
        if cfg.SAVE_FORMAT == 'ply':
            self.point_cloud_processor.save_ply(points, path)
        else:
            self.point_cloud_processor.save_pcd(points, path)
        self._log(f"point cloud saved: {path} ({len(points)} points)")
        return True

    def run(self, max_frames=None, duration=None) -> int:
        """The frame loop of run_realtime (main.py:252-265) without its display: processes frames until the source
        runs dry, ``max_frames`` are done or ``duration`` seconds have passed; returns the number of frames."""
        self.start_time = time.time()
        done = 0
        while max_frames is None or done < max_frames:
            if duration and time.time() - self.start_time > duration:
                break
            color_image, _, _ = self.process_frame()
            if color_image is None:
                break
            done += 1
        return done


class ImprovedLaserSystem:
    """The loop body of the reference's second pipeline (test_improved_laser.py:23-207) without its windows and keys:
    raw frame -> split -> remap -> left matcher only (`camera.stereo_matcher.compute`, disparity / 16) -> laser centre
    line by the current extractor ('hybrid' by default, 'steger', 'simple') -> `reconstruct_with_interpolation`
    (3x3 window, disparities from 1.0) -> laser-only depth image; while `collecting` is on, the frame's points go to the
    collector with colours picked from the rectified left image (:184-200).

    ``process_frame`` keeps the frame on the device from the upload of the combined camera frame to the downloads of
    what it returns; the reference has the same steps inline in ``run`` (:112-207)."""

    def __init__(self, frame_source=None, device: int = 0, config=Config, output_dir: str = "output", verbose: bool = False):
        self.config = config
        self.frame_source = frame_source
        self.device = device
        self.verbose = verbose
        self.camera: Optional[SingleUSBStereoCameraManager] = None
        self.extractors = {
            'simple': None,
            'steger': ImprovedStegerExtractor(sigma=3.0, brightness_threshold=200, response_threshold=0.5, device=device),
            'hybrid': HybridLaserExtractor(hsv_lower=np.array([50, 100, 180]), hsv_upper=np.array([70, 255, 255]),
                                           brightness_threshold=200, sigma=2.0, device=device),
        }
        self.current_extractor = 'hybrid'
        self.reconstructor: Optional[ImprovedLaserReconstructor] = None
        self.point_cloud_collector = RealTimePointCloudCollector(output_dir=output_dir, auto_save_interval=0, max_points=100000,
                                                                 device=device)
        self.collecting = False
        self.frame_count = 0
        self.total_laser_points = 0
        self.total_3d_points = 0

    def initialize(self, camera_id=None) -> bool:
        cfg = self.config
        self.camera = SingleUSBStereoCameraManager(camera_id=camera_id if camera_id is not None else cfg.SINGLE_USB_CAMERA_ID,
                                                   width=cfg.CAMERA_WIDTH, height=cfg.CAMERA_HEIGHT, fps=cfg.CAMERA_FPS,
                                                   split_mode=cfg.SPLIT_MODE, calibration_file=cfg.STEREO_CALIBRATION_FILE,
                                                   device=self.device, frame_source=self.frame_source, verbose=self.verbose)
        if not self.camera.initialize():
            return False
        if self.camera.Q is not None:
            self.reconstructor = ImprovedLaserReconstructor(self.camera.Q, device=self.device)
        return True

    def _extractor(self):
        ex = self.extractors[self.current_extractor]
        if ex is None:      # the Simple extractor is built from the Config at first use (:157-166)
            cfg = self.config
            ex = SimpleLaserExtractor(hsv_lower=cfg.SIMPLE_LASER_HSV_LOWER, hsv_upper=cfg.SIMPLE_LASER_HSV_UPPER,
                                      brightness_threshold=cfg.SIMPLE_LASER_BRIGHTNESS_THRESHOLD, min_area=cfg.SIMPLE_LASER_MIN_AREA,
                                      device=self.device)
            self.extractors['simple'] = ex
        return ex

    def process_frame(self):
        """One turn of the reference's loop (:113-207).  Returns None when no frame can be read, else a dict with
        ``left_clean`` (H,W,3) uint8, ``disparity`` (H,W) float32, ``laser_points`` [(x, y), ...], ``points_3d``
        (N,3) float32 (np.array([]) when there are none) and ``laser_depth`` (H,W) float32."""
        import torch

        from . import ops

        cam = self.camera
        if cam.cap is None or not cam.cap.isOpened():
            return None
        ret, frame = cam.cap.read()
        if not ret or frame is None:
            return None
        if frame.ndim != 3 or frame.shape[2] != 3 or frame.dtype != np.uint8:
            raise ValueError("ImprovedLaserSystem expects BGR uint8 camera frames")
        if cam._ensure_rectifiers():
            F = cam._upload(frame)[None]
            fh, fw, ch = frame.shape
            w, h = cam.single_width, cam.single_height
            right_off = w * ch if cam.split_mode == 'horizontal' else h * fw * ch
            view = dict(row_stride=fw * ch, img_stride=fh * fw * ch, src_size=(w, h))
            left, gl = cam.rectifier_left.remap_device(F, want_bgr=True, want_gray=True, **view)
            _, gr = cam.rectifier_right.remap_device(F, want_bgr=False, want_gray=True, offset_bytes=right_off, **view)
        else:
            left_img, right_img = cam._split_frame(frame)
            left, R = cam._upload(left_img)[None], cam._upload(right_img)[None]
            gl, gr = ops.bgr2gray_device(left), ops.bgr2gray_device(R)
        # (camera.roi_left / roi_right stay None as in the reference, so its ROI crop at :137-142 never runs)
        disparity = cam.stereo_matcher.compute_device(gl, gr)[0].to(torch.float32) / 16.0
        ex = self._extractor()
        if isinstance(ex, SimpleLaserExtractor):
            laser_points = ex.extract_device(left[0])
            uv = torch.tensor(np.asarray(laser_points, np.float64).reshape(-1, 2), dtype=torch.float64, device=left.device)
        else:
            xy, count = ex.extract_device(left[0])
            n = int(count[0].item())
            laser_points = _as_tuples(xy[:n].cpu().numpy())
            uv = xy[:n].to(torch.float64).contiguous()
        self.total_laser_points += len(laser_points)
        points_3d = np.array([])
        if self.reconstructor is not None and len(laser_points) > 0:
            xyz, keep = self.reconstructor.triangulate_device(uv, disparity.contiguous(), window_size=3, min_disparity=1.0)
            kept = xyz[keep != 0].cpu().numpy()
            points_3d = kept if len(kept) else np.array([])
            self.total_3d_points += len(points_3d)
        left_clean = left[0].cpu().numpy()
        disparity_host = disparity.cpu().numpy()
        if self.reconstructor is not None and self.collecting and len(points_3d) > 0:
            # colours of the FIRST len(points_3d) laser pixels, as the reference pairs them (:188): BGR -> RGB in 0..1
            colors = []
            for x, y in laser_points[:len(points_3d)]:
                px, py = int(round(x)), int(round(y))
                if 0 <= px < left_clean.shape[1] and 0 <= py < left_clean.shape[0]:
                    bgr = left_clean[py, px]
                    colors.append(np.array([bgr[2], bgr[1], bgr[0]]) / 255.0)
                else:
                    colors.append([0, 1, 0])
            self.point_cloud_collector.add_points(points_3d, np.array(colors))
        laser_depth = np.zeros(left_clean.shape[:2], dtype=np.float32)
        if self.reconstructor is not None and len(laser_points) > 0:
            laser_depth = self.reconstructor.create_laser_depth_map(laser_points, disparity_host, left_clean.shape[:2])
        self.frame_count += 1
        return dict(left_clean=left_clean, disparity=disparity_host, laser_points=laser_points, points_3d=points_3d,
                    laser_depth=laser_depth)

    def run(self, max_frames=None) -> int:
        """Frames until the source runs dry or `max_frames` are done (the reference's loop ends on a key)."""
        if self.camera is None and not self.initialize():
            return 0
        done = 0
        while max_frames is None or done < max_frames:
            if self.process_frame() is None:
                break
            done += 1
        return done
