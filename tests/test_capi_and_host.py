"""CPU tests: the C ABI library loads and exports every symbol include/l3d.h declares, the
host-side mirror of the reference classes behaves like the reference, and the N>1 plumbing
works over gloo."""
import ctypes as C
import os
import re
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    text = open(os.path.join(ROOT, "include", "l3d.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(l3d_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    from laser_3d_reconstruction_amd import _native as N

    lib = C.CDLL(N.LIB_PATH)
    names = _declared_symbols()
    assert len(names) >= 25
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/l3d.h but not exported"
    assert set(N.EXPORTED) == set(names), set(N.EXPORTED) ^ set(names)
    assert b"gfx950" in N.lib.l3d_version()


def test_entry_points_fail_loudly_without_a_gpu():
    from laser_3d_reconstruction_amd import _native as N

    if N.device_count() > 0:
        pytest.skip("a GPU is present")
    p = N.SgbmParams(0, 64, 5, 600, 2400, 1, 63, 10, 100, 32, 2)
    h = C.c_void_p()
    assert N.lib.l3d_sgbm_create(C.byref(p), 640, 360, 1, 0, C.byref(h)) == -4
    assert b"no HIP device" in N.lib.l3d_last_error()
    from laser_3d_reconstruction_amd import stereo

    with pytest.raises(N.L3DError):
        stereo.StereoSGBM_create(0, 64, 5, mode=2).compute(np.zeros((64, 128), np.uint8), np.zeros((64, 128), np.uint8))


def test_argument_validation_needs_no_gpu():
    from laser_3d_reconstruction_amd import _native as N

    h = C.c_void_p()
    bad_mode = N.SgbmParams(0, 64, 5, 600, 2400, 1, 63, 10, 100, 32, 3)   # MODE_HH4 is not built
    assert N.lib.l3d_sgbm_create(C.byref(bad_mode), 640, 360, 1, 0, C.byref(h)) == -2
    assert b"3WAY" in N.lib.l3d_last_error()
    bad_d = N.SgbmParams(0, 24, 5, 600, 2400, 1, 63, 10, 100, 32, 2)
    assert N.lib.l3d_sgbm_create(C.byref(bad_d), 640, 360, 1, 0, C.byref(h)) == -2
    out = (C.c_int * 8)()
    ok = N.SgbmParams(0, 64, 5, 600, 2400, 1, 63, 10, 100, 32, 2)
    assert N.lib.l3d_sgbm_geometry(C.byref(ok), 640, 360, out) == 0
    assert list(out) == [64, 576, 90, 12, -16, 63, 600, 2400]


def test_geometry_agrees_with_oracle(O):
    from laser_3d_reconstruction_amd import stereo

    for (w, h, d, b, mind) in [(640, 360, 64, 5, 0), (1280, 720, 96, 9, 0), (1920, 1080, 128, 11, 0), (333, 131, 64, 3, -63)]:
        m = stereo.StereoSGBM_create(mind, d, b, 8 * 3 * b * b, 32 * 3 * b * b, 1, 63, 10, 100, 32, 2)
        g = m.geometry(w, h)
        og = O.sgbm_geometry(O.sgbm_params(minDisparity=mind, numDisparities=d, blockSize=b), w, h)
        og.pop("rc")
        assert g == og


def test_matcher_factories_follow_opencv():
    from laser_3d_reconstruction_amd import stereo

    left = stereo.StereoSGBM_create(minDisparity=0, numDisparities=64, blockSize=5, P1=600, P2=2400, disp12MaxDiff=1,
                                    uniquenessRatio=10, speckleWindowSize=100, speckleRange=32, preFilterCap=63,
                                    mode=stereo.STEREO_SGBM_MODE_SGBM_3WAY)
    right = stereo.createRightMatcher(left)
    assert right.params.as_dict() == dict(minDisparity=-63, numDisparities=64, blockSize=5, P1=600, P2=2400,
                                          disp12MaxDiff=1000000, preFilterCap=63, uniquenessRatio=0,
                                          speckleWindowSize=0, speckleRange=0, mode=2)
    assert left.getUniquenessRatio() == 10 and left.getSpeckleWindowSize() == 100
    wls = stereo.createDisparityWLSFilter(left)
    # the factory reconfigures the left matcher exactly as OpenCV's does
    assert (left.getDisp12MaxDiff(), left.getSpeckleWindowSize(), left.getUniquenessRatio()) == (1000000, 0, 0)
    assert wls.getDepthDiscontinuityRadius() == 3 and wls.getLRCthresh() == 24 and wls.getLambda() == 8000.0
    wls.setLambda(10000.0)
    wls.setSigmaColor(1.5)
    p = wls.params()
    assert (p.lambda_, p.sigma_color, p.min_disparity, p.num_disparities, p.fgs_iters) == (10000.0, 1.5, 0, 64, 3)


def test_config_keeps_reference_knobs():
    from laser_3d_reconstruction_amd import Config

    d = Config.to_dict()
    expect = {
        "SINGLE_USB_CAMERA_ID": 1, "CAMERA_WIDTH": 640, "CAMERA_HEIGHT": 240, "CAMERA_FPS": 30, "SPLIT_MODE": "horizontal",
        "STEREO_CALIBRATION_FILE": "stereo_calibration.json", "STEREO_NUM_DISPARITIES": 64, "STEREO_BLOCK_SIZE": 5,
        "STEREO_USE_WLS_FILTER": True, "STEREO_LAMBDA": 8000.0, "STEREO_SIGMA": 1.5, "LASER_EXTRACTOR_TYPE": "simple",
        "SIMPLE_LASER_HSV_LOWER": [50, 100, 180], "SIMPLE_LASER_HSV_UPPER": [70, 255, 255],
        "SIMPLE_LASER_BRIGHTNESS_THRESHOLD": 200, "SIMPLE_LASER_MIN_AREA": 50, "STEGER_SIGMA": 3.0,
        "STEGER_BRIGHTNESS_THRESHOLD": 200, "STEGER_USE_LUT": True, "LASER_PLANE_COEFFICIENTS": [0, 0, 1, 0],
        "USE_REFRACTION_CORRECTION": False, "WATER_REFRACTION_INDEX": 1.33, "VOXEL_SIZE": 0.002,
        "OUTLIER_REMOVAL_NEIGHBORS": 20, "OUTLIER_REMOVAL_STD_RATIO": 2.0, "SAVE_FORMAT": "ply", "DEBUG_MODE": False,
        "OUTPUT_DIR": "output", "AUTO_SAVE_INTERVAL": 60, "MIN_POINT_CLOUD_SIZE": 100, "JETSON_OPTIMIZED": False,
        "USE_CUDA": False, "NUM_THREADS": 4,
    }
    assert d == expect


def test_camera_manager_surface_and_knobs():
    from laser_3d_reconstruction_amd import Config, SingleUSBStereoCameraManager

    cam = SingleUSBStereoCameraManager()
    assert (cam.width, cam.height, cam.fps, cam.split_mode, cam.calibration_file) == (640, 240, 30, "horizontal", "stereo_calibration.json")
    assert (cam.single_width, cam.single_height) == (320, 240)
    for attr in ("cap", "Q", "P1", "T", "map_left_x", "map_left_y", "map_right_x", "map_right_y", "roi_left", "roi_right",
                 "stereo_matcher", "right_matcher", "wls_filter"):
        assert getattr(cam, attr) is None
    assert cam.get_frames() == (None, None) and cam.initialize() is False and cam.get_camera_intrinsics() is None
    frame = np.arange(240 * 640 * 3, dtype=np.uint8).reshape(240, 640, 3)
    l, r = cam._split_frame(frame)
    assert l.shape == (240, 320, 3) and np.array_equal(r, frame[:, 320:])
    # width rule of the reference when no Config is given
    assert cam._matcher_knobs()[:2] == (64, 5)
    assert SingleUSBStereoCameraManager(width=1280, height=360)._matcher_knobs()[:2] == (96, 7)

    class Knobs(Config):
        STEREO_NUM_DISPARITIES = 128
        STEREO_BLOCK_SIZE = 11
        STEREO_LAMBDA = 10000.0

    cam = SingleUSBStereoCameraManager(width=3840, height=1080, config=Knobs, calibration_file="/nonexistent.json")
    cam.initialize_offline()
    assert cam.stereo_matcher.getNumDisparities() == 128 and cam.stereo_matcher.getBlockSize() == 11
    assert cam.stereo_matcher.getP1() == 8 * 3 * 121 and cam.stereo_matcher.getP2() == 32 * 3 * 121
    assert cam.wls_filter.getLambda() == 10000.0 and cam.wls_filter.getSigmaColor() == 1.5
    assert cam.right_matcher.getMinDisparity() == -127
    assert cam.get_camera_intrinsics()["fx"] == 350.0 and cam.get_camera_intrinsics()["baseline"] == 0.06


def test_reconstruction_system_surface():
    """The attributes and return conventions of the reference's LaserReconstructionSystem (main.py:40-49, 164-169, 193-195)."""
    from laser_3d_reconstruction_amd import Config, LaserReconstructionSystem

    class Source:
        def __init__(self, ok):
            self.ok, self.reads = ok, 0

        def isOpened(self):
            return True

        def read(self):
            self.reads += 1
            return (True, np.zeros((240, 640, 3), np.uint8)) if self.ok else (False, None)

        def release(self):
            pass

    s = LaserReconstructionSystem()
    assert isinstance(s.config, Config) and s.point_cloud == [] and s.frame_count == 0 and s.show_depth is True
    for attr in ("camera", "laser_extractor", "reconstructor", "point_cloud_processor", "start_time"):
        assert getattr(s, attr) is None
    assert s.initialize() is False                       # no frame source: the camera cannot be opened
    assert s.camera is not None and (s.camera.camera_id, s.camera.width, s.camera.height) == (1, 640, 240)
    src = Source(ok=False)
    s = LaserReconstructionSystem(frame_source=src)
    assert s.initialize(camera_id=0, width=1280, height=360) is False and src.reads == 1   # the test read failed
    assert (s.camera.camera_id, s.camera.single_width) == (0, 640) and s.laser_extractor is None
    assert s.save_point_cloud() is False                 # fewer points than MIN_POINT_CLOUD_SIZE


def test_point_cloud_collector_bookkeeping(tmp_path):
    """Counters, default colours, statistics keys, automatic saves and the file pair of the reference's collector
    (point_cloud_collector.py:41-67, 185-212, 278-306); clouds this small skip the downsampling and outlier steps."""
    import json
    from laser_3d_reconstruction_amd.point_cloud_collector import PointCloudCollector, RealTimePointCloudCollector

    c = PointCloudCollector(output_dir=str(tmp_path / "a"))
    assert c.get_statistics() == {'frame_count': 0, 'total_points': 0, 'avg_points_per_frame': 0}
    assert c.save() == "" and c.export_numpy() == ""
    c.add_points(np.zeros((0, 3)))
    assert c.frame_count == 0
    p1 = np.arange(12, dtype=np.float64).reshape(4, 3)
    c.add_points(p1)
    c.add_points(p1[:2] + 100, np.full((2, 3), 0.5))
    assert (c.frame_count, c.total_points) == (2, 6)
    pts, cols = c.get_point_cloud()
    assert pts.shape == (6, 3) and cols[:4].tolist() == [[0, 1, 0]] * 4 and cols[4:].tolist() == [[0.5] * 3] * 2
    st = c.get_statistics()
    assert st['unique_points'] == 6 and st['avg_points_per_frame'] == 3 and st['bounds_min'] == [0, 1, 2] and st['bounds_max'] == [103, 104, 105]
    path = c.save(filename="cloud")
    header, body = open(path, "rb").read().split(b"end_header\n", 1)
    assert path.endswith("cloud.ply") and b"element vertex 6" in header and len(body) == 6 * 27
    rec = np.frombuffer(body, np.dtype([("xyz", "<f8", 3), ("rgb", "u1", 3)]))
    assert np.array_equal(rec["xyz"], pts) and rec["rgb"][0].tolist() == [0, 255, 0] and rec["rgb"][5].tolist() == [128, 128, 128]
    meta = json.load(open(tmp_path / "a" / "cloud_metadata.json"))
    assert meta["total_points"] == 6 and meta["frame_count"] == 2 and meta["final_points"] == 6 and meta["voxel_size"] == 0.002
    assert c.save(format="obj") == ""
    assert c.export_numpy("n").endswith("n_points.npy") and np.array_equal(np.load(tmp_path / "a" / "n_colors.npy"), cols)
    c.clear()
    assert (c.frame_count, c.total_points, c.points, c.colors) == (0, 0, [], [])

    r = RealTimePointCloudCollector(output_dir=str(tmp_path / "b"), auto_save_interval=2, max_points=10)
    r.add_points(p1)
    assert (r.frames_since_save, r.save_count) == (1, 0)
    r.add_points(np.zeros((0, 3)))                       # an empty frame still counts towards the interval
    assert (r.frames_since_save, r.save_count, r.frame_count) == (0, 1, 1)
    r.add_points(p1)
    r.add_points(p1)                                     # interval reached again, then 12 points >= max_points: saved and emptied
    assert r.save_count == 3 and r.total_points == 0 and r.points == []
    assert len([f for f in os.listdir(tmp_path / "b") if f.endswith(".ply")]) == 3


def test_camera_manager_loads_reference_style_calibration(tmp_path):
    import json
    from laser_3d_reconstruction_amd import SingleUSBStereoCameraManager

    calib = {"camera_matrix_left": [[208.9, 0, 166.3], [0, 209.5, 131.2], [0, 0, 1]], "dist_coeffs_left": [-0.06, 0.18, 0, 0, 0],
             "camera_matrix_right": [[209.2, 0, 174.0], [0, 209.9, 133.1], [0, 0, 1]], "dist_coeffs_right": [-0.04, 0.09, 0, 0, 0],
             "R": np.eye(3).tolist(), "T": [-0.0627, -0.0001, -0.0004],
             "P1": [[209.0, 0, 159.5, 0], [0, 209.7, 119.5, 0], [0, 0, 1, 0]],
             "Q": [[1, 0, 0, -159.5], [0, 1, 0, -119.5], [0, 0, 0, 209.0], [0, 0, -15.94, 0]]}
    f = tmp_path / "calib.json"
    f.write_text(json.dumps(calib))
    cam = SingleUSBStereoCameraManager(calibration_file=str(f))
    assert cam._load_calibration()
    # like the reference (camera/single_usb_stereo_camera.py:176-187) the rectification is recomputed from
    # K, dist, R, T; whatever R1/P1/Q the file carries is ignored
    k = cam.get_camera_intrinsics()
    assert k["fx"] == k["fy"] == cam.P1[0, 0] == cam.P2[0, 0] and k["fx"] != 209.0
    assert abs(k["baseline"] - np.linalg.norm(calib["T"])) < 1e-12
    assert cam.Q.shape == (4, 4) and cam.Q[2, 3] == cam.P1[0, 0] and cam.Q[0, 3] == -cam.P1[0, 2]
    assert abs(cam.P2[0, 3] / cam.P2[0, 0] + np.linalg.norm(calib["T"])) < 1e-6     # Tx * f, rotated baseline
    assert cam.valid_roi_left[2] > 300 and cam.valid_roi_left[3] > 220
    assert cam.roi_left is None and cam.roi_right is None      # the reference never assigns them


def _random_rig(rng):
    from laser_3d_reconstruction_amd.camera import rectify as RC

    K1 = np.array([[400 + rng.uniform(-20, 20), 0, 320 + rng.uniform(-15, 15)], [0, 405 + rng.uniform(-20, 20), 180 + rng.uniform(-10, 10)], [0, 0, 1]])
    K2 = np.array([[400 + rng.uniform(-20, 20), 0, 320 + rng.uniform(-15, 15)], [0, 405 + rng.uniform(-20, 20), 180 + rng.uniform(-10, 10)], [0, 0, 1]])
    R = RC.rodrigues_to_matrix(rng.uniform(-0.03, 0.03, 3))
    T = np.array([-0.06 + rng.uniform(-0.005, 0.005), rng.uniform(-0.002, 0.002), rng.uniform(-0.002, 0.002)])
    D1 = np.array([rng.uniform(-0.1, 0.05), rng.uniform(-0.05, 0.1), rng.uniform(-1e-3, 1e-3), rng.uniform(-1e-3, 1e-3), 0.0])
    D2 = np.array([rng.uniform(-0.1, 0.05), rng.uniform(-0.05, 0.1), rng.uniform(-1e-3, 1e-3), rng.uniform(-1e-3, 1e-3), 0.0])
    return K1, D1, K2, D2, R, T


def test_stereo_rectify_geometry():
    """Properties cv2.stereoRectify guarantees (no OpenCV here to compare numbers with): R1, R2 are rotations,
    both rectified cameras share one focal length and (ZERO_DISPARITY) one principal point, every scene point
    lands on the same row in both views, its column difference is -Tx*f/Z, Q inverts that, alpha=0 keeps only
    valid pixels."""
    from laser_3d_reconstruction_amd.camera import rectify as RC

    rng = np.random.default_rng(5)
    for trial in range(6):
        K1, D1, K2, D2, R, T = _random_rig(rng)
        if trial == 4:      # vertical rig
            T = np.array([0.001, -0.07, 0.002])
        R1, R2, P1, P2, Q, roi1, roi2 = RC.stereo_rectify(K1, D1, K2, D2, (640, 360), R, T, alpha=0)
        for Rk in (R1, R2):
            assert np.allclose(Rk @ Rk.T, np.eye(3), atol=1e-12) and abs(np.linalg.det(Rk) - 1) < 1e-12
        assert P1[0, 0] == P1[1, 1] == P2[0, 0] == P2[1, 1] > 0
        assert np.array_equal(P1[:, :3], P2[:, :3])
        idx = 1 if trial == 4 else 0
        assert P2[idx, 3] != 0 and P2[1 - idx, 3] == 0 and np.all(P1[:, 3] == 0)
        assert abs(abs(P2[idx, 3]) / P2[0, 0] - np.linalg.norm(T)) < 1e-9       # the whole baseline lies along the axis
        X = np.c_[rng.uniform(-0.4, 0.4, 50), rng.uniform(-0.3, 0.3, 50), rng.uniform(0.4, 3.0, 50)]   # left camera frame
        a = (R1 @ X.T).T                                  # rectified left frame
        b = (R2 @ ((R @ X.T).T + T).T).T                  # rectified right frame
        assert np.allclose(b - a, (R2 @ T)[None, :], atol=1e-12) and abs((R2 @ T)[1 - idx]) < 1e-12 and abs((R2 @ T)[2]) < 1e-12
        ua = (P1 @ np.c_[a, np.ones(50)].T).T
        ub = (P2 @ np.c_[a, np.ones(50)].T).T
        ua, ub = ua[:, :2] / ua[:, 2:], ub[:, :2] / ub[:, 2:]
        assert np.abs(ua[:, 1 - idx] - ub[:, 1 - idx]).max() < 1e-9
        d = ua[:, idx] - ub[:, idx]
        assert np.allclose(d, -P2[idx, 3] / a[:, 2], rtol=1e-12)
        if idx == 0:
            h = np.c_[ua, d, np.ones(50)] @ Q.T
            assert np.abs(h[:, :3] / h[:, 3:] - a).max() < 1e-9
        for roi in (roi1, roi2):
            assert roi[0] >= 0 and roi[1] >= 0 and roi[0] + roi[2] <= 640 and roi[1] + roi[3] <= 360 and roi[2] > 500 and roi[3] > 250
    # identical ideal cameras, pure x translation: nothing to rotate
    K = np.array([[350.0, 0, 160], [0, 350.0, 120], [0, 0, 1]])
    R1, R2, P1, P2, Q, roi1, roi2 = RC.stereo_rectify(K, np.zeros(5), K, np.zeros(5), (320, 240), np.eye(3), [-0.06, 0, 0], alpha=0)
    assert np.allclose(R1, np.eye(3)) and np.allclose(R2, np.eye(3))
    assert abs(P2[0, 3] + 0.06 * P1[0, 0]) < 1e-12 and abs(Q[3, 2] - 1 / 0.06) < 1e-9


def test_rodrigues_and_undistort_points_roundtrip():
    from laser_3d_reconstruction_amd.camera import rectify as RC

    rng = np.random.default_rng(2)
    for _ in range(20):
        om = rng.normal(size=3)
        om *= rng.uniform(0, 3.1) / np.linalg.norm(om)      # rotation angle below pi: the vector form is unique
        Rm = RC.rodrigues_to_matrix(om)
        assert np.allclose(Rm @ Rm.T, np.eye(3), atol=1e-13)
        assert np.allclose(RC.rodrigues_to_vector(Rm), om, atol=1e-9)
    assert np.array_equal(RC.rodrigues_to_matrix(np.zeros(3)), np.eye(3))
    # distort normalised points with the forward model, undistort_points must bring them back (5 iterations: ~1e-6)
    K = np.array([[400.0, 0, 320], [0, 410.0, 180], [0, 0, 1]])
    dist = np.array([-0.08, 0.05, 1e-3, -5e-4, 0.01])
    xy = rng.uniform(-0.5, 0.5, (30, 2))
    k1, k2, p1, p2, k3 = dist
    r2 = (xy ** 2).sum(1)
    kr = 1 + ((k3 * r2 + k2) * r2 + k1) * r2
    xd = xy[:, 0] * kr + 2 * p1 * xy[:, 0] * xy[:, 1] + p2 * (r2 + 2 * xy[:, 0] ** 2)
    yd = xy[:, 1] * kr + p1 * (r2 + 2 * xy[:, 1] ** 2) + 2 * p2 * xy[:, 0] * xy[:, 1]
    px = np.c_[xd * 400 + 320, yd * 410 + 180]
    assert np.abs(RC.undistort_points(px, K, dist) - xy).max() < 2e-5


def test_reconstructor_host_helpers():
    from laser_3d_reconstruction_amd import Reconstructor

    K = np.array([[209.025, 0, 159.5], [0, 209.72, 119.5], [0, 0, 1.0]])
    r = Reconstructor(K, np.array([0, 0, 1, -0.5]))
    assert r.water_refraction_index == 1.33 and r.use_refraction is True
    assert r.reconstruct_laser_line([]).shape == (0,) and r.reconstruct_from_depth([], np.zeros((4, 4), np.float32)).shape == (0,)
    pts = np.array([[0, 0, 1.0], [0.001, 0, 1.0], [0.5, 0, 1.0], [0.5005, 0, 1.0], [0.501, 0, 1.0]])
    out = r.filter_outliers(pts, threshold=0.01)
    np.testing.assert_array_equal(out, pts[[0, 3, 4]])     # same rule as core/reconstruction.py:201-217
    assert r.filter_outliers(pts[:2]) is not None and len(r.filter_outliers(pts[:2])) == 2
    R = np.array([[0, -1, 0], [1, 0, 0], [0, 0, 1.0]])
    np.testing.assert_allclose(r.transform_points(pts, R, np.array([1, 2, 3.0])), (R @ pts.T).T + [1, 2, 3])
    assert r.merge_point_clouds([]).shape == (0,) and r.merge_point_clouds([np.array([]), pts]).shape == (5, 3)


def test_synthetic_frames_are_deterministic_and_shaped_for_the_extractors(O):
    from laser_3d_reconstruction_amd.utils import synthetic as S

    a = S.stereo_pair(640, 360, 64, 5)
    b = S.stereo_pair(640, 360, 64, 5)
    assert all(np.array_equal(x, y) for x, y in zip(a, b))
    lb = a[0]
    g = O.bgr2gray(lb)
    hsv = O.bgr2hsv(lb)
    bright = g > 200
    assert 500 < bright.sum() < 5000                      # only the ridge is above the Steger threshold
    inr = (hsv[..., 0] >= 50) & (hsv[..., 0] <= 70) & (hsv[..., 1] >= 100) & (hsv[..., 2] >= 180)
    assert (inr & bright).sum() > 0.5 * bright.sum()      # and it is green enough for the HSV extractor


def test_shard_indices_partition_the_frames():
    from laser_3d_reconstruction_amd import parallel

    for world in (1, 2, 3, 8):
        got = sorted(i for r in range(world) for i in parallel.shard_indices(37, r, world))
        assert got == list(range(37))
    with pytest.raises(ValueError):
        parallel.shard_indices(4, 2, 2)


def _gloo_worker(rank, world, port, q):
    import torch.distributed as dist

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    sys.path.insert(0, ROOT)
    from laser_3d_reconstruction_amd import parallel

    calib = None
    if rank == 0:
        calib = {"Q": np.arange(16, dtype=np.float64).reshape(4, 4), "map_left_x": np.random.default_rng(1).random((7, 9), np.float32),
                 "num_disparities": np.int32(64)}
    got = parallel.broadcast_calibration(calib, src=0)
    mine = parallel.shard_indices(10, rank, world)
    counts = parallel.gather_counts(float(len(mine)))
    q.put((rank, got["Q"].tolist(), float(got["map_left_x"].sum()), int(got["num_disparities"]), mine, counts))
    dist.barrier()
    dist.destroy_process_group()


def test_broadcast_and_sharding_over_gloo_world_size_2():
    import torch.multiprocessing as mp

    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + os.getpid() % 2000
    procs = [ctx.Process(target=_gloo_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(2))
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    want_sum = float(np.random.default_rng(1).random((7, 9), np.float32).sum())
    for rank, Q, s, nd, mine, counts in res:
        assert Q == np.arange(16, dtype=np.float64).reshape(4, 4).tolist() and abs(s - want_sum) < 1e-6 and nd == 64
        assert mine == list(range(rank, 10, 2)) and counts == [5.0, 5.0]


def test_bench_launcher_plumbing(monkeypatch):
    """bench.py --gpus N starts N ranks itself (as a child process, before it touches the GPU) unless it already runs
    under torch.distributed.run; the child gets the same arguments."""
    import argparse
    import bench

    calls = []

    def fake_run(cmd, env=None):
        calls.append((cmd, env))
        return 7

    argv = ["--gpus", "4", "--steps", "3", "--warmup", "1", "--config", "C4"]
    ns = argparse.Namespace(gpus=4)
    assert bench.spawn_ranks_if_needed(ns, argv, environ={"PATH": "/usr/bin"}, run=fake_run) == 7
    cmd, env = calls[0]
    assert cmd[1:4] == ["-m", "torch.distributed.run", "--nnodes=1"] and "--nproc-per-node=4" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1"
    assert cmd[-len(argv):] == argv and os.path.basename(cmd[-len(argv) - 1]) == "bench.py"
    assert env["MASTER_ADDR"] == "127.0.0.1" and env["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"
    # already a rank, or a single GPU: no child
    assert bench.spawn_ranks_if_needed(ns, argv, environ={"WORLD_SIZE": "4"}, run=fake_run) is None
    assert bench.spawn_ranks_if_needed(argparse.Namespace(gpus=1), argv, environ={}, run=fake_run) is None
    assert len(calls) == 1
    # the eight streams of configs[3] land on ranks round-robin, every stream exactly once
    from laser_3d_reconstruction_amd import parallel
    for world in (1, 2, 4, 8):
        owned = [parallel.shard_indices(bench.C4_STREAMS, r, world) for r in range(world)]
        assert sorted(s for o in owned for s in o) == list(range(8)) and all(len(o) == 8 // world for o in owned)
        # and a rank's launch holds 256 frames whatever N is: 256 / (streams it owns) consecutive frames of each stream
        for r in range(world):
            B, ids, scaling, sharding = bench.plan_frames("C4", None, r, world)
            assert B == 256 and len(set(ids)) == 256 and scaling == "weak" and f"{256 * world // 8} frames per stream" in sharding
    assert bench.plan_frames("C4", 2, 1, 2)[0] == 8 and bench.plan_frames("C2", None, 3, 4)[1] == list(range(3, 1024, 4))
    with pytest.raises(SystemExit):
        bench.plan_frames("C4", None, 8, 9)
    # the synthetic frames of a shard come from worker processes and equal the generator's
    from laser_3d_reconstruction_amd.utils import synthetic as S
    l, r_ = bench.host_frames(96, 40, 16, [5, 2, 9])
    assert l.shape == (3, 40, 96, 3) and np.array_equal(l[1], S.stereo_pair(96, 40, 16, 2)[0]) and np.array_equal(r_[2], S.stereo_pair(96, 40, 16, 9)[1])
    assert bench.alg_bytes(640, 360, 64, False, 0)[1] == 4 * 640 * 360 * 65 + 9 * 640 * 360      # Z-only depth: 6 bytes a pixel
    # under a preloaded profiler the process neither starts other programs nor forks workers
    assert not bench.under_profiler({"LD_PRELOAD": "/some/other/lib.so"})
    assert bench.under_profiler({"LD_PRELOAD": "/opt/rocm/lib/librocprofiler-sdk-tool.so"}) and bench.under_profiler({"ROCPROF_OUTPUT_PATH": "x"})
    assert bench.alg_bytes(640, 360, 64, False, 0, xyz=True)[1] == 4 * 640 * 360 * 65 + 19 * 640 * 360


def test_bench_cpu_baseline_leg_runs_on_a_tiny_sample():
    """The CPU leg of bench.py (the C restatement in the reference's frame-loop order, one thread and a forked pool)."""
    import bench

    r = bench.cpu_baseline(160, 64, 32, 5, False, 8000.0, 1.5, True, seconds_budget=0.05)
    assert r["kind"] == "port" and r["value"] > 0 and r["one_thread"]["value"] > 0 and r["cores"] >= 1
    r = bench.cpu_baseline(160, 64, 32, 5, True, 8000.0, 1.5, False, seconds_budget=0.05)
    assert r["value"] > 0
