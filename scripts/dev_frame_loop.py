"""Single-stream rate of the two frame loops (one camera frame per call, host frame in, host results out): what a user of
the reference's main.py / test_improved_laser.py sees after switching.  The reference's README quotes 18-22 frames/s for
its main loop at 320x360 per eye, 64 disparities, on a 4-core laptop CPU (readme.md:238).

usage: python scripts/dev_frame_loop.py [frames]      -> one JSON line
"""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from laser_3d_reconstruction_amd import Config, LaserReconstructionSystem  # noqa: E402
from laser_3d_reconstruction_amd.system import ImprovedLaserSystem  # noqa: E402
from laser_3d_reconstruction_amd.utils import synthetic as S  # noqa: E402


class Stream:
    def __init__(self, frames):
        self.frames, self.reads = frames, 0

    def isOpened(self):
        return True

    def read(self):
        f = self.frames[self.reads % len(self.frames)]
        self.reads += 1
        return True, f

    def release(self):
        pass


def frames_of(w, h, n=8):
    out = []
    for k in range(n):
        lb, rb, _ = S.stereo_pair(w, h, 64, 100 + k)
        out.append(np.ascontiguousarray(np.concatenate([lb, rb], axis=1)))
    return out


def rate(step, n):
    for _ in range(5):
        step()
    t0 = time.perf_counter()
    for _ in range(n):
        step()
    return n / (time.perf_counter() - t0)


def main():
    import torch

    n = int(sys.argv[1]) if len(sys.argv) > 1 else 200
    out = {}
    for w, h in ((320, 360), (640, 360)):
        frames = frames_of(w, h)
        for kind in ("simple", "steger"):
            cfg = Config()
            cfg.LASER_EXTRACTOR_TYPE = kind
            cfg.STEREO_CALIBRATION_FILE = "/nonexistent.json"      # default intrinsics, no remap: the README's quoted case
            s = LaserReconstructionSystem(config=cfg, frame_source=Stream(frames))
            assert s.initialize(width=2 * w, height=h)
            out[f"main_loop_{kind}_{w}x{h}_per_eye_fps"] = round(rate(s.process_frame, n), 1)
            out[f"main_loop_{kind}_{w}x{h}_points_per_frame"] = len(s.point_cloud) // s.frame_count

        class Cfg(Config):
            CAMERA_WIDTH, CAMERA_HEIGHT, STEREO_CALIBRATION_FILE = 2 * w, h, "/nonexistent.json"

        for kind in ("hybrid", "steger"):
            s = ImprovedLaserSystem(frame_source=Stream(frames), config=Cfg, output_dir="/tmp/l3d_frame_loop_out")
            assert s.initialize()
            s.current_extractor = kind
            out[f"second_loop_{kind}_{w}x{h}_per_eye_fps"] = round(rate(s.process_frame, n), 1)
    out["device"] = torch.cuda.get_device_name(0)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
