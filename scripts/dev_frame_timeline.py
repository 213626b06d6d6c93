"""Kernel timeline of one turn of the main frame loop.  Two uses:
  python3 scripts/dev_frame_timeline.py run [simple|steger] [eye width] [height]   (under rocprofv3 --kernel-trace)
  python3 scripts/dev_frame_timeline.py show <kernel_trace.csv>                    (start offset, duration, queue per kernel)"""
import csv
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def run(kind, w, h):
    import numpy as np
    from laser_3d_reconstruction_amd import Config, LaserReconstructionSystem
    from laser_3d_reconstruction_amd.utils import synthetic as S

    class Src:
        def __init__(self):
            lb, rb, _ = S.stereo_pair(w, h, 64, 100)
            self.f = np.ascontiguousarray(np.concatenate([lb, rb], axis=1))
        isOpened = lambda self: True
        read = lambda self: (True, self.f)
        release = lambda self: None

    cfg = Config()
    cfg.LASER_EXTRACTOR_TYPE, cfg.STEREO_CALIBRATION_FILE = kind, "/nonexistent.json"
    s = LaserReconstructionSystem(config=cfg, frame_source=Src())
    assert s.initialize(width=2 * w, height=h)
    for _ in range(12):
        s.process_frame()


def show(path):
    rows = list(csv.DictReader(open(path)))
    for r in rows:
        r["s"], r["e"] = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        r["n"] = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("l3d::", "")[:40]
    rows.sort(key=lambda r: r["s"])
    # a frame begins with the gray conversion of the left eye; take the one before the last
    marks = [i for i, r in enumerate(rows) if r["n"].startswith("k_bgr2gray")]
    frames = marks[::2]
    a, b = frames[-2], frames[-1]
    t0 = rows[a]["s"]
    for r in rows[a:b]:
        print(f"{(r['s'] - t0) / 1e3:8.1f} us  +{(r['e'] - r['s']) / 1e3:7.1f} us  q{r['Queue_Id']:>3}  {r['n']}")
    print(f"kernels of the frame: span {(max(r['e'] for r in rows[a:b]) - t0) / 1e3:.1f} us, busy {sum(r['e'] - r['s'] for r in rows[a:b]) / 1e3:.1f} us; "
          f"next frame starts at {(rows[b]['s'] - t0) / 1e3:.1f} us")


if __name__ == "__main__":
    if sys.argv[1] == "run":
        run(sys.argv[2] if len(sys.argv) > 2 else "simple", int(sys.argv[3]) if len(sys.argv) > 3 else 320, int(sys.argv[4]) if len(sys.argv) > 4 else 360)
    else:
        show(sys.argv[2])
