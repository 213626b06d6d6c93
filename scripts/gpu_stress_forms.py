"""Differential stress of the matcher's kernel variants on the GPU (development aid, not part of the test suite).

Random sizes, disparity ranges, block sizes, penalties and image statistics; for each case the disparity of
  * the unsigned form of the recurrence (default where it applies) and the signed form (L3D_SGBM_UFORM=0),
  * the path-only layout and the layout with both volumes (L3D_SGBM_TONLY=0),
  * a single-frame handle and a batch handle (other lane-group shapes),
  * the vertical kernel with its ring of box-sum rows in LDS instead of registers (L3D_SGBM_RING_REGS=0: the older row loop with two
    barriers and unshared box sums) and with 256-thread blocks (L3D_SGBM_VERTICAL_THREADS=256)
must be identical.  With a third argument N the first N cases (kept small) are also compared with the CPU restatement.
usage: python scripts/gpu_stress_forms.py [cases] [seed] [cases checked against the oracle]
"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from laser_3d_reconstruction_amd import stereo  # noqa: E402


def image_pair(rng, W, H, kind):
    if kind == 0:      # uniform noise
        a, b = rng.integers(0, 256, (H, W)), rng.integers(0, 256, (H, W))
    elif kind == 1:    # textured scene with a shift
        base = rng.integers(0, 256, (H, W + 64)).astype(np.float32)
        base = (base + np.roll(base, 1, 1) + np.roll(base, 1, 0)) / 3
        s = int(rng.integers(0, 40))
        a, b = base[:, 64:64 + W], base[:, 64 - s:64 - s + W]
    elif kind == 2:    # saturated stripes
        a = np.where((np.arange(W)[None, :] // int(rng.integers(1, 9))) % 2 == 0, 255, 0) * np.ones((H, 1))
        b = np.where((np.arange(W)[None, :] // int(rng.integers(1, 9))) % 2 == 0, 0, 255) * np.ones((H, 1))
    else:              # flat with sparse spikes
        a, b = np.full((H, W), 128.0), np.full((H, W), 120.0)
        a[rng.random((H, W)) < 0.02] = 255
        b[rng.random((H, W)) < 0.02] = 0
    return np.ascontiguousarray(a, dtype=np.uint8), np.ascontiguousarray(b, dtype=np.uint8)


ENV_KEYS = ("L3D_SGBM_UFORM", "L3D_SGBM_TONLY", "L3D_SGBM_RING_REGS", "L3D_SGBM_VERTICAL_THREADS")


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
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 200
    rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
    t0, bad, computed, errors = time.time(), 0, 0, {}
    n_oracle = int(sys.argv[3]) if len(sys.argv) > 3 else 0
    if n_oracle:
        from oracle import oracle as O
    for case in range(n):
        D = int(rng.choice([16, 32, 48, 64, 80, 96, 112, 128, 144, 160, 192, 256]))
        b = int(rng.choice([1, 3, 5, 5, 7, 9, 11, 13, 15]))
        if rng.random() < 0.25:   # the shapes with a register-ring variant of the vertical kernel
            D, b = int(rng.choice([96, 128])), int(rng.choice([7, 9, 11]))
        big = rng.random() < 0.05 and case >= n_oracle
        W = D + int(rng.integers(4, 1200 if big else 300))
        H = int(rng.integers(b + 2, 500 if big else 120))
        mind = int(rng.choice([0, 0, 0, -D + 1, -7, 5]))
        if rng.random() < 0.5:
            P1, P2 = 8 * 3 * b * b, 32 * 3 * b * b
        else:
            P1 = int(rng.integers(0, 3000))
            P2 = P1 + 1 + int(rng.integers(0, 12000))
        kw = dict(minDisparity=mind, numDisparities=D, blockSize=b, P1=P1, P2=P2, disp12MaxDiff=int(rng.choice([1, 2, -1, 1000000])),
                  uniquenessRatio=int(rng.choice([0, 5, 10, 15, 40, 100, 120])), speckleWindowSize=int(rng.choice([0, 50, 100])),
                  speckleRange=int(rng.choice([1, 2, 32])), preFilterCap=int(rng.choice([5, 31, 63])), mode=2)
        gl, gr = image_pair(rng, W, H, int(rng.integers(0, 4)))
        outs = {}
        for name, env, batch in [("default", {}, 1), ("signed", {"L3D_SGBM_UFORM": "0"}, 1), ("two-volume", {"L3D_SGBM_TONLY": "0"}, 1),
                                 ("batch", {}, 5), ("batch-signed", {"L3D_SGBM_UFORM": "0"}, 5),
                                 ("ring-in-lds", {"L3D_SGBM_RING_REGS": "0"}, 1), ("256-threads", {"L3D_SGBM_VERTICAL_THREADS": "256"}, 5)]:
            for k in ENV_KEYS:
                os.environ.pop(k, None)
            os.environ.update(env)
            try:
                m = stereo.StereoSGBM_create(**kw, max_batch=batch)
                outs[name] = m.compute(gl, gr)
            except Exception as e:   # a shape no variant is built for: the same answer from every route is fine
                outs[name] = repr(e)[:80]
                m = None
            del m
        ref = outs["default"]
        if isinstance(ref, str):
            errors[ref] = errors.get(ref, 0) + 1
        else:
            computed += 1
            if case < n_oracle:
                okw = {k: v for k, v in kw.items() if k != "mode"}
                want = O.sgbm_compute(O.sgbm_params(**okw), gl, gr)
                if not np.array_equal(want, ref):
                    bad += 1
                    print(f"ORACLE MISMATCH case {case}: W={W} H={H} {kw}", flush=True)
        for name, o in outs.items():
            same = (o == ref) if isinstance(ref, str) or isinstance(o, str) else np.array_equal(o, ref)
            if not same:
                bad += 1
                print(f"MISMATCH case {case} route {name}: W={W} H={H} {kw}", flush=True)
        if case % 100 == 99:
            print(f"{case + 1} cases, {bad} mismatches, {time.time() - t0:.0f} s", flush=True)
    for k in ENV_KEYS:
        os.environ.pop(k, None)
    print(f"done: {n} cases, {computed} computed, {bad} mismatches; refused: {errors}")
    write_summary("matcher_routes", {"script": "scripts/gpu_stress_forms.py", "seed": int(sys.argv[2]) if len(sys.argv) > 2 else 1, "cases": n,
                                     "computed": computed, "compared_with_cpu_restatement": min(n_oracle, n), "mismatches": bad,
                                     "routes": ["default", "signed recurrence", "two-volume layout", "batch handle", "batch handle, signed",
                                                "box-sum ring in LDS", "256-thread blocks"],
                                     "refused_shapes": errors, "seconds": round(time.time() - t0)})
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
