"""Differential stress of the smoother's two horizontal-pass routes on the GPU (development aid, not part of the test suite):
random sizes, disparity ranges, batches, lambda and sigma; the row-sweep route and the transposing route must give the same
filtered disparity and confidence, and the first N cases are also compared with the CPU restatement.
usage: python scripts/gpu_stress_wls.py [cases] [seed] [cases checked against the oracle]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from laser_3d_reconstruction_amd import stereo  # noqa: E402
from laser_3d_reconstruction_amd.utils import synthetic as S  # noqa: E402


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
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 100
    rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
    n_oracle = int(sys.argv[3]) if len(sys.argv) > 3 else 0
    if n_oracle:
        from oracle import oracle as O
    t0, bad, rows_route = time.time(), 0, 0
    for case in range(n):
        D = int(rng.choice([16, 32, 64, 96, 128]))
        b = int(rng.choice([3, 5, 7, 9, 11]))
        small = case < n_oracle
        W = D + 4 * int(rng.integers(1, 40 if small else 120)) + int(rng.choice([0, 0, 0, 1, 2, 3]))
        H = int(rng.integers(4 * b + 8, 90 if small else 260))
        B = int(rng.integers(1, 4))
        lam, sigma = float(rng.choice([8000.0, 10000.0, 500.0, 80000.0])), float(rng.choice([1.5, 0.8, 3.0]))
        frames = [S.stereo_pair(W, H, D, int(rng.integers(0, 1 << 30))) for _ in range(B)]
        gl = np.stack([f[0][:, :, 1] for f in frames])
        gr = np.stack([f[1][:, :, 1] for f in frames])
        kw = dict(minDisparity=0, numDisparities=D, blockSize=b, P1=8 * 3 * b * b, P2=32 * 3 * b * b, disp12MaxDiff=1, uniquenessRatio=10,
                  speckleWindowSize=100, speckleRange=32, preFilterCap=63, mode=2)
        outs = {}
        for route in ("0", "1"):
            os.environ["L3D_WLS_TRANSPOSE"] = route
            left = stereo.StereoSGBM_create(**kw, max_batch=B)
            right = stereo.createRightMatcher(left)
            wls = stereo.createDisparityWLSFilter(left)
            wls.setLambda(lam)
            wls.setSigmaColor(sigma)
            dl, dr = left.compute(gl, gr), right.compute(gr, gl)
            outs[route] = (wls.filter(dl, gl, disparity_map_right=dr), wls.getConfidenceMap())
        rows_route += (W - D) % 4 == 0
        if not (np.array_equal(outs["0"][0], outs["1"][0]) and np.array_equal(outs["0"][1], outs["1"][1])):
            bad += 1
            print(f"ROUTE MISMATCH case {case}: W={W} H={H} D={D} b={b} B={B} lambda={lam} sigma={sigma}", flush=True)
        if small:
            p = O.sgbm_params(**{k: v for k, v in kw.items() if k != "mode"})
            for k in range(B):
                want = O.wls_filter(O.wls_params(p, lam, sigma), dl[k], dr[k], gl[k])
                if not np.array_equal(outs["0"][0][k], want):
                    bad += 1
                    print(f"ORACLE MISMATCH case {case} frame {k}: W={W} H={H} D={D} b={b} lambda={lam} sigma={sigma}", flush=True)
        if case % 50 == 49:
            print(f"{case + 1} cases, {bad} mismatches, {time.time() - t0:.0f} s", flush=True)
    os.environ.pop("L3D_WLS_TRANSPOSE", None)
    print(f"done: {n} cases ({rows_route} with a sub-image width that takes the row sweeps), {bad} mismatches")
    write_summary("smoother_routes", {"script": "scripts/gpu_stress_wls.py", "seed": int(sys.argv[2]) if len(sys.argv) > 2 else 1, "cases": n,
                                      "on_row_sweep_route": int(rows_route), "compared_with_cpu_restatement": min(n_oracle, n), "mismatches": bad,
                                      "compared": "filtered disparity and confidence map, row-sweep route (four-pixel confidence kernel) against "
                                                  "transposing route (one-pixel confidence kernel)", "seconds": round(time.time() - t0)})
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
