"""Generates tests/golden/pointcloud_golden.npz by running the reference's own PointCloudProcessor
(utils/point_cloud.py; pure numpy/scipy when Open3D is absent, as it is here) on points taken from the
point cloud the reference ships (output/pointcloud_20260115_202445.ply, binary little-endian, parsed here
by hand) and on seeded synthetic clouds.  Run once in the build container, where /root/reference exists:

    python tests/golden/make_pointcloud_golden.py

Only numeric inputs and outputs are written; no reference source is copied.
"""
import contextlib
import importlib.util
import io
import os
import sys

sys.dont_write_bytecode = True   # importing the reference must not leave a bytecode cache beside its sources

import numpy as np

REF = "/root/reference"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "pointcloud_golden.npz")


def read_ply_xyz(path):
    with open(path, "rb") as f:
        n = None
        while True:
            line = f.readline().decode("ascii").strip()
            if line.startswith("element vertex"):
                n = int(line.split()[-1])
            if line == "end_header":
                break
        rec = np.dtype([("x", "<f8"), ("y", "<f8"), ("z", "<f8"), ("r", "u1"), ("g", "u1"), ("b", "u1")])
        data = np.frombuffer(f.read(n * rec.itemsize), dtype=rec, count=n)
    return np.stack([data["x"], data["y"], data["z"]], axis=1)


def main():
    spec = importlib.util.spec_from_file_location("ref_point_cloud", os.path.join(REF, "utils", "point_cloud.py"))
    mod = importlib.util.module_from_spec(spec)
    with contextlib.redirect_stdout(io.StringIO()):
        spec.loader.exec_module(mod)
        proc = mod.PointCloudProcessor()
    assert not proc.try_open3d, "golden vectors are for the numpy/scipy path"
    rng = np.random.default_rng(20260211)
    real = read_ply_xyz(os.path.join(REF, "output", "pointcloud_20260115_202445.ply"))
    real = real[rng.choice(len(real), 3000, replace=False)]
    dense = rng.normal(0, 0.004, (2500, 3)) + [0.01, -0.02, 0.5]          # many points per 2 mm voxel
    grid = np.stack(np.meshgrid(np.arange(12) * 0.01, np.arange(10) * 0.01, [0.3, 0.31]), -1).reshape(-1, 3)   # equal neighbour distances
    dup = np.concatenate([real[:200], real[:50], grid[:30]])                # exact duplicates
    neg = rng.uniform(-0.05, 0.05, (800, 3))                                # negative coordinates, floor() not trunc()
    line = np.stack([np.arange(40) * 0.0078125, np.zeros(40), np.full(40, 0.25)], 1)   # exactly equal spacings: std == 0 inside
    out = {}
    with contextlib.redirect_stdout(io.StringIO()):
        for name, pts, voxel, nb, ratio in [("real", real, 0.002, 20, 2.0), ("dense", dense, 0.002, 20, 2.0), ("grid", grid, 0.015, 6, 1.0),
                                            ("dup", dup, 0.003, 8, 2.0), ("neg", neg, 0.01, 20, 0.5), ("few", real[:15], 0.002, 20, 2.0),
                                            ("line", line, 0.02, 2, 2.0), ("nb1", real[:300], 0.001, 1, 2.0)]:
            out[name + "_in"] = pts
            out[name + "_params"] = np.array([voxel, nb, ratio])
            out[name + "_voxel"] = np.asarray(proc.voxel_downsample(pts, voxel), np.float64).reshape(-1, 3)
            out[name + "_sor"] = np.asarray(proc.statistical_outlier_removal(pts, nb, ratio), np.float64).reshape(-1, 3)
            m = proc.compute_point_cloud_metrics(pts)
            out[name + "_metrics"] = np.concatenate([m["center"], m["std"], m["min"], m["max"], m["dimensions"], [m["volume"], m["avg_point_spacing"]]])
        merged = proc.merge_and_clean([real[:1000], np.zeros((0, 3)), dense[:1000], real[500:1500]], 0.004)
        out["merge_out"] = np.asarray(merged, np.float64).reshape(-1, 3)
        path = "/tmp/_golden_cloud.ply"
        proc.save_ply(real[:5], path, colors=np.array([[0, 255, 0], [1, 2, 3], [255, 255, 255], [9.7, 0.2, 200.9], [0, 0, 0]]))
        out["ply_text"] = np.frombuffer(open(path, "rb").read(), np.uint8)
        proc.save_ply(neg[:3] * 1e3, path)
        out["ply_text_nocolor"] = np.frombuffer(open(path, "rb").read(), np.uint8)
    np.savez_compressed(OUT, **out)
    print("wrote", OUT, os.path.getsize(OUT), "bytes;", {k: v.shape for k, v in out.items() if k.endswith(("_voxel", "_sor", "_out"))})


if __name__ == "__main__":
    main()
