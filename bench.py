#!/usr/bin/env python3
"""Benchmark of the stereo-depth + laser centre-line hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W

One "step" = one pass of the hot path over one batch of synthetic rectified
frame pairs that already sit in HBM:  BGR->gray (both views) -> SGBM-3WAY with the
reference's constructor parameters (D=64, block=5, uniqueness 10, LR check,
median, speckle filter) -> /16 + reprojectImageTo3D Z + masks -> Steger
(sigma=3) on the left BGR frame.  This is BASELINE.json's configs[1]
("640x360 D=64 block=5 + Steger") and the metric is stereo frame-pairs/s.

For N > 1 launch with torch.distributed.run (one rank per GPU).  Frames shard
across ranks with no data-path collective; rank 0 broadcasts the calibration (Q)
once at start-up.  Rank 0 prints ONE JSON line.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

CONFIGS = {
    # name: (W, H, D, block, use_wls, lambda, sigma)
    "C2": (640, 360, 64, 5, False, 8000.0, 1.5),
    "C3": (1280, 720, 96, 9, True, 10000.0, 1.5),
    "C5": (1920, 1080, 128, 11, True, 8000.0, 1.5),
    # what the reference's own constructor picks for a 640-px-wide eye (camera/single_usb_stereo_camera.py:239-282)
    "REFDEFAULT": (640, 360, 96, 7, True, 8000.0, 1.5),
}


def alg_bytes(W, H, D, use_wls, n_pts):
    """Algorithmic HBM bytes per frame pair: one SGBM pass = two u8 inputs + s16 output + one
    write and one read of the s16 cost volume = 4*W*H*(1+D); WLS adds a second pass + 103*W*H;
    reprojection 16*W*H; Steger 3*W*H + 8*n_pts."""
    sgbm = 4 * W * H * (1 + D)
    total = sgbm + 16 * W * H + 3 * W * H + 8 * n_pts
    if use_wls:
        total += sgbm + 103 * W * H
    return sgbm, total


def make_Q(W, H):
    # reprojection matrix of the reference's calibration (fx ~ 209 px at 320 px width, 62.7 mm baseline), scaled to W
    f = 209.02533991591997 * (W / 320.0)
    return np.array([[1, 0, 0, -(W - 1) / 2.0], [0, 1, 0, -(H - 1) / 2.0], [0, 0, 0, f], [0, 0, 15.941451590053976, 0]], np.float64)


def cpu_baseline(W, H, D, block, use_wls, lam, sigma, seconds_budget=12.0):
    """The CPU oracle (a port of the same algorithms, single thread) on a bounded sample of the workload."""
    from oracle import oracle as O
    from laser_3d_reconstruction_amd.utils import synthetic as S

    p = O.sgbm_params(numDisparities=D, blockSize=block)
    Q = make_Q(W, H)
    n, t_total = 0, 0.0
    while t_total < seconds_budget and n < 64:
        lb, rb, _ = S.stereo_pair(W, H, D, 1000 + n)
        t0 = time.perf_counter()
        gl, gr = O.bgr2gray(lb), O.bgr2gray(rb)
        if use_wls:
            pl = O.wls_left_matcher_params(p)
            dl = O.sgbm_compute(pl, gl, gr)
            dr = O.sgbm_compute(O.right_matcher_params(p), gr, gl)
            d = O.wls_filter(O.wls_params(p, lam, sigma), dl, dr, gl)
        else:
            d = O.sgbm_compute(p, gl, gr)
        O.depth_from_disp16(d, Q)
        O.steger_extract(lb, 3.0, 200)
        t_total += time.perf_counter() - t0
        n += 1
    model = ""
    try:
        with open("/proc/cpuinfo") as f:
            model = next((ln.split(":", 1)[1].strip() for ln in f if ln.startswith("model name")), "")
    except OSError:
        pass
    return {"value": n / t_total, "unit": "frame-pairs/s", "cores": 1, "kind": "port",
            "sample": f"{n} synthetic {W}x{H} D={D} pairs, same stages as the GPU step, one thread of the C oracle",
            "host": {"cpu_model": model, "os_cpu_count": os.cpu_count()}}


def make_rig(W, H):
    """A stereo rig of the reference's kind (stereo_calibration.json: ~63 mm baseline, mild radial distortion,
    a degree of relative rotation), scaled to the frame size."""
    s = W / 320.0
    K1 = np.array([[208.9 * s, 0, 166.3 * s], [0, 209.5 * s, H / 2 + 11.0 * s], [0, 0, 1]])
    K2 = np.array([[209.2 * s, 0, 174.0 * s], [0, 209.9 * s, H / 2 + 13.0 * s], [0, 0, 1]])
    R = np.array([[0.999955735517614, 0.000791832491916667, -0.009375500345710967],
                  [-0.0009647212415556836, 0.9998293125001468, -0.018450343585144066],
                  [0.009359290483459346, 0.01845857163456914, 0.9997858194707794]])
    return {"K1": K1, "D1": np.array([-0.0651, 0.1874, 0, 0, 0]), "K2": K2, "D2": np.array([-0.0435, 0.0895, 0, 0, 0]),
            "R": R, "T": np.array([-0.0627, -0.00015, -0.00043])}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=None, help="frame pairs per step per GPU (default: 256 for C2, 64 for C3, 16 for C5)")
    ap.add_argument("--config", default="C2", choices=sorted(CONFIGS))
    ap.add_argument("--pipelines", type=int, default=1, help="independent pipelines (matcher + buffers + streams) the steps alternate between")
    ap.add_argument("--separate-calls", action="store_true", help="gray, matcher and depth as three library calls instead of the fused one")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (no CPU fallback)")
    # one rank per GPU; L3D_BENCH_BACKEND=gloo lets several ranks share one GPU to rehearse the N > 1 path on a 1-GPU box
    backend = os.environ.get("L3D_BENCH_BACKEND", "nccl")
    local_rank = local_rank % torch.cuda.device_count() if backend != "nccl" else local_rank
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)

    from laser_3d_reconstruction_amd import ops, parallel, stereo
    from laser_3d_reconstruction_amd.core.laser_extractor import FastStegerExtractor
    from laser_3d_reconstruction_amd.utils import synthetic as S

    W, H, D, block, use_wls, lam, sigma = CONFIGS[args.config]
    B = args.batch if args.batch else {"C2": 256, "C3": 64, "C5": 16, "REFDEFAULT": 128}.get(args.config, 64)

    # start-up: rank 0 owns the calibration and broadcasts it (the only collective of the job)
    calib = dict(make_rig(W, H), Q=make_Q(W, H)) if rank == 0 else None
    calib = parallel.broadcast_calibration(calib, src=0, device=dev)
    Q = calib["Q"]

    # synthetic frames of this rank's shard, uploaded once
    n_distinct = min(B, 16)
    frames = [S.stereo_pair(W, H, D, k) for k in parallel.shard_indices(n_distinct * world, rank, world)]
    lbgr = np.stack([frames[i % n_distinct][0] for i in range(B)])
    rbgr = np.stack([frames[i % n_distinct][1] for i in range(B)])
    d_l = torch.from_numpy(lbgr).to(dev)
    d_r = torch.from_numpy(rbgr).to(dev)

    # A pipeline = matcher (with its work buffers), output buffers and a pair of streams.  Frame batches are
    # independent, so with --pipelines 2 consecutive steps alternate between two of them and step k+1 starts while
    # step k drains; measured no better than one (the matcher already overlaps two half batches inside), the default.
    def make_pipe():
        p = {}
        p["left"] = stereo.StereoSGBM_create(minDisparity=0, numDisparities=D, blockSize=block, P1=8 * 3 * block ** 2,
                                             P2=32 * 3 * block ** 2, disp12MaxDiff=1, uniquenessRatio=10, speckleWindowSize=100,
                                             speckleRange=32, preFilterCap=63, mode=stereo.STEREO_SGBM_MODE_SGBM_3WAY,
                                             device=local_rank, max_batch=B)
        p["right"] = p["wls"] = None
        if use_wls:
            p["right"] = stereo.createRightMatcher(p["left"])
            p["wls"] = stereo.createDisparityWLSFilter(p["left"])
            p["wls"].setLambda(lam)
            p["wls"].setSigmaColor(sigma)
        p["steger"] = FastStegerExtractor(3.0, 200, device=local_rank)
        p["gl"] = torch.empty((B, H, W), dtype=torch.uint8, device=dev)
        p["gr"] = torch.empty_like(p["gl"])
        p["disp"] = torch.empty((B, H, W), dtype=torch.int16, device=dev)
        p["dr"] = torch.empty_like(p["disp"])
        p["df"] = torch.empty_like(p["disp"])
        p["depth"] = torch.empty((B, H, W), dtype=torch.float32, device=dev)
        # the laser extractor works on the colour frame alone: it runs beside the matcher on a stream of its own
        p["main"] = torch.cuda.Stream()
        p["side"] = torch.cuda.Stream()
        return p

    cap = 4 * H  # points kept per frame (the synthetic ridge yields ~1.6 per row)
    pipes = [make_pipe() for _ in range(max(1, args.pipelines))]
    left, right, steger = pipes[0]["left"], pipes[0]["right"], pipes[0]["steger"]
    state = {}

    def step(k=0, beside=True):
        p = pipes[k % len(pipes)]
        with torch.cuda.stream(p["main"]):
            p["side"].wait_stream(p["main"])
            with torch.cuda.stream(p["side"] if beside else p["main"]):
                state["xy"], state["counts"] = p["steger"].extract_device(d_l, cap)
            if use_wls or args.separate_calls:
                ops.bgr2gray_device(d_l, p["gl"])
                ops.bgr2gray_device(d_r, p["gr"])
                p["left"].compute_device(p["gl"], p["gr"], p["disp"])
                d = p["disp"]
                if use_wls:
                    p["right"].compute_device(p["gr"], p["gl"], p["dr"])
                    d = p["wls"].filter_device(p["disp"], p["gl"], p["dr"], p["df"])
                ops.depth_from_disparity16_device(d, Q, p["depth"])
            else:
                # gray of both views -> matcher -> depth in one library call: the reprojection of each half batch runs
                # on the matcher's own streams beside the other half's kernels
                p["left"].compute_depth_device(d_l, d_r, Q, p["disp"], p["depth"])
            p["main"].wait_stream(p["side"])

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for k in range(max(args.warmup, len(pipes))):
        step(k)
    fence()
    t0 = time.perf_counter()
    for k in range(args.steps):
        step(k)
    fence()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev if backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # Per-kernel durations (HIP events inside the library, on the stream the kernels run on) come from a second
    # pass of the same K steps: in the timed pass the matcher runs a batch as two halves on two streams, whose
    # kernels overlap, so a single kernel has no duration of its own there; with the events switched on it
    # launches every kernel once per batch on one stream.
    left.profile(True)
    if right is not None:
        right.profile(True)
    for _ in range(args.steps):
        step(0, beside=False)   # nothing beside the kernel being timed
    fence()
    prof = left.profile_read()
    left.profile(False)
    if right is not None:
        right.profile(False)

    n_pts = float(state["counts"].float().mean().item())
    value = B * world * args.steps / elapsed

    # Rectification (the step before the path), outside the timed region: maps from the broadcast calibration,
    # remap of one eye of every frame with the gray plane written alongside
    from laser_3d_reconstruction_amd.camera import rectify
    R1, R2, P1, P2, _, _, _ = rectify.stereo_rectify(calib["K1"], calib["D1"], calib["K2"], calib["D2"], (W, H), calib["R"], calib["T"], alpha=0)
    rect = rectify.Rectifier.from_calibration(calib["K1"], calib["D1"], R1, P1, (W, H), device=local_rank)
    rect.remap_device(d_l, want_bgr=True, want_gray=True)
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    for _ in range(10):
        rect.remap_device(d_l, want_bgr=True, want_gray=True)
    torch.cuda.synchronize()
    remap_mpix = 10 * B * W * H / (time.perf_counter() - t1) / 1e6

    # Steger alone (Mpix/s over all pixels), outside the timed region
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    for _ in range(10):
        steger.extract_device(d_l, cap)
    torch.cuda.synchronize()
    steger_mpix = 10 * B * W * H / (time.perf_counter() - t1) / 1e6

    if rank == 0:
        sgbm_b, total_b = alg_bytes(W, H, D, use_wls, n_pts)
        calls = max(prof.pop("calls"), 1)
        kernel_ms = {k: v / calls for k, v in prof.items()}
        dom = max(kernel_ms, key=kernel_ms.get)
        # algorithmic bytes of one launch of each SGBM kernel (B frame pairs): the vertical kernel reads the two
        # images and writes the cost volume once, the horizontal kernel reads it once and writes the disparity
        per_kernel_bytes = {
            "k_prefilter": 2 * W * H * B,
            "k_cost_vertical": (2 * W * H + 2 * W * H * D) * B,
            "k_horizontal": (2 * W * H * D + 2 * W * H) * B,
            "k_lrcheck": 2 * W * H * B,
            "k_median3": 4 * W * H * B,
            "k_speckle_*": 4 * W * H * B,
        }
        achieved = per_kernel_bytes[dom] / (kernel_ms[dom] * 1e-3) / 1e9
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "traffic.json")
        if os.path.exists(tpath):
            try:
                tj = json.load(open(tpath))
                traffic = tj.get(args.config, {}).get(dom)
                if traffic is not None:   # recorded per launch at the batch of the profiling run
                    traffic = traffic * B / tj.get(args.config + "_batch", 64)
            except Exception:
                traffic = None
        out = {
            "metric": "stereo frame-pairs/sec @640x360 D=64" if args.config == "C2" else f"stereo frame-pairs/sec {args.config}",
            "value": value, "unit": "frame-pairs/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "s16", "data": "synthetic",
            "config": {"workload": f"{args.config}: {W}x{H} rectified pairs, SGBM-3WAY D={D} block={block}"
                                   + (f" + right matcher + WLS lambda={lam:g}" if use_wls else "")
                                   + " + depth + Steger sigma=3.0", "batch_per_gpu": B, "sharding": f"frames x{world}",
                       "pipelines_in_flight": len(pipes)},
            "roofline": {"bound": "hbm", "kernel": dom, "achieved": achieved, "peak": 8000.0, "unit": "GB/s",
                         "frac": achieved / 8000.0, "traffic": traffic,
                         "timing": "HIP events around each kernel, second pass of the same steps with whole-batch launches on one stream"},
            "pipeline_roofline": {"alg_bytes_per_pair": total_b, "achieved_GBps": total_b * value / world / 1e9,
                                  "frac": total_b * value / world / 8e12},
            "kernel_ms_per_launch": kernel_ms,
            "steger_mpix_per_s": steger_mpix, "steger_points_per_frame": n_pts,
            "remap_mpix_per_s": remap_mpix,
        }
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline(W, H, D, block, use_wls, lam, sigma)
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
