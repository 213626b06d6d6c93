#!/usr/bin/env python3
"""Benchmark of the stereo-depth + laser centre-line hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W [--config C2|C3|C4|C5|REFDEFAULT]

One "step" = one pass of the reference's frame loop (main.py:164-189) over one batch of synthetic rectified frame
pairs that already sit in HBM: BGR->gray (both views) -> SGBM-3WAY with the reference's constructor parameters
[-> right matcher -> WLS] -> /16 + reprojectImageTo3D + masks, Steger (sigma = 3) on the left BGR frame, and
reconstruct_from_depth of the laser pixels through the depth map.  The default workload is BASELINE.json's
configs[1] ("640x360 D=64 block=5 + Steger"); the metric is stereo frame-pairs/s.

--gpus N > 1: when the process was not started by torch.distributed.run it starts N ranks itself
(`python -m torch.distributed.run --nproc-per-node N bench.py ...`) as a child process, before anything touches the
GPU, and exits with the child's status.  Frames shard across ranks with no data-path collective; rank 0 broadcasts
the calibration (rectification maps included) once at start-up.  Rank 0 prints ONE JSON line.
"""
from __future__ import annotations

import argparse
import json
import os
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

CONFIGS = {
    # name: (W, H, D, block, use_wls, lambda, sigma)
    "C2": (640, 360, 64, 5, False, 8000.0, 1.5),
    "C3": (1280, 720, 96, 9, True, 10000.0, 1.5),
    # 8 independent camera streams of the C2 shape, stream s on rank s % N: a fixed job (strong scaling)
    "C4": (640, 360, 64, 5, False, 8000.0, 1.5),
    # + XYZ reprojection and the plane / refraction reconstruction of the laser pixels (core/reconstruction.py:30-119)
    "C5": (1920, 1080, 128, 11, True, 8000.0, 1.5),
    # what the reference's own constructor picks for a 640-px-wide eye (camera/single_usb_stereo_camera.py:239-282)
    "REFDEFAULT": (640, 360, 96, 7, True, 8000.0, 1.5),
}
DEFAULT_BATCH = {"C2": 256, "C3": 192, "C5": 64, "REFDEFAULT": 128}     # C3 / C5: +3 to 4 % over 128 / 32 (longer launches, shorter tails); C3 at 256 needs 240 of the 288 GB
C4_STREAMS = 8
C4_FRAMES_PER_RANK = 256     # a rank batches ceil(256 / streams it owns) consecutive frames of each of its streams per step
ALSO_CONFIGS = ("C3", "C5", "REFDEFAULT")
LASER_PLANE = np.array([0.0, 0.0, 1.0, -0.5])   # z = 0.5 m, the plane of the reference-generated golden vectors


def alg_bytes(W, H, D, use_wls, n_pts, xyz=False):
    """Algorithmic HBM bytes per frame pair: one SGBM pass = two u8 inputs + s16 output + one write and one read of
    the s16 cost volume = 4*W*H*(1+D); WLS adds a second pass + 103*W*H; depth: 16*W*H when the XYZ image is written
    (f32 disparity in, XYZ out), 6*W*H when only Z is (s16 disparity in, f32 Z out); Steger 3*W*H + 8*n_pts."""
    sgbm = 4 * W * H * (1 + D)
    total = sgbm + (16 if xyz else 6) * W * H + 3 * W * H + 8 * n_pts
    if use_wls:
        total += sgbm + 103 * W * H
    return sgbm, total


def make_Q(W, H):
    # reprojection matrix of the reference's calibration (fx ~ 209 px at 320 px width, 62.7 mm baseline), scaled to W
    f = 209.02533991591997 * (W / 320.0)
    return np.array([[1, 0, 0, -(W - 1) / 2.0], [0, 1, 0, -(H - 1) / 2.0], [0, 0, 0, f], [0, 0, 15.941451590053976, 0]], np.float64)


def make_K(W, H):
    f = 209.02533991591997 * (W / 320.0)
    return np.array([[f, 0, (W - 1) / 2.0], [0, f, (H - 1) / 2.0], [0, 0, 1.0]])


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


# ------------------------------------------------------------------------------------------------------------------
# launcher: --gpus N without a torchrun environment starts the ranks as a child process
# ------------------------------------------------------------------------------------------------------------------
def launcher_command(gpus: int, argv, port: int | None = None):
    """The command that runs bench.py as `gpus` ranks on this node (argv = the arguments bench.py was given)."""
    port = port if port is not None else 29500 + (os.getpid() % 2000)
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={gpus}", "--master-addr", "127.0.0.1",
            "--master-port", str(port), os.path.abspath(__file__)] + list(argv)


def spawn_ranks_if_needed(args, argv, environ=os.environ, run=subprocess.call):
    """Returns the child's exit status when this process is only the launcher, None when it is a rank (or N = 1)."""
    if args.gpus <= 1 or "WORLD_SIZE" in environ:
        return None
    env = dict(environ)
    env.setdefault("MASTER_ADDR", "127.0.0.1")
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    return run(launcher_command(args.gpus, argv), env=env)


def plan_frames(config, batch, rank, world):
    """(frames per step of this rank, their ids, scaling, sharding).  C4 = whole streams (stream s on rank s % world): the
    frames of a stream are independent of each other, so a rank takes ceil(256 / streams it owns) consecutive frames of
    each of its streams into one launch -- every rank launches about 256 frames whatever N is, and the frames per step of
    the whole job grow with N (weak scaling over a fixed set of 8 streams).  Otherwise frames round-robin."""
    from laser_3d_reconstruction_amd import parallel

    if config == "C4":
        my_streams = parallel.shard_indices(C4_STREAMS, rank, world)
        if not my_streams:
            raise SystemExit(f"rank {rank} owns no stream: C4 needs --gpus <= {C4_STREAMS}")
        per_stream = batch if batch else -(-C4_FRAMES_PER_RANK // len(my_streams))
        ids = [10_000 * (s + 1) + i for s in my_streams for i in range(per_stream)]
        return per_stream * len(my_streams), ids, "weak", (f"{C4_STREAMS} streams over {world} rank(s), stream s on rank s % {world}, "
                                                           f"{per_stream} frames per stream per step")
    B = batch if batch else DEFAULT_BATCH.get(config, 64)
    return B, parallel.shard_indices(B * world, rank, world), "weak", f"frames x{world}"


def under_profiler(environ=os.environ):
    """True when a ROCm profiler has been preloaded into this process (rocprofv3 sets ROCPROF* / ROCP_* variables and
    LD_PRELOAD): the GPU is then initialised before Python starts, so this process must not start other programs, and
    it does not fork workers either."""
    pre = environ.get("LD_PRELOAD", "")
    return "rocprof" in pre or any(k.startswith(("ROCPROF", "ROCP_")) for k in environ)


def _pair_job(job):
    from laser_3d_reconstruction_amd.utils import synthetic as S

    W, H, D, k = job
    return S.stereo_pair(W, H, D, k)[:2]


def host_frames(W, H, D, frame_ids):
    """The synthetic pairs of this rank, every frame distinct, generated by worker processes (forked: call before this
    process touches the GPU).  Returns two uint8 arrays (B, H, W, 3)."""
    import multiprocessing as mp

    workers = max(1, min(int(os.environ.get("OMP_NUM_THREADS", "16") or 16), os.cpu_count() or 1, len(frame_ids)))
    if under_profiler():
        workers = 1
    jobs = [(W, H, D, k) for k in frame_ids]
    if workers > 1:
        try:
            with mp.get_context("fork").Pool(workers) as pool:
                pairs = pool.map(_pair_job, jobs, chunksize=max(1, len(jobs) // (4 * workers)))
        except Exception:
            pairs = [_pair_job(j) for j in jobs]
    else:
        pairs = [_pair_job(j) for j in jobs]
    return np.stack([p[0] for p in pairs]), np.stack([p[1] for p in pairs])


def also_configs(steps=5, warmup=2, repeats=3, run=subprocess.run):
    """The other single-GPU workloads as short runs of this script, one child process each, before this process
    touches the GPU.  Returns {config: summary of the child's JSON line}."""
    out = {}
    for cfg in ALSO_CONFIGS:
        cmd = [sys.executable, os.path.abspath(__file__), "--config", cfg, "--steps", str(steps), "--warmup", str(warmup),
               "--repeats", str(repeats), "--no-cpu-baseline", "--no-also"]
        t0 = time.perf_counter()
        try:
            r = run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=150, text=True)
            line = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
            if r.returncode != 0 or not line:
                out[cfg] = {"error": f"exit {r.returncode}: " + (r.stderr or "")[-300:]}
                continue
            d = json.loads(line[-1])
            out[cfg] = {"value": d["value"], "unit": d["unit"], "ms_per_step": d["ms_per_step"], "steps": d["steps"],
                        "repeats": d["timing"]["repeats"], "batch_per_gpu": d["config"]["batch_per_gpu"],
                        "workload": d["config"]["workload"], "verified": d["verified"],
                        "roofline": {k: d["roofline"][k] for k in ("kernel", "achieved", "frac", "traffic")},
                        "pipeline_roofline_frac": d["pipeline_roofline"]["frac"],
                        "kernel_ms_per_launch": {k: v for k, v in d["kernel_ms_per_launch"].items()
                                                 if "vertical" in k or "horizontal" in k},
                        "stage_ms_per_step_serialised": d["stage_ms_per_step_serialised"],
                        "wall_s": time.perf_counter() - t0}
        except Exception as e:      # a failed side run must not lose the headline measurement
            out[cfg] = {"error": repr(e)}
    return out


# ------------------------------------------------------------------------------------------------------------------
# CPU baseline: the C restatement of the same algorithms (oracle/), the reference's frame-loop order
# ------------------------------------------------------------------------------------------------------------------
def _cpu_frame(job):
    """One frame of the reference's frame loop on the CPU restatement; returns the seconds it took."""
    W, H, D, block, use_wls, lam, sigma, xyz, k = job
    from oracle import oracle as O
    from laser_3d_reconstruction_amd.utils import synthetic as S

    lb, rb, _ = S.stereo_pair(W, H, D, 1000 + k)
    p = O.sgbm_params(numDisparities=D, blockSize=block)
    Q, K = make_Q(W, H), make_K(W, H)
    t0 = time.perf_counter()
    gl, gr = O.bgr2gray(lb), O.bgr2gray(rb)
    if use_wls:
        dl = O.sgbm_compute(O.wls_left_matcher_params(p), gl, gr)
        dr = O.sgbm_compute(O.right_matcher_params(p), gr, gl)
        d = O.wls_filter(O.wls_params(p, lam, sigma), dl, dr, gl)
    else:
        d = O.sgbm_compute(p, gl, gr)
    depth = O.depth_from_disp16(d, Q, want_xyz=True)[0] if xyz else O.depth_from_disp16(d, Q)
    pts = O.steger_extract(lb, 3.0, 200)
    if len(pts):
        O.reconstruct_from_depth(np.asarray(pts, np.float64), depth, K)
        if xyz:
            O.reconstruct_plane(np.asarray(pts, np.float64), K, LASER_PLANE, True, 1.33)
    return time.perf_counter() - t0


def cpu_baseline(W, H, D, block, use_wls, lam, sigma, xyz, seconds_budget=10.0):
    """Two legs over the same kind of sample: one thread, and one worker process per host core available to this job.
    Runs before the GPU is touched (the workers are forked)."""
    import multiprocessing as mp
    from oracle import oracle as O

    O.build()
    job = (W, H, D, block, use_wls, lam, sigma, xyz)
    # one thread
    n1, t1 = 0, 0.0
    while t1 < seconds_budget and n1 < 64:
        t1 += _cpu_frame(job + (n1,))
        n1 += 1
    per_frame = t1 / n1
    # all cores: the machine's share for one GPU (OMP_NUM_THREADS is set to it on the GPU box), at most what the OS reports
    cores = max(1, min(int(os.environ.get("OMP_NUM_THREADS", "16") or 16), os.cpu_count() or 1))
    n_all = int(max(cores, min(64 * cores, cores * round(seconds_budget / per_frame))))
    all_rate = None
    try:
        with mp.get_context("fork").Pool(cores) as pool:
            pool.map(_cpu_frame, [job + (10_000 + i,) for i in range(cores)])          # load the library in every worker
            t0 = time.perf_counter()
            pool.map(_cpu_frame, [job + (20_000 + i,) for i in range(n_all)], chunksize=1)
            all_rate = n_all / (time.perf_counter() - t0)
    except Exception as e:                                                                # report the one-thread leg alone
        all_rate = None
        err = repr(e)
    model = ""
    try:
        with open("/proc/cpuinfo") as f:
            model = next((ln.split(":", 1)[1].strip() for ln in f if ln.startswith("model name")), "")
    except OSError:
        pass
    out = {"value": all_rate if all_rate else n1 / t1, "unit": "frame-pairs/s", "cores": cores if all_rate else 1, "kind": "port",
           "sample": f"{n_all if all_rate else n1} synthetic {W}x{H} D={D} pairs through the C restatement of the same stages "
                     "(oracle/l3d_oracle.c; OpenCV itself is not installed), reference frame-loop order "
                     "(depth -> extract_centerline -> reconstruct_from_depth), one worker process per core",
           "one_thread": {"value": n1 / t1, "cores": 1, "frames": n1},
           "host": {"cpu_model": model, "os_cpu_count": os.cpu_count(), "workers": cores}}
    if not all_rate:
        out["all_cores_error"] = err
    return out


def main(argv=None):
    argv = sys.argv[1:] if argv is None else argv
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=40)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--repeats", type=int, default=5, help="the K timed steps are run this many times; the median run is reported")
    ap.add_argument("--batch", type=int, default=None, help="frame pairs per step per GPU (default: 256 for C2, 192 for C3, 64 for C5, 128 for REFDEFAULT; C4: 256 frames a rank over the streams it owns)")
    ap.add_argument("--config", default="C2", choices=sorted(CONFIGS))
    ap.add_argument("--pipelines", type=int, default=2,
                    help="independent pipelines (matchers + buffers + streams) the steps alternate between, so that the tail of one "
                         "batch runs beside the head of the next (+2 %% at C2; with the right matcher + WLS the smoother's line sweeps "
                         "of one batch run beside the other batch's matchers: +10 %%)")
    ap.add_argument("--separate-calls", action="store_true", help="gray, matcher and depth as three library calls instead of the fused one")
    ap.add_argument("--steger-stream", default="main", choices=["side", "main"],
                    help="main: the laser extractor runs ahead of the matcher on the step's stream; side: beside it on a stream of its own "
                         "(streams share a few hardware queues: on a side stream it can end up queued in front of one half batch and delay it)")
    ap.add_argument("--sgbm-mode", type=int, default=2, choices=[0, 1, 2],
                    help="2: MODE_SGBM_3WAY, the reference's mode and the benchmark; 0 / 1: MODE_SGBM / MODE_HH (5 / 8 paths), GPU leg only")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-also", action="store_true",
                    help="skip the short runs of C3 / C5 / REFDEFAULT that the default C2 run on one GPU adds as \"also\"")
    ap.add_argument("--no-verify", action="store_true")
    args = ap.parse_args(argv)

    rc = spawn_ranks_if_needed(args, argv)
    if rc is not None:
        return rc

    # stdout carries the one JSON line and nothing else: whatever libraries print on the way (RCCL announces its
    # version on file descriptor 1 when a communicator is created) goes to stderr
    sys.stdout.flush()
    json_out = os.fdopen(os.dup(1), "w")
    os.dup2(2, 1)

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    W, H, D, block, use_wls, lam, sigma = CONFIGS[args.config]
    want_xyz = args.config == "C5"

    # The other workloads first, as child processes of their own (this process has not touched the GPU yet), then the
    # CPU leg and the synthetic frames, whose worker processes are forked: all before this process holds a GPU.
    also = None
    if (rank == 0 and world == 1 and "WORLD_SIZE" not in os.environ and args.config == "C2" and args.sgbm_mode == 2 and not args.no_also
            and not under_profiler()):
        also = also_configs()
    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline and args.sgbm_mode == 2 and not under_profiler():
        try:
            cpu = cpu_baseline(W, H, D, block, use_wls, lam, sigma, want_xyz)
        except Exception as e:
            cpu = {"error": repr(e)}

    B, frame_ids, scaling, sharding = plan_frames(args.config, args.batch, rank, world)
    np_l, np_r = host_frames(W, H, D, frame_ids)

    import torch
    import torch.distributed as dist

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (no CPU fallback)")
    # one rank per GPU; L3D_BENCH_BACKEND=gloo lets several ranks share one GPU to rehearse the N > 1 path on a 1-GPU box
    backend = os.environ.get("L3D_BENCH_BACKEND", "nccl")
    local_rank = local_rank % torch.cuda.device_count() if backend != "nccl" else local_rank
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    # a process group also for a single rank when it was started by torch.distributed.run: the collectives of the
    # N > 1 path (broadcast, barrier, all_reduce, all_gather over RCCL) then run on a one-GPU box as well
    use_dist = world > 1 or ("WORLD_SIZE" in os.environ and "MASTER_PORT" in os.environ)
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)

    from laser_3d_reconstruction_amd import ops, parallel, stereo
    from laser_3d_reconstruction_amd.camera import rectify
    from laser_3d_reconstruction_amd.core.laser_extractor import FastStegerExtractor
    from laser_3d_reconstruction_amd.core.reconstruction import Reconstructor
    from laser_3d_reconstruction_amd.utils import synthetic as S


    # start-up: rank 0 owns the calibration -- K/D/R/T, Q and the four rectification maps -- and broadcasts it (the
    # only collective of the job).  Timed twice, with and without the maps in the payload.
    calib = None
    if rank == 0:
        rig = make_rig(W, H)
        R1, R2, P1, P2, _, _, _ = rectify.stereo_rectify(rig["K1"], rig["D1"], rig["K2"], rig["D2"], (W, H), rig["R"], rig["T"], alpha=0)
        rl = rectify.Rectifier.from_calibration(rig["K1"], rig["D1"], R1, P1, (W, H), device=local_rank)
        rr = rectify.Rectifier.from_calibration(rig["K2"], rig["D2"], R2, P2, (W, H), device=local_rank)
        calib = dict(rig, Q=make_Q(W, H), K=make_K(W, H), map_left_x=rl.maps[0], map_left_y=rl.maps[1],
                     map_right_x=rr.maps[0], map_right_y=rr.maps[1])
    bcast_ms = {}
    if use_dist:
        small = {k: v for k, v in calib.items() if not k.startswith("map_")} if rank == 0 else None
        torch.cuda.synchronize(); dist.barrier(); t0 = time.perf_counter()
        parallel.broadcast_calibration(small, src=0, device=dev)
        torch.cuda.synchronize(); dist.barrier(); bcast_ms["kdrt_q_only"] = (time.perf_counter() - t0) * 1e3
        t0 = time.perf_counter()
    calib = parallel.broadcast_calibration(calib, src=0, device=dev)
    if use_dist:
        torch.cuda.synchronize(); dist.barrier(); bcast_ms["with_rectification_maps"] = (time.perf_counter() - t0) * 1e3
        bcast_ms["payload_bytes_with_maps"] = int(sum(np.asarray(v).nbytes for v in calib.values()))
    Q, K = calib["Q"], calib["K"]
    rect = rectify.Rectifier.from_maps(calib["map_left_x"], calib["map_left_y"], device=local_rank)

    # synthetic frames of this rank's shard, every frame of the batch distinct, uploaded once
    h_l = torch.from_numpy(np_l).pin_memory()
    h_r = torch.from_numpy(np_r).pin_memory()
    del np_l, np_r
    d_l, d_r = h_l.to(dev), h_r.to(dev)

    recon = Reconstructor(K, LASER_PLANE, use_refraction_correction=True, device=local_rank)

    # A pipeline = matcher (with its work buffers), output buffers and a pair of streams.
    def make_pipe(batch):
        p = {}
        p["left"] = stereo.StereoSGBM_create(minDisparity=0, numDisparities=D, blockSize=block, P1=8 * 3 * block ** 2,
                                             P2=32 * 3 * block ** 2, disp12MaxDiff=1, uniquenessRatio=10, speckleWindowSize=100,
                                             speckleRange=32, preFilterCap=63, mode=args.sgbm_mode,
                                             device=local_rank, max_batch=batch)
        p["right"] = p["wls"] = None
        if use_wls:
            p["right"] = stereo.createRightMatcher(p["left"])
            p["wls"] = stereo.createDisparityWLSFilter(p["left"])
            p["wls"].setLambda(lam)
            p["wls"].setSigmaColor(sigma)
        p["steger"] = FastStegerExtractor(3.0, 200, device=local_rank)
        p["gl"] = torch.empty((batch, H, W), dtype=torch.uint8, device=dev)
        p["gr"] = torch.empty_like(p["gl"])
        p["disp"] = torch.empty((batch, H, W), dtype=torch.int16, device=dev)
        p["dr"] = torch.empty_like(p["disp"])
        p["df"] = torch.empty_like(p["disp"])
        p["depth"] = torch.empty((batch, H, W), dtype=torch.float32, device=dev)
        p["xyz"] = torch.empty((batch, H, W, 3), dtype=torch.float32, device=dev) if want_xyz else None
        # the laser extractor works on the colour frame alone: it runs beside the matcher on a stream of its own
        p["main"] = torch.cuda.Stream()
        p["side"] = torch.cuda.Stream()
        return p

    cap = 4 * H  # points kept per frame (the synthetic ridge yields ~1.6 per row)
    pipes = [make_pipe(B) for _ in range(max(1, args.pipelines))]
    left, right = pipes[0]["left"], pipes[0]["right"]
    state = {}
    stage_ev = None   # profile pass: torch events around the stages

    def mark(name):
        if stage_ev is not None:
            e = torch.cuda.Event(enable_timing=True)
            e.record()
            stage_ev.append((name, e))

    def step(k=0, beside=True, p=None, L=None, R=None):
        p = p if p is not None else pipes[k % len(pipes)]
        L = d_l if L is None else L
        R = d_r if R is None else R
        with torch.cuda.stream(p["main"]):
            mark("start")
            p["side"].wait_stream(p["main"])
            with torch.cuda.stream(p["side"] if (beside and args.steger_stream == "side") else p["main"]):
                xy, counts = p["steger"].extract_device(L, cap)
            mark("steger")
            if use_wls or args.separate_calls or want_xyz:
                ops.bgr2gray_device(L, p["gl"][: L.shape[0]])
                ops.bgr2gray_device(R, p["gr"][: L.shape[0]])
                mark("gray")
                d = p["left"].compute_device(p["gl"][: L.shape[0]], p["gr"][: L.shape[0]], p["disp"][: L.shape[0]])
                mark("sgbm_left")
                if use_wls:
                    dr = p["right"].compute_device(p["gr"][: L.shape[0]], p["gl"][: L.shape[0]], p["dr"][: L.shape[0]])
                    mark("sgbm_right")
                    d = p["wls"].filter_device(d, p["gl"][: L.shape[0]], dr, p["df"][: L.shape[0]])
                    mark("wls")
                depth = p["depth"][: L.shape[0]]
                ops.depth_from_disparity16_device(d, Q, depth, p["xyz"][: L.shape[0]] if want_xyz else None)
                mark("depth")
            else:
                # gray of both views -> matcher -> depth in one library call: the reprojection of each half batch runs
                # on the matcher's own streams beside the other half's kernels
                d, depth = p["left"].compute_depth_device(L, R, Q, p["disp"][: L.shape[0]], p["depth"][: L.shape[0]])
                mark("gray+sgbm_left+depth")
            p["main"].wait_stream(p["side"])
            # the reference's frame loop: laser pixels -> 3-D through the depth map (main.py:176)
            pts, keep = recon.reconstruct_from_depth_device(xy, counts, depth)
            plane_pts = recon.reconstruct_laser_line_device(xy, counts) if want_xyz else None
            mark("reconstruct")
        state.update(xy=xy, counts=counts, disp=d, depth=depth, pts=pts, keep=keep, plane_pts=plane_pts)

    def fence():
        torch.cuda.synchronize()
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    for k in range(max(args.warmup, len(pipes))):
        step(k)
    # K timed steps bracketed by barrier + synchronize, MAX over ranks; repeated, the median run is the result
    runs = []
    for _ in range(max(1, args.repeats)):
        fence()
        t0 = time.perf_counter()
        for k in range(args.steps):
            step(k)
        fence()
        elapsed = time.perf_counter() - t0
        if use_dist:
            t = torch.tensor([elapsed], dtype=torch.float64, device=dev if backend == "nccl" else "cpu")
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            elapsed = float(t.item())
        runs.append(elapsed)
    elapsed = float(np.median(runs))
    frames_all = parallel.gather_counts(float(B), device=dev)          # frames per step of every rank
    total_per_step = sum(frames_all)
    value = total_per_step * args.steps / elapsed

    # ---- verification of the timed workload's own outputs: first and last frame of the batch against a handle built
    # for single frames (another kernel variant), every stage bit for bit
    verified = None
    if not args.no_verify:
        torch.cuda.synchronize()
        one = make_pipe(1)
        got = {k: (v.clone() if v is not None else None) for k, v in state.items()}
        verified = True
        for f in (0, B - 1):
            step(0, beside=False, p=one, L=d_l[f:f + 1], R=d_r[f:f + 1])
            torch.cuda.synchronize()
            n = int(state["counts"][0].item())
            ok = (torch.equal(state["disp"][0], got["disp"][f]) and torch.equal(state["depth"][0], got["depth"][f])
                  and n == int(got["counts"][f].item()) and torch.equal(state["xy"][0, :n], got["xy"][f, :n])
                  and torch.equal(state["keep"][0, :n], got["keep"][f, :n]))
            kp = state["keep"][0, :n].bool()
            ok = ok and torch.equal(state["pts"][0, :n][kp], got["pts"][f, :n][kp])
            if want_xyz:
                a, b2 = state["plane_pts"][0, :n], got["plane_pts"][f, :n]
                ok = ok and torch.equal(torch.isnan(a), torch.isnan(b2)) and torch.equal(torch.nan_to_num(a), torch.nan_to_num(b2))
            ok = ok and n > H // 2 and float((got["depth"][f] > 0).float().mean()) > 0.2
            verified = verified and bool(ok)
        del one, got
        step(0)
        torch.cuda.synchronize()

    # ---- per-kernel and per-stage durations: a second pass of the same K steps.  In the timed pass the matcher runs a
    # batch as two halves on two streams, whose kernels overlap, so a single kernel has no duration of its own there;
    # with the events switched on it launches every kernel once per batch on one stream, nothing beside it.
    left.profile(True)
    if right is not None:
        right.profile(True)
    stage_tot, stage_n = {}, 0
    for _ in range(min(args.steps, 20)):
        stage_ev = []
        step(0, beside=False)
        torch.cuda.synchronize()
        for (n0, e0), (n1, e1) in zip(stage_ev[:-1], stage_ev[1:]):
            stage_tot[n1] = stage_tot.get(n1, 0.0) + e0.elapsed_time(e1)
        stage_n += 1
    stage_ev = None
    prof = left.profile_read()
    left.profile(False)
    prof_r = None
    if right is not None:
        prof_r = right.profile_read()
        right.profile(False)
    stage_ms = {k: v / stage_n for k, v in stage_tot.items()}

    n_pts = float(state["counts"].float().mean().item())

    # ---- H2D-inclusive rate: the same step fed from pinned host memory (never `value`).  Two input buffers, so that
    # the frames of step i + 1 are copied (on a stream of their own) while step i runs; the host orders the two: it
    # starts a copy once the step that last read that buffer is done, and a step once its copy is done.
    up = torch.cuda.Stream()
    bufs = [(d_l, d_r), (torch.empty_like(d_l), torch.empty_like(d_r))]
    freed = [None, None]

    def fed_steps(n):
        for i in range(n):
            j = i % 2
            if freed[j] is not None:
                freed[j].synchronize()
            with torch.cuda.stream(up):
                bufs[j][0].copy_(h_l, non_blocking=True)
                bufs[j][1].copy_(h_r, non_blocking=True)
            up.synchronize()
            pp = pipes[i % len(pipes)]
            step(i, L=bufs[j][0], R=bufs[j][1])
            freed[j] = torch.cuda.Event()
            freed[j].record(pp["main"])
        torch.cuda.synchronize()

    fed_steps(2)        # the first write to a fresh buffer pays for mapping its pages
    nh = 8
    t1 = time.perf_counter()
    fed_steps(nh)
    h2d_rate = nh * B / (time.perf_counter() - t1)

    # Rectification (the step before the path), outside the timed region: remap of one eye of every frame with the
    # broadcast maps, gray plane written alongside
    rect.remap_device(d_l, want_bgr=True, want_gray=True)
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    for _ in range(10):
        rect.remap_device(d_l, want_bgr=True, want_gray=True)
    torch.cuda.synchronize()
    remap_mpix = 10 * B * W * H / (time.perf_counter() - t1) / 1e6

    # Steger alone (Mpix/s over all pixels), outside the timed region
    steger = pipes[0]["steger"]
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    for _ in range(10):
        steger.extract_device(d_l, cap)
    torch.cuda.synchronize()
    steger_mpix = 10 * B * W * H / (time.perf_counter() - t1) / 1e6

    per_rank = parallel.gather_counts(B * args.steps / runs[len(runs) // 2], device=dev)
    if verified is not None:
        verified = all(v > 0.5 for v in parallel.gather_counts(1.0 if verified else 0.0, device=dev))   # every rank's check
    if rank == 0:
        sgbm_b, total_b = alg_bytes(W, H, D, use_wls, n_pts, want_xyz)
        calls = max(prof.pop("calls"), 1)
        kernel_ms = {k: v / calls for k, v in prof.items()}
        dom = max(kernel_ms, key=kernel_ms.get)
        if prof_r is not None:
            calls_r = max(prof_r.pop("calls"), 1)
            kernel_ms.update({"right/" + k: v / calls_r for k, v in prof_r.items()})
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
                key = "C2" if args.config == "C4" else args.config
                traffic = tj.get(key, {}).get(dom)
                if traffic is not None:   # recorded per launch at the batch of the profiling run
                    traffic = traffic * B / tj.get(key + "_batch", 64)
            except Exception:
                traffic = None
        out = {
            "metric": "stereo frame-pairs/sec @640x360 D=64" if args.config in ("C2", "C4") else f"stereo frame-pairs/sec {args.config}",
            "value": value, "unit": "frame-pairs/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": scaling,
            "vs_baseline": None, "dtype": "s16", "data": "synthetic",
            "config": {"workload": f"{args.config}: {W}x{H} rectified pairs, SGBM-{('5PATH', '8PATH', '3WAY')[args.sgbm_mode]} D={D} block={block}"
                                   + (f" + right matcher + WLS lambda={lam:g}" if use_wls else "")
                                   + " + depth" + (" + XYZ reprojection" if want_xyz else "") + " + Steger sigma=3.0 + reconstruct_from_depth"
                                   + (" + plane/refraction reconstruction" if want_xyz else ""),
                       "batch_per_gpu": B, "frames_per_step_all_ranks": total_per_step, "distinct_frames": True, "sharding": sharding,
                       "pipelines_in_flight": len(pipes)},
            "timing": {"repeats": len(runs), "reported": "median", "seconds_per_run": runs,
                       "value_min": total_per_step * args.steps / max(runs), "value_max": total_per_step * args.steps / min(runs)},
            "verified": verified,
            "per_rank_frame_pairs_per_s": per_rank, "rccl_ranks": world if (use_dist and backend == "nccl") else 0,
            "calibration_broadcast_ms": bcast_ms,
            "roofline": {"bound": "hbm", "kernel": dom, "achieved": achieved, "peak": 8000.0, "unit": "GB/s",
                         "frac": achieved / 8000.0, "traffic": traffic,
                         "timing": "HIP events around each kernel, second pass of the same steps with whole-batch launches on one stream"},
            "pipeline_roofline": {"alg_bytes_per_pair": total_b, "achieved_GBps": total_b * value / world / 1e9,
                                  "frac": total_b * value / world / 8e12},
            "kernel_ms_per_launch": kernel_ms,
            "stage_ms_per_step_serialised": stage_ms,
            "h2d_inclusive": {"value": h2d_rate * world, "unit": "frame-pairs/s",
                              "note": "both BGR frames of every pair copied from pinned host memory for every step, into a second input buffer while the previous step runs"},
            "steger_mpix_per_s": steger_mpix, "steger_points_per_frame": n_pts,
            "remap_mpix_per_s": remap_mpix,
        }
        if cpu is not None:
            out["cpu_baseline"] = cpu
        elif world > 1:
            out["cpu_baseline_note"] = "reported by the N = 1 run only (rank 0, before the GPU is touched)"
        if also is not None:
            out["also"] = also
        json_out.write(json.dumps(out) + "\n")
        json_out.flush()
    if use_dist:
        dist.destroy_process_group()
    if verified is False:
        print("bench.py: the batch's outputs differ from the single-frame handle's", file=sys.stderr)
        return 1
    return 0


if __name__ == "__main__":
    sys.exit(main())
