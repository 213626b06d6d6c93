"""The N > 1 path of bench.py on the one GPU that is there: two ranks started by bench.py's own launcher
(`python -m torch.distributed.run --nproc-per-node 2 ...`), sharing the card, with the gloo backend for the
collectives (two RCCL ranks cannot share a device; the RCCL calls themselves are covered with one rank in
test_gpu_rccl.py).  Each run is a fresh child process of the test process, started with subprocess.run.

Checked: one JSON line on stdout, n_gpus 2, the sharding string, both ranks' rates, the calibration broadcast, and
that every rank's outputs verified against the single-frame handle.
"""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench(*args, timeout=300):
    env = dict(os.environ, L3D_BENCH_BACKEND="gloo", MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *args], cwd=ROOT, env=env, stdout=subprocess.PIPE,
                       stderr=subprocess.PIPE, text=True, timeout=timeout)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, r.stdout[-2000:]          # ONE JSON line, nothing else on stdout
    return json.loads(lines[0])


def test_two_ranks_share_frames_round_robin():
    d = _bench("--gpus", "2", "--config", "C2", "--batch", "8", "--steps", "2", "--warmup", "1", "--repeats", "1", "--no-cpu-baseline")
    assert d["n_gpus"] == 2 and d["scaling"] == "weak"
    assert d["config"]["sharding"] == "frames x2"
    assert d["config"]["batch_per_gpu"] == 8 and d["config"]["frames_per_step_all_ranks"] == 16
    assert d["verified"] is True
    assert len(d["per_rank_frame_pairs_per_s"]) == 2 and all(v > 0 for v in d["per_rank_frame_pairs_per_s"])
    assert d["value"] > 0 and d["metric"].startswith("stereo frame-pairs/sec")
    assert d["calibration_broadcast_ms"]["payload_bytes_with_maps"] > 4 * 640 * 360 * 4
    assert "cpu_baseline" not in d and "also" not in d and "cpu_baseline_note" in d
    assert d["rccl_ranks"] == 0                       # gloo rehearsal: says so


def test_two_ranks_take_four_streams_each():
    d = _bench("--gpus", "2", "--config", "C4", "--batch", "2", "--steps", "2", "--warmup", "1", "--repeats", "1", "--no-cpu-baseline")
    assert d["n_gpus"] == 2
    assert d["config"]["sharding"].startswith("8 streams over 2 rank(s), stream s on rank s % 2")
    assert d["config"]["batch_per_gpu"] == 8 and d["config"]["frames_per_step_all_ranks"] == 16     # 4 streams x 2 frames a rank
    assert d["verified"] is True and len(d["per_rank_frame_pairs_per_s"]) == 2
