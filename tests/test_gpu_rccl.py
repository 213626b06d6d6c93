"""The collectives of the multi-GPU path on the real backend: a one-rank RCCL process group on the GPU that is there.
(Two ranks cannot share a device under RCCL; the world-size-2 behaviour is covered by the gloo test on the CPU.)"""
import socket

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def test_calibration_broadcast_and_timing_collectives_over_rccl():
    import torch
    import torch.distributed as dist
    from laser_3d_reconstruction_amd import parallel
    from laser_3d_reconstruction_amd.camera import rectify

    assert not dist.is_initialized()
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{_free_port()}", rank=0, world_size=1, device_id=dev)
    try:
        assert dist.get_backend() == "nccl"
        W, H = 640, 360
        K = np.array([[418.0, 0, 319.5], [0, 418.0, 179.5], [0, 0, 1]])
        D = np.array([-0.1, 0.02, 0.0005, -0.0003, 0.0])
        rl = rectify.Rectifier.from_calibration(K, D, np.eye(3), np.hstack([K, np.zeros((3, 1))]), (W, H), device=0)
        calib = dict(K1=K, D1=D, Q=np.arange(16.0).reshape(4, 4), map_left_x=rl.maps[0], map_left_y=rl.maps[1], nd=np.int32(64))
        got = parallel.broadcast_calibration(calib, src=0, device=dev)          # payload travels as a device tensor
        assert sorted(got) == sorted(calib)
        for k in calib:
            assert np.array_equal(got[k], np.asarray(calib[k])) and got[k].dtype == np.asarray(calib[k]).dtype
        rect = rectify.Rectifier.from_maps(got["map_left_x"], got["map_left_y"], device=0)
        img = np.random.default_rng(0).integers(0, 256, (H, W, 3), dtype=np.uint8)
        assert np.array_equal(rect.remap(img), rl.remap(img))
        # the bench's fence and reductions
        torch.cuda.synchronize()
        dist.barrier()
        t = torch.tensor([1.25], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        assert float(t.item()) == 1.25
        assert parallel.gather_counts(256.0, device=dev) == [256.0]
        assert parallel.shard_indices(10, 0, 1) == list(range(10))
    finally:
        dist.destroy_process_group()
