"""Multi-GPU plumbing: one process per GPU, frames sharded across ranks.

The path has no per-frame exchange: frames are independent
(reference main.py:164-189 only appends each frame's points to a list), so the
only collective is a start-up broadcast of the calibration (Q, rectification
maps, matcher parameters) from rank 0, over RCCL when the process group uses
the "nccl" backend.
"""
from __future__ import annotations

import io
from typing import Dict, List, Optional

import numpy as np


def shard_indices(n_items: int, rank: int, world: int) -> List[int]:
    """Round-robin shard: rank r owns items r, r+world, r+2*world, ..."""
    if world < 1 or not (0 <= rank < world):
        raise ValueError(f"bad rank/world {rank}/{world}")
    return list(range(rank, n_items, world))


def _is_dist() -> bool:
    try:
        import torch.distributed as dist
    except Exception:
        return False
    return dist.is_available() and dist.is_initialized()


def broadcast_calibration(calib: Optional[Dict[str, np.ndarray]], src: int = 0, device=None) -> Dict[str, np.ndarray]:
    """Broadcasts a dict of NumPy arrays (and scalars) from rank ``src`` to every rank.

    Two collectives: the byte length, then the npz-serialised payload as a uint8
    tensor (on ``device`` for the nccl backend, on the CPU for gloo).  Without an
    initialised process group the input is returned unchanged."""
    if not _is_dist():
        if calib is None:
            raise ValueError("no process group and no calibration given")
        return {k: np.asarray(v) for k, v in calib.items()}
    import torch
    import torch.distributed as dist

    rank = dist.get_rank()
    backend = dist.get_backend()
    dev = device if (backend == "nccl" and device is not None) else torch.device("cpu")
    if rank == src:
        buf = io.BytesIO()
        np.savez(buf, **{k: np.asarray(v) for k, v in calib.items()})
        payload = np.frombuffer(buf.getvalue(), dtype=np.uint8).copy()
        n = torch.tensor([payload.size], dtype=torch.int64, device=dev)
    else:
        payload = None
        n = torch.zeros(1, dtype=torch.int64, device=dev)
    dist.broadcast(n, src=src)
    if rank == src:
        t = torch.from_numpy(payload).to(dev)
    else:
        t = torch.empty(int(n.item()), dtype=torch.uint8, device=dev)
    dist.broadcast(t, src=src)
    data = np.load(io.BytesIO(t.cpu().numpy().tobytes()), allow_pickle=False)
    return {k: data[k] for k in data.files}


def gather_counts(value: float, device=None) -> List[float]:
    """all_gather of one float per rank (frame counts / elapsed times for the throughput report)."""
    if not _is_dist():
        return [float(value)]
    import torch
    import torch.distributed as dist

    backend = dist.get_backend()
    dev = device if (backend == "nccl" and device is not None) else torch.device("cpu")
    t = torch.tensor([value], dtype=torch.float64, device=dev)
    out = [torch.zeros_like(t) for _ in range(dist.get_world_size())]
    dist.all_gather(out, t)
    return [float(o.item()) for o in out]
