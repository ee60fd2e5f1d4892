"""Frame sharding for multi-GPU inference: independent replicas, one process per GPU, no data-path
collective.  torch.distributed (RCCL on ROCm, gloo in the CPU tests) is used only to
line the ranks up and to take the MAX over ranks of the elapsed time."""
from __future__ import annotations

import datetime
import os
from typing import List, Optional, Tuple

import torch


def rank_world() -> Tuple[int, int, int]:
    return (int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")),
            int(os.environ.get("WORLD_SIZE", "1")))


def init(backend: str, device: Optional[torch.device] = None):
    """Returns the torch.distributed module when WORLD_SIZE > 1, else None."""
    _, _, world = rank_world()
    if world <= 1:
        return None
    import torch.distributed as dist
    if not dist.is_initialized():
        kw = {"device_id": device} if (backend == "nccl" and device is not None) else {}
        # a collective that never completes (a rank died, mismatched calls) aborts the job after this long instead of
        # holding N GPUs for torch's default 10 minutes
        kw["timeout"] = datetime.timedelta(seconds=int(os.environ.get("BEVF_DIST_TIMEOUT_S", "300")))
        dist.init_process_group(backend, **kw)
    return dist


def shard_frames(n_frames: int, rank: int, world: int) -> List[int]:
    """Static round-robin of frame indices over ranks (frames are independent)."""
    return list(range(rank, n_frames, world))


def frame_seed(base: int, config: int, frame: int) -> int:
    """seed = base + 1000*config + frame index (identical on every machine)."""
    return base + 1000 * config + frame


def barrier(dist) -> None:
    if dist is not None:
        dist.barrier()


def max_over_ranks(value: float, dist, device: torch.device) -> float:
    if dist is None:
        return value
    if dist.get_backend() == "gloo":
        device = torch.device("cpu")
    t = torch.tensor([value], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def gather_over_ranks(value: float, dist, device: torch.device) -> List[float]:
    """Every rank's value, in rank order, on every rank (bench.py: per-rank step times, so a straggler shows)."""
    if dist is None:
        return [value]
    if dist.get_backend() == "gloo":
        device = torch.device("cpu")
    mine = torch.tensor([value], dtype=torch.float64, device=device)
    out = [torch.zeros_like(mine) for _ in range(dist.get_world_size())]
    dist.all_gather(out, mine)
    return [float(t.item()) for t in out]


def aggregate_fps(frames_per_rank: int, world: int, elapsed_max_s: float) -> float:
    """Whole-job throughput: all ranks' frames over the slowest rank's time."""
    return world * frames_per_rank / elapsed_max_s


# ---- data-parallel training: one exchange step per iteration = all-reduce of the flat gradient ----------

def allreduce_gradients(parameters, dist, bucket_bytes: int = 256 << 20) -> int:
    """Average `p.grad` over the ranks: gradients are packed into flat buckets (a few large collectives, not 243
    small ones), summed with one all-reduce each (RCCL over xGMI on the GPU node, gloo in the CPU tests) and
    divided by the world size; BatchNorm buffers stay local (the reference has no SyncBN).  Returns the number of
    collectives issued.  With dist None (single process) this is a no-op."""
    if dist is None:
        return 0
    world = dist.get_world_size()
    params = [p for p in parameters if p.grad is not None]
    n_coll, bucket, size = 0, [], 0

    def flush():
        nonlocal n_coll, bucket, size
        if not bucket:
            return
        flat = torch.cat([p.grad.detach().reshape(-1) for p in bucket])
        dist.all_reduce(flat, op=dist.ReduceOp.SUM)
        flat.div_(world)
        o = 0
        for p in bucket:
            n = p.grad.numel()
            p.grad.copy_(flat[o:o + n].view_as(p.grad))
            o += n
        n_coll += 1
        bucket, size = [], 0

    for p in params:
        nbytes = p.grad.numel() * p.grad.element_size()
        if bucket and size + nbytes > bucket_bytes:
            flush()
        bucket.append(p)
        size += nbytes
    flush()
    return n_coll


class GradReducer:
    """Averages gradients over the ranks WHILE the backward is still running (training.set_grad_reducer).

    The detector's hand-written backward calls `submit` each time a group of gradients is final -- first the head,
    fusion, radar and LiDAR branches (dominated by one 164 MB dense layer), then each ResNet stage: the group is packed
    into one flat buffer and its all-reduce starts asynchronously on RCCL's stream, so the exchange over xGMI hides
    under the camera trunk's backward (~half of the step).  `finish` waits for the collectives (stream-side for RCCL),
    divides by the world size and hands the averaged views back as the gradients.  A few large collectives, as the
    point-to-point xGMI links prefer; BatchNorm buffers stay local (the reference has no SyncBN)."""

    def __init__(self, dist):
        self.dist = dist
        self.world = dist.get_world_size()
        self.pending = []
        self.collectives = 0

    def submit(self, sink, keys) -> None:
        grads = [sink.g[k] for k in keys]
        flat = torch.cat([g.reshape(-1) for g in grads])
        work = self.dist.all_reduce(flat, op=self.dist.ReduceOp.SUM, async_op=True)
        self.pending.append((work, flat, keys, [g.shape for g in grads]))
        self.collectives += 1

    def finish(self, sink) -> None:
        for work, flat, keys, shapes in self.pending:
            work.wait()
            flat.div_(self.world)
            o = 0
            for k, shp in zip(keys, shapes):
                n = 1
                for d in shp:
                    n *= d
                sink.g[k] = flat[o:o + n].view(shp)
                o += n
        self.pending = []
