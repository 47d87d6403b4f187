"""Image sharding of the LSM ROI-head path over the GPUs of one node.

Every ROI depends only on its own image's feature map; the text bank and the FC / Res5 weights are
replicated read-only.  So the path shards BY IMAGE with no data-path collective: each rank (one
process per GPU, `torch.distributed` over RCCL) runs the head on its own images.  The reference
does the same through Detectron2's DDP + TrainingSampler / InferenceSampler
(ovr/engine/trainer.py:45,61-66); its only collectives are DDP's gradient all-reduce and the
end-of-eval result gather (evaluator.py:75,87), which are reproduced here as
`gather_per_image` (inference results) -- the gradient exchange stays with DDP.
"""
from __future__ import annotations

from typing import Any, List, Sequence, Tuple

import torch
import torch.distributed as dist


def shard_range(n_items: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous [begin, end) of the items rank `rank` owns; sizes differ by at most one and the
    shards tile [0, n_items) exactly (same split as Detectron2's InferenceSampler)."""
    if not (0 <= rank < world):
        raise ValueError(f"rank {rank} outside world of size {world}")
    base, extra = divmod(n_items, world)
    begin = rank * base + min(rank, extra)
    return begin, begin + base + (1 if rank < extra else 0)


def shard_list(items: Sequence[Any], rank: int, world: int) -> List[Any]:
    b, e = shard_range(len(items), rank, world)
    return list(items[b:e])


def world_info() -> Tuple[int, int]:
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(), dist.get_world_size()
    return 0, 1


def max_over_ranks(seconds: float, device=None) -> float:
    """Wall time of the slowest rank (the bench contract times the whole job by it)."""
    rank, world = world_info()
    if not (dist.is_available() and dist.is_initialized()):
        return float(seconds)
    t = torch.tensor([seconds], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def all_ranks(seconds: float, device=None) -> List[float]:
    """Every rank's value, in rank order (the bench line lists them next to the maximum so that a scaling run explains its
    own efficiency: one slow rank vs all ranks slower)."""
    rank, world = world_info()
    if not (dist.is_available() and dist.is_initialized()):      # (an initialised one-rank group still runs the collective: bench.py --force-dist)
        return [float(seconds)]
    t = torch.zeros(world, dtype=torch.float64, device=device)
    t[rank] = seconds
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return [float(v) for v in t.tolist()]


def gather_per_image(local_results: List[Any], n_images: int) -> List[Any]:
    """All ranks' per-image results in global image order (every rank gets the full list).
    The counterpart of the evaluator's comm.gather of predictions."""
    rank, world = world_info()
    if world == 1:
        assert len(local_results) == n_images
        return list(local_results)
    b, e = shard_range(n_images, rank, world)
    assert len(local_results) == e - b, f"rank {rank} produced {len(local_results)} results for {e - b} images"
    gathered: List[Any] = [None] * world
    dist.all_gather_object(gathered, local_results)
    out: List[Any] = []
    for part in gathered:
        out.extend(part)
    assert len(out) == n_images
    return out
