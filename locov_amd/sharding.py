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


class GradientExchangeTrace:
    """Where DistributedDataParallel's buckets become ready on the timeline of the path's backward (the one
    exchange step of a training step is the all-reduce of the gradients of the parameters the path touches, overlapped with the
    backward -- ovr/engine/trainer.py:61-66).

    Res5's backward is two autograd nodes per bottleneck (res5_train.Res5BlockFn: "tail" = conv2 + conv3, "head" = conv1 + shortcut),
    so a half-block's weight gradients reach DDP's hooks when that half's kernels are enqueued.  This registers a DDP communication hook that records a HIP event on the launch stream
    at the moment a bucket is handed to the all-reduce (the point its collective waits for on the communication stream) and
    then runs the stock all-reduce, plus marks at the begin / middle / end of every block's backward; stop() places every
    bucket on [0, 1] of the Res5 backward -- in device time (`ready_at`) and in kernel launches of this library issued so far
    (`ready_at_launch`: the share of the Res5 backward's kernels that were NOT yet enqueued is 1 - that) -- and lists the host-side
    order of marks and buckets.

        trace = GradientExchangeTrace(ddp_module, module.named_parameters(), device)
        trace.start(); loss.backward(); rep = trace.stop()
    """

    def __init__(self, ddp, named_parameters, device):
        from torch.distributed.algorithms.ddp_comm_hooks import default_hooks
        self.device = torch.device(device)
        self.names = {}
        for n, p in named_parameters:
            self.names.setdefault(id(p), n)
        self.seq: List[Tuple[str, Any, Any]] = []
        self.on = False
        self._allreduce = default_hooks.allreduce_hook
        ddp.register_comm_hook(None, self._hook)

    def _event(self):
        """(HIP event on the launch stream, the library's launch counter) at this point of the host's enqueueing."""
        from . import _lib
        ev = torch.cuda.Event(enable_timing=True)
        ev.record(torch.cuda.current_stream(self.device))
        return ev, int(_lib.load().locov_launch_count())

    def _hook(self, state, bucket):
        if self.on:
            buf = bucket.buffer()
            info = {"index": int(bucket.index()), "bytes": int(buf.numel() * buf.element_size()),
                    "params": [self.names.get(id(p), "?") for p in bucket.parameters()]}
            self.seq.append(("bucket", info, self._event()))
        return self._allreduce(None, bucket)

    def _mark(self, kind: str, bi: int) -> None:
        self.seq.append((kind, int(bi), self._event()))

    def start(self) -> None:
        from . import res5_train
        self.seq = []
        self.on = True
        self.t0 = self._event()               # (every time is taken from here: elapsed_time wants its events in order)
        res5_train._BACKWARD_MARK = self._mark

    def stop(self) -> dict:
        from . import res5_train
        res5_train._BACKWARD_MARK = None
        self.on = False
        torch.cuda.synchronize(self.device)
        seq = self.seq
        head = next((e for k, _, e in seq if k == "backward_begin"), None)
        ends = [e for k, b, e in seq if k == "block_end" and b == 0]
        out = {"host_order": [k if k == "backward_begin" else (f"{k}:{b}" if k != "bucket" else f"bucket:{b['index']}") for k, b, _ in seq]}
        if head is None or not ends:
            out["error"] = "no Res5 backward inside the traced region"
            return out
        at = lambda e: self.t0[0].elapsed_time(e[0])
        t_head = at(head)
        total = at(ends[-1]) - t_head
        # the same points counted in the library's kernel launches (host side, exact; the device times above are this process's
        # view of a GPU it may share with other ranks)
        l_head, l_total = head[1], max(ends[-1][1] - head[1], 1)
        out["res5_backward_ms"] = total
        out["res5_backward_launches"] = l_total
        out["blocks_end_at"] = {f"block{b}": (at(e) - t_head) / total for k, b, e in seq if k == "block_end"}
        out["blocks_end_at_launch"] = {f"block{b}": (e[1] - l_head) / l_total for k, b, e in seq if k == "block_end"}
        out["block_tails_end_at"] = {f"block{b}": (at(e) - t_head) / total for k, b, e in seq if k == "block_mid"}
        buckets = []
        for k, info, e in seq:
            if k != "bucket":
                continue
            t = (at(e) - t_head) / total
            res5 = sorted({n.split("res5.")[1].split(".")[0] for n in info["params"] if "res5." in n})
            buckets.append({"index": info["index"], "MB": round(info["bytes"] / 1e6, 2), "ready_at": t,
                            "ready_at_launch": (e[1] - l_head) / l_total, "res5_blocks": res5, "params": info["params"]})
        out["buckets"] = buckets
        return out
