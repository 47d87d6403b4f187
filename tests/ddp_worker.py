"""Rank program of tests/test_gpu_multirank.py: one LSM training step of the path under DistributedDataParallel.

Launched as `python -m torch.distributed.run --nproc-per-node 2 tests/ddp_worker.py OUT_DIR`; both ranks share cuda:0
(the test box has one GPU), so the process group runs on gloo -- the module, the kernels and the DDP gradient averaging are
the ones `bench.py --mode train --gpus N` uses over RCCL.  Rank r trains on its own synthetic image shard (seed 100 + r) and
rank 0 writes the averaged gradients."""
import os
import sys

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

SHAPE = ["--train-images", "2", "--proposals", "96", "--train-samples", "64", "--classes", "80"]
ARGS = SHAPE + ["--share-gpu", "--dist-backend", "gloo", "--gpus", "2"]
RCCL_ARGS = SHAPE + ["--force-dist", "--dist-backend", "nccl", "--gpus", "1"]


def ignore_band_boxes(n_props: int, budget: int, seed: int):
    """(proposals [n_props, 4], ground truth [3, 4]) of one 800 x 1333 image on which a sampler with an IGNORE band (Matcher
    thresholds [0.3, 0.7], labels [0, -1, 1]) cannot fill `budget`: most proposals are a GT box shifted by a third of its width
    (IoU 0.5: ignored), budget // 3 are 16-pixel boxes away from every GT (background).  With the GT appended the image has
    budget // 3 + 3 candidates < budget, although it has n_props >= budget proposals: the speculated sample MISSES."""
    import numpy as np
    rng = np.random.default_rng(seed)
    gt = np.array([[100, 100, 400, 400], [500, 200, 900, 600], [920, 300, 1220, 700]], dtype=np.float32)
    n_bg = budget // 3
    assert n_props >= budget > n_bg + 3
    band = gt[rng.integers(0, 3, n_props - n_bg)].copy()
    w = band[:, 2] - band[:, 0]
    shift = (w / 3.0 + rng.uniform(-2, 2, w.shape)).astype(np.float32)
    band[:, 0] += shift
    band[:, 2] += shift
    xy = np.stack([rng.uniform(0, 80, n_bg), rng.uniform(720, 780, n_bg)], axis=1).astype(np.float32)       # bottom-left corner strip
    bg = np.concatenate([xy, xy + 16.0], axis=1)
    return np.concatenate([band, bg]).astype(np.float32), gt


def ignore_band_batch(tw, seed: int):
    """bench.TrainWorkload batch (same maps / captions as set_data(seed)) whose images all miss the speculation."""
    from locov_amd.structures import Boxes, Instances
    tw.set_data(seed)
    budget = tw.heads.batch_size_per_image
    props, targets = [], []
    gen = torch.Generator().manual_seed(seed)
    for i in range(tw.n_images):
        b, gt = ignore_band_boxes(tw.args.proposals, budget, seed * 10 + i)
        p = Instances((800, 1333))
        p.proposal_boxes = Boxes(torch.from_numpy(b).to(tw.device))
        p.objectness_logits = torch.zeros(len(b), device=tw.device)
        t = Instances((800, 1333))
        t.gt_boxes = Boxes(torch.from_numpy(gt).to(tw.device))
        t.gt_classes = torch.randint(0, tw.n_classes, (len(gt),), generator=gen).to(tw.device)
        props.append(p)
        targets.append(t)
    tw.proposals, tw.targets = props, targets


def set_case(tw, case: str, shard: int):
    """LOCOV_DDP_WORKER_CASE "legs": every rank's sampler has an ignore band; shard 0's batch misses the speculation, shard 1's res4
    map leaves the split arithmetic's range (its forward is repeated on the f32 MFMA) -- two ranks on two different legs of the
    training forward's retry machine inside ONE DistributedDataParallel step."""
    if case != "legs":
        return
    from locov_amd.roi_heads.roi_emb_heads import Matcher
    tw.heads.proposal_matcher = Matcher([0.3, 0.7], [0, -1, 1], allow_low_quality_matches=False)
    if shard == 0:
        ignore_band_batch(tw, 100)
    else:
        tw.set_data(101)
        tw.features = tw.features * 3.0e4


def main():
    out_dir = sys.argv[1]
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    backend = os.environ.get("LOCOV_DDP_WORKER_BACKEND", "gloo")      # "nccl": the one-rank RCCL rehearsal (one process per GPU only)
    torch.cuda.set_device(0)
    device = torch.device("cuda", 0)
    if backend == "nccl":
        assert world == 1, "two RCCL ranks cannot share a GPU"
        dist.init_process_group("nccl", device_id=device)
    else:
        dist.init_process_group("gloo")
    args = bench.parse(RCCL_ARGS if backend == "nccl" else ARGS)
    tw = bench.TrainWorkload(args, device, "hip", world, data_seed=100 + rank)
    case = os.environ.get("LOCOV_DDP_WORKER_CASE", "")
    set_case(tw, case, rank)
    torch.manual_seed(500 + rank)                    # the proposal sampler draws from the global RNG
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter("ignore", RuntimeWarning)          # ("legs": the repeated forward says so)
        loss, n = tw.forward_backward()
    torch.cuda.synchronize()
    grads = {k: p.grad.detach().cpu() for k, p in tw.module.named_parameters() if p.grad is not None}
    rec = {"loss": float(loss), "n_sampled": n, "grads": grads if rank == 0 else None, "stats": dict(tw.heads.stats)}
    if case:
        # this rank's OWN gradients of the same step without DistributedDataParallel (a second, identically seeded workload): what
        # the all-reduce should have averaged
        tw2 = bench.TrainWorkload(args, device, "hip", 1, data_seed=100 + rank, ddp=False)
        set_case(tw2, case, rank)
        torch.manual_seed(500 + rank)
        with warnings.catch_warnings():
            warnings.simplefilter("ignore", RuntimeWarning)
            loss2, n2 = tw2.forward_backward()
        torch.cuda.synchronize()
        rec["local"] = {"loss": float(loss2), "n_sampled": n2,
                        "grads": {k: p.grad.detach().cpu() for k, p in tw2.module.named_parameters() if p.grad is not None}}
        del tw2
    if os.environ.get("LOCOV_DDP_WORKER_TRACE"):
        # (behind the stock step whose gradients are compared: a communication hook that records an event per ready bucket and
        # runs the stock all-reduce; the last of three traced steps -- DDP has rebuilt its buckets in arrival order by then)
        # (on a batch of the LSM step's order of size -- 2 x 200 sampled proposals: the few thousand rows of the gradient test's batch
        # spend their backward on weight-sized work, which block 0's projection shortcut has most of)
        tw.heads.batch_size_per_image = 200
        tw.args.proposals = 600
        tw.set_data(300 + rank)
        rec["exchange"] = tw.trace_exchange(3)
        rec["bucket_cap_mb"] = tw.bucket_cap_mb
    torch.save(rec, os.path.join(out_dir, f"rank{rank}.pt"))
    ones = torch.ones(1, device=device if backend == "nccl" else "cpu")
    dist.all_reduce(ones)
    assert int(ones.item()) == world
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
