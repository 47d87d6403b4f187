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
    torch.manual_seed(500 + rank)                    # the proposal sampler draws from the global RNG
    loss, n = tw.forward_backward()
    torch.cuda.synchronize()
    grads = {k: p.grad.detach().cpu() for k, p in tw.module.named_parameters() if p.grad is not None}
    torch.save({"loss": float(loss), "n_sampled": n, "grads": grads if rank == 0 else None}, os.path.join(out_dir, f"rank{rank}.pt"))
    ones = torch.ones(1, device=device if backend == "nccl" else "cpu")
    dist.all_reduce(ones)
    assert int(ones.item()) == world
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
