"""The N > 1 product path on the GPU box (train_ovnet.py:100-107 launches one process per GPU,
ovr/engine/trainer.py:61-66 wraps the model in DistributedDataParallel): two ranks run one LSM training step of the ROI
heads + GroundingHead under DDP on their own image shards, and the gradients DDP leaves on every rank must equal the
average of the two shards' gradients computed by ONE process.  Also: `bench.py --gpus 2` started WITHOUT a launcher starts
its two ranks itself and reports n_gpus = 2."""
import json
import os
import socket
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _env():
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env["PYTHONPATH"] = ROOT + os.pathsep + env.get("PYTHONPATH", "")
    return env


def test_ddp_training_step_matches_single_process(tmp_path):
    if not torch.cuda.is_available():
        pytest.fail("GPU tests need a ROCm device")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
           "--master-port", str(_port()), os.path.join(ROOT, "tests", "ddp_worker.py"), str(tmp_path)]
    r = subprocess.run(cmd, env=_env(), capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    ranks = [torch.load(os.path.join(tmp_path, f"rank{i}.pt")) for i in range(2)]
    ddp_grads = ranks[0]["grads"]
    # (emb_pred IS the grounding head's v2l_projection -- distill_prop_mmss_gcnn.py:117-125 -- and is listed once, under its first name)
    assert any(k.startswith("heads.res5.") for k in ddp_grads) and "heads.box_predictor.emb_pred.weight" in ddp_grads

    sys.path.insert(0, ROOT)
    import bench
    from tests import ddp_worker
    args = bench.parse(ddp_worker.ARGS)
    device = torch.device("cuda", 0)
    tw = bench.TrainWorkload(args, device, "hip", world=1, data_seed=100)
    losses = []
    for shard in range(2):                               # the two ranks' shards, one after the other, gradients accumulated
        tw.set_data(100 + shard)
        torch.manual_seed(500 + shard)
        loss, n = tw.forward_backward(scale=0.5)         # DDP averages over the ranks
        losses.append(float(loss))
        assert n == ranks[shard]["n_sampled"]
    for shard in range(2):
        assert abs(losses[shard] - ranks[shard]["loss"]) <= 1e-5 * max(1.0, abs(losses[shard]))
    single = {k: p.grad.detach().cpu() for k, p in tw.module.named_parameters() if p.grad is not None}
    assert set(single) == set(ddp_grads)
    for k in single:
        scale = float(single[k].abs().max().clamp_min(1e-30))
        err = float((single[k] - ddp_grads[k]).abs().max()) / scale
        assert err <= 1e-5, (k, err)


def test_bench_starts_its_own_ranks():
    """`python bench.py --gpus 2` with no launcher (it used to run ONE rank and report n_gpus 1)."""
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--share-gpu", "--dist-backend", "gloo", "--steps", "2",
           "--warmup", "1", "--images", "1", "--proposals", "200", "--classes", "80", "--no-cpu-baseline", "--skip-s1",
           "--skip-f32-reference", "--skip-variants"]
    env = _env()
    env.pop("WORLD_SIZE", None)
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    line = [ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1]
    out = json.loads(line)
    assert out["n_gpus"] == 2 and out["rccl_ranks"] == 2 and out["value"] > 0
    assert out["config"]["parallelism"].startswith("image-sharded x2")


def test_bench_rehearses_eight_ranks():
    """The driver's SCALE run is the first time eight ranks exist (train_ovnet.py:100-107 launches one process per GPU): rehearse
    the flow on ONE GPU -- `bench.py --gpus 8` starts eight ranks itself, every rank times the inference step and one LSM / STT
    training step under DistributedDataParallel on its own images, and the line names all eight."""
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--share-gpu", "--dist-backend", "gloo", "--steps", "2",
           "--warmup", "1", "--images", "1", "--proposals", "200", "--classes", "80", "--train-images", "1", "--train-samples", "32",
           "--unfrozen-steps", "0", "--no-cpu-baseline", "--skip-s1", "--skip-f32-reference", "--skip-variants"]
    env = _env()
    env.pop("WORLD_SIZE", None)
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=1500)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    line = [ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1]
    out = json.loads(line)
    assert out["n_gpus"] == 8 and out["rccl_ranks"] == 8
    assert out["value"] > 0 and out["value"] == out["value"] and out["value"] != float("inf")
    assert out["config"]["parallelism"].startswith("image-sharded x8")
    assert len(out["per_rank_ms_per_step"]) == 8 and all(0 < t < 1e5 for t in out["per_rank_ms_per_step"])
    assert abs(max(out["per_rank_ms_per_step"]) - out["ms_per_step"]) <= 1e-6 * out["ms_per_step"]
    train = out["train"]
    assert "8 ranks" in train["gradient_exchange"]
    for cfg in ("lsm", "stt"):
        assert len(train[cfg]["per_rank_ms_per_step"]) == 8 and train[cfg]["ms_per_step"] > 0


def test_rccl_one_rank_rehearsal():
    """RCCL itself on this code (the two-rank tests above share one GPU and therefore run gloo): `bench.py --force-dist` creates the
    RCCL communicator with ONE rank, wraps both training steps in DistributedDataParallel -- its bucketed all-reduce hooks fire
    inside the joint Res5 backward with its side stream, on the RCCL stream -- and takes the timing collectives on device tensors."""
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--force-dist", "--dist-backend", "nccl", "--steps", "2",
           "--warmup", "1", "--images", "1", "--proposals", "200", "--classes", "80", "--train-images", "1", "--train-samples", "32",
           "--unfrozen-steps", "0", "--no-cpu-baseline", "--skip-s1", "--skip-f32-reference", "--skip-variants"]
    env = _env()
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    out = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert out["n_gpus"] == 1 and out["rccl_ranks"] == 1 and out["process_group"] == "nccl x1" and out["value"] > 0
    assert "DistributedDataParallel over 1 ranks (nccl)" in out["train"]["gradient_exchange"]
    for cfg in ("lsm", "stt"):
        assert out["train"][cfg]["ms_per_step"] > 0


def test_rccl_one_rank_ddp_gradients_equal_the_plain_step(tmp_path):
    """The gradients DistributedDataParallel leaves behind an RCCL all-reduce over one rank are the plain step's."""
    worker = os.path.join(ROOT, "tests", "ddp_worker.py")
    env = _env()
    env.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_port()), RANK="0", WORLD_SIZE="1", LOCAL_RANK="0", LOCOV_DDP_WORKER_BACKEND="nccl")
    r = subprocess.run([sys.executable, worker, str(tmp_path)], env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    got = torch.load(os.path.join(tmp_path, "rank0.pt"))
    sys.path.insert(0, ROOT)
    import bench
    from tests import ddp_worker
    args = bench.parse(ddp_worker.ARGS)
    tw = bench.TrainWorkload(args, torch.device("cuda", 0), "hip", world=1, data_seed=100)
    torch.manual_seed(500)
    loss, n = tw.forward_backward()
    assert n == got["n_sampled"] and abs(float(loss) - got["loss"]) <= 1e-5 * max(1.0, abs(float(loss)))
    single = {k: p.grad.detach().cpu() for k, p in tw.module.named_parameters() if p.grad is not None}
    assert set(single) == set(got["grads"])
    for k in single:
        scale = float(single[k].abs().max().clamp_min(1e-30))
        assert float((single[k] - got["grads"][k]).abs().max()) / scale <= 1e-5, k
