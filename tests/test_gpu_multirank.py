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
    env = _env()
    env["LOCOV_DDP_WORKER_TRACE"] = "1"
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    ranks = [torch.load(os.path.join(tmp_path, f"rank{i}.pt")) for i in range(2)]
    _check_bucket_schedule(ranks[0]["exchange"])
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


def test_ddp_ranks_on_different_legs_of_the_retry_machine(tmp_path):
    """inside ONE DistributedDataParallel step rank 0's speculated sample misses (an ignore-band sampler that
    cannot fill its budget: the forward is repeated from the true counts) while rank 1's forward leaves the split arithmetic's
    range (RES5_TRAIN_GUARD "sync": repeated on the f32 MFMA).  Both repeats happen inside the forward, so both ranks enter the
    same collectives in the backward: the step completes and DDP's gradients equal the single-process average of the two shards."""
    if not torch.cuda.is_available():
        pytest.fail("GPU tests need a ROCm device")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
           "--master-port", str(_port()), os.path.join(ROOT, "tests", "ddp_worker.py"), str(tmp_path)]
    env = _env()
    env["LOCOV_DDP_WORKER_CASE"] = "legs"
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    ranks = [torch.load(os.path.join(tmp_path, f"rank{i}.pt")) for i in range(2)]
    s0, s1 = ranks[0]["stats"], ranks[1]["stats"]
    assert s0.get("speculation_misses") == 1 and not s0.get("fp32_repeats"), s0
    assert s1.get("fp32_repeats") == 1 and s1.get("guard_trips") == 1 and not s1.get("speculation_misses"), s1
    ddp_grads = ranks[0]["grads"]
    sys.path.insert(0, ROOT)
    import warnings
    import bench
    from tests import ddp_worker
    args = bench.parse(ddp_worker.ARGS)
    tw = bench.TrainWorkload(args, torch.device("cuda", 0), "hip", world=1, data_seed=100)
    for shard in range(2):
        ddp_worker.set_case(tw, "legs", shard)
        torch.manual_seed(500 + shard)
        with warnings.catch_warnings():
            warnings.simplefilter("ignore", RuntimeWarning)
            loss, n = tw.forward_backward(scale=0.5)
        assert n == ranks[shard]["n_sampled"]
        assert abs(float(loss) - ranks[shard]["loss"]) <= 1e-5 * max(1.0, abs(float(loss)))
    assert tw.heads.stats.get("speculation_misses") == 1 and tw.heads.stats.get("fp32_repeats") == 1
    single = {k: p.grad.detach().cpu() for k, p in tw.module.named_parameters() if p.grad is not None}
    assert set(single) == set(ddp_grads)
    rel = lambda a, b: float((a - b).abs().max()) / float(b.abs().max().clamp_min(1e-30))
    # the ranks' own (un-reduced) gradients of the same step, averaged: what DDP's all-reduce has to produce
    local = {k: 0.5 * (ranks[0]["local"]["grads"][k] + ranks[1]["local"]["grads"][k]) for k in ddp_grads}
    errs_ddp = {k: rel(ddp_grads[k], local[k]) for k in ddp_grads}
    errs_single = {k: rel(single[k], local[k]) for k in single}
    assert all(bool(torch.isfinite(v).all()) for v in ddp_grads.values())
    assert max(errs_ddp.values()) <= 1e-5, ("DDP vs the average of the ranks' own gradients", errs_ddp)
    assert max(errs_single.values()) <= 1e-5, ("one process running both shards vs the average of the ranks' own gradients", errs_single)


def _check_bucket_schedule(ex):
    """Res5's backward is one autograd node per bottleneck, so DistributedDataParallel sees block 2's weight
    gradients -- and can start their all-reduce -- while blocks 1 and 0 are still to run.  `ex`: GradientExchangeTrace.stop() of a
    traced step (a communication hook recorded a HIP event per ready bucket on the launch stream, then ran the stock all-reduce)."""
    assert "error" not in ex, ex
    order, buckets = ex["host_order"], ex["buckets"]
    assert [o for o in order if not o.startswith("bucket")] == ["backward_begin"] + [
        f"{k}:{b}" for b in (2, 1, 0) for k in ("block_begin", "block_mid", "block_end")], order
    with_res5 = [b for b in buckets if b["res5_blocks"]]
    assert len(with_res5) >= 3, buckets
    first = min(with_res5, key=lambda b: b["ready_at_launch"])
    assert first["res5_blocks"] == ["2"], first
    # ready while >= 60 % of the Res5 backward's kernels are still to be enqueued (counted in the library's launches: exact on the
    # host; the two ranks of this test share ONE GPU, whose time slices make device-time fractions of a single rank noisy) ...
    assert first["ready_at_launch"] <= 0.40, (first["ready_at_launch"], ex["blocks_end_at_launch"])
    assert ex["res5_backward_launches"] >= 30
    # ... and, on the host, before block 1's backward has been entered
    assert order.index(f"bucket:{first['index']}") < order.index("block_begin:1"), order
    # the bucket that closes with block 1's conv2 / conv3 gradients is handed over before block 0's backward is entered
    b1 = [b for b in with_res5 if any(".res5.1.conv2." in n for n in b["params"])]
    assert b1 and order.index(f"bucket:{b1[0]['index']}") < order.index("block_begin:0"), order
    assert b1[0]["ready_at_launch"] <= ex["blocks_end_at_launch"]["block1"] + 1e-9
    # two nodes per bottleneck: block 0's conv2 / conv3 gradients (13.6 MB) are handed over in the MIDDLE of block 0, not behind
    # the stage's last kernel -- what is left for the end is its conv1 / shortcut (10.5 MB)
    b0 = [b for b in with_res5 if any(".res5.0.conv2." in n for n in b["params"])]
    assert b0 and order.index(f"bucket:{b0[0]['index']}") < order.index("block_end:0"), order
    every = sorted(n for b in buckets for n in b["params"])
    assert len(every) == len(set(every)) and sum("res5." in n for n in every) == 10, every


def test_bench_starts_its_own_ranks():
    """`python bench.py --gpus 2` with no launcher (it used to run ONE rank and report n_gpus 1)."""
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--share-gpu", "--dist-backend", "gloo", "--steps", "2",
           "--warmup", "1", "--images", "1", "--proposals", "200", "--classes", "80", "--no-cpu-baseline", "--skip-s1",
           "--skip-f32-reference", "--skip-variants", "--skip-eval", "--multiscale-batches", "2"]
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
           "--unfrozen-steps", "0", "--no-cpu-baseline", "--skip-s1", "--skip-f32-reference", "--skip-variants", "--skip-eval", "--multiscale-batches", "2"]
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
    assert "8 ranks" in train["gradient_exchange"]["how"]
    sched = train["gradient_exchange"]["schedule"]["lsm"]
    assert "error" not in sched and any(b["res5_blocks"] == ["2"] for b in sched["buckets"]), sched
    for cfg in ("lsm", "stt"):
        assert len(train[cfg]["per_rank_ms_per_step"]) == 8 and train[cfg]["ms_per_step"] > 0


def test_rccl_one_rank_rehearsal():
    """RCCL itself on this code (the two-rank tests above share one GPU and therefore run gloo): `bench.py --force-dist` creates the
    RCCL communicator with ONE rank, wraps both training steps in DistributedDataParallel -- its bucketed all-reduce hooks fire
    inside the joint Res5 backward with its side stream, on the RCCL stream -- and takes the timing collectives on device tensors."""
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--force-dist", "--dist-backend", "nccl", "--steps", "2",
           "--warmup", "1", "--images", "1", "--proposals", "200", "--classes", "80", "--train-images", "1", "--train-samples", "32",
           "--unfrozen-steps", "0", "--no-cpu-baseline", "--skip-s1", "--skip-f32-reference", "--skip-variants", "--skip-eval", "--multiscale-batches", "2"]
    env = _env()
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    out = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert out["n_gpus"] == 1 and out["rccl_ranks"] == 1 and out["process_group"] == "nccl x1" and out["value"] > 0
    assert "DistributedDataParallel over 1 ranks (nccl)" in out["train"]["gradient_exchange"]["how"]
    sched = out["train"]["gradient_exchange"]["schedule"]["lsm"]              # RCCL's own all-reduce behind the traced hook
    assert "error" not in sched and any(b["res5_blocks"] == ["2"] for b in sched["buckets"]), sched
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
