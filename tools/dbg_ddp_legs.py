"""Developer aid: one rank of tests/ddp_worker.py's "legs" case under a ONE-rank gloo DistributedDataParallel against the plain module
(which gradients does DDP lose when the forward was repeated?).  usage: python3 tools/dbg_ddp_legs.py [shard]"""
import datetime, os, sys, warnings
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.distributed as dist
import bench
from tests import ddp_worker

shard = int(sys.argv[1]) if len(sys.argv) > 1 else 1
dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{bench.free_port()}", rank=0, world_size=1, timeout=datetime.timedelta(seconds=60))
args = bench.parse(ddp_worker.ARGS[:8] + ["--share-gpu", "--dist-backend", "gloo"])
out = {}
for ddp in (False, True):
    tw = bench.TrainWorkload(args, dev, "hip", 1, data_seed=100 + shard, ddp=ddp)
    ddp_worker.set_case(tw, "legs", shard)
    torch.manual_seed(500 + shard)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore", RuntimeWarning)
        loss, n = tw.forward_backward()
    torch.cuda.synchronize()
    out[ddp] = {k: p.grad.detach().clone() for k, p in tw.module.named_parameters() if p.grad is not None}
    print("ddp" if ddp else "plain", float(loss), n, dict(tw.heads.stats))
for k in out[False]:
    a, b = out[True][k], out[False][k]
    print(f"{k:45s} plain max {float(b.abs().max()):12.4e}  ddp max {float(a.abs().max()):12.4e}  rel diff {float((a - b).abs().max() / b.abs().max().clamp_min(1e-30)):.3e}")
dist.destroy_process_group()
