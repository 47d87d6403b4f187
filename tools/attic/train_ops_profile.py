"""Developer aid: which Python lines of one LSM / STT training step issue the small torch ops (fills, adds, copies, cats)?  A TorchDispatchMode over one
step of bench.py's TrainWorkload counts every aten call by (op, innermost frame inside this repository); ops issued by autograd's backward nodes have no
Python frame and are listed under "(autograd engine)".
usage: python3 tools/attic/train_ops_profile.py [lsm|stt]"""
import collections, os, sys, traceback
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from torch.utils._python_dispatch import TorchDispatchMode
import bench

cfg = sys.argv[1] if len(sys.argv) > 1 else "lsm"
args = bench.parse([])
dev = torch.device("cuda:0")
torch.cuda.set_device(dev)
tw = bench.TrainWorkload(args, dev, "hip", 1, config=cfg)
for _ in range(6): tw.step()
torch.cuda.synchronize()
SKIP = ("aten.view", "aten.detach", "aten.t.", "aten._unsafe_view", "aten.slice", "aten.select", "aten.expand", "aten.alias", "aten.as_strided",
        "aten.unsqueeze", "aten.squeeze", "aten.permute", "aten.transpose", "aten.reshape", "aten.empty", "aten.split", "aten.unbind", "aten.narrow",
        "aten.is_", "aten.sym_", "aten.stride", "aten.size", "aten.lift_fresh", "aten.diagonal.", "aten._local_scalar_dense", "aten.new_empty",
        "aten.empty_like", "aten.is_pinned", "aten.record_stream", "aten.unfold")
count = collections.Counter()


class Mode(TorchDispatchMode):
    def __torch_dispatch__(self, func, types, a=(), kw=None):
        name = str(func)
        if not name.startswith(SKIP):
            where = "(autograd engine)"
            for fr in reversed(traceback.extract_stack()[:-1]):
                fn = fr.filename
                if ("locov_amd" in fn or fn.endswith("bench.py")) and "tools" not in fn:
                    where = f"{fn[fn.find('locov_amd'):] if 'locov_amd' in fn else 'bench.py'}:{fr.lineno} {fr.name}"
                    break
            count[(name, where)] += 1
        return func(*a, **(kw or {}))


with Mode():
    tw.step()
torch.cuda.synchronize()
print(f"# {cfg}: aten calls of one training step that may launch (views and metadata calls dropped): {sum(count.values())}")
bywhere = collections.Counter()
for (n, w), c in count.items(): bywhere[w] += c
for w, c in bywhere.most_common(60):
    ops = ", ".join(f"{n.replace('aten.', '')} x{k}" for (n, ww), k in sorted(count.items(), key=lambda kv: -kv[1]) if ww == w)
    print(f"{c:4d}  {w:75s} {ops[:150]}")
