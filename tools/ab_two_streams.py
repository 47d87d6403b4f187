"""Developer experiment: the S2 step on one stream vs split in two half-batches on two HIP streams."""
import sys, os, types
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
from locov_amd import ops
args = types.SimpleNamespace(images=8, proposals=1000, classes=1203, dim=768, sim_dtype="fp32", res5="hip", conv3x3="winograd", block0="map", res5_dtype="f16x2")
dev = torch.device("cuda")
wl = bench.Workload(args, dev)
nh = ops.nchw_to_nhwc(wl.feat)
halves = [wl.rois[:4000].contiguous(), wl.rois[4000:].contiguous()]
streams = [torch.cuda.Stream(), torch.cuda.Stream()]

def part(rois):
    y = wl.res5.forward_from_map(nh, rois, 14, 1.0 / 16, 0, True, winograd=True, split=True)
    return wl.head(y.view(7, 7, rois.shape[0], 2048), channels_last=2)

def one():
    return part(wl.rois)

def two():
    cur = torch.cuda.current_stream()
    outs = []
    for st, r in zip(streams, halves):
        st.wait_stream(cur)
        with torch.cuda.stream(st):
            outs.append(part(r))
    for st in streams:
        cur.wait_stream(st)
    return outs

def t(f, n=10):
    for _ in range(5): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n

with torch.no_grad():
    a = one(); b = two()
    print("same logits:", torch.equal(a[3], torch.cat([b[0][3], b[1][3]])))
    for rnd in range(3):
        print("one stream %.2f ms   two streams %.2f ms" % (t(one), t(two)))
