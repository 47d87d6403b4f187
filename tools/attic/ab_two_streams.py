"""Developer experiment: the S2 step on one stream vs split in two half-batches (4 images each) on two HIP streams -- do the
HBM-bound kernels of one half (ROIAlign, Winograd transforms) run under the MFMA-bound GEMMs of the other?"""
import sys, os, time, types
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, bench
args = bench.parse([])
dev = torch.device("cuda")
wl = bench.Workload(args, dev)
heads = wl.heads
heads.res5_overflow_check = os.environ.get('CHECK', '1') == '1'   # (the per-call read of the guard word is a host sync)
feats = wl.features["res4"]
boxes = [p.proposal_boxes for p in wl.proposals]
B = len(boxes)
halves = [(feats[:B // 2].contiguous(), boxes[:B // 2]), (feats[B // 2:].contiguous(), boxes[B // 2:])]
streams = [torch.cuda.Stream(), torch.cuda.Stream()]

def part(f, b):
    with torch.no_grad():
        x = heads._shared_roi_transform([f], b, pooled=True)
        return heads.box_predictor(x)

def one():
    return part(feats, boxes)

def two():
    cur = torch.cuda.current_stream()
    outs = []
    for st, (f, b) in zip(streams, halves):
        st.wait_stream(cur)
        with torch.cuda.stream(st):
            outs.append(part(f, b))
    for st in streams:
        cur.wait_stream(st)
    return outs

def timed(fn, n=10):
    fn(); fn(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3

a = one(); b = two()
print("equal", torch.equal(a[0], torch.cat([b[0][0], b[1][0]])))
for _ in range(2):
    print(f"one stream {timed(one):.2f} ms   two streams {timed(two):.2f} ms")
