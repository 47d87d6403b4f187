"""Developer experiment: the S2 inference step captured in a HIP graph (torch.cuda.CUDAGraph) vs eager launches, at small batches
where ~25 launches per step are a visible share of the step."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, bench
for images in (1, 2, 8):
    args = bench.parse(["--images", str(images)])
    dev = torch.device("cuda")
    wl = bench.Workload(args, dev)
    heads = wl.heads
    heads.res5_overflow_check = False            # the graph version reads the guard word after the replay instead
    feats = wl.features["res4"]
    boxes = [p.proposal_boxes for p in wl.proposals]

    def step():
        with torch.no_grad():
            return heads.box_predictor(heads._shared_roi_transform([feats], boxes, pooled=True))

    def timed(fn, n=20):
        fn(); fn(); torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(n): fn()
        torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3

    eager = step()
    t_eager = timed(step)
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        for _ in range(2): step()
    torch.cuda.current_stream().wait_stream(s)
    with torch.cuda.graph(g):
        out = step()
    t_graph = timed(g.replay)
    g.replay(); torch.cuda.synchronize()
    print(f"images {images}: eager {t_eager:.3f} ms   graph {t_graph:.3f} ms   equal {torch.equal(out[0], eager[0])}")
