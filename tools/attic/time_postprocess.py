"""Developer aid: time of the post-processing tail (softmax, box decoding, threshold, class-wise NMS, top-k: fast_rcnn_inference,
roi_emb_heads.py:280,357) next to the logits path."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, bench
args = bench.parse(sys.argv[1:])
dev = torch.device("cuda")
wl = bench.Workload(args, dev)
heads = wl.heads
feats = wl.features
def timed(fn, n=10):
    fn(); fn(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
with torch.no_grad():
    t_logits = timed(wl.step_s2)
    t_full = timed(lambda: heads(None, feats, wl.proposals, None))
    x = heads._shared_roi_transform([feats["res4"]], [p.proposal_boxes for p in wl.proposals], pooled=True)
    pred = heads.box_predictor(x)
    t_post = timed(lambda: heads.box_predictor.inference(pred, wl.proposals))
    inst, _ = heads.box_predictor.inference(pred, wl.proposals)
print(f"logits path {t_logits:.2f} ms   full inference_detection {t_full:.2f} ms   post-processing alone {t_post:.2f} ms   detections/img {[len(i) for i in inst][:4]}")
