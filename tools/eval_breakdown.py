"""Developer aid: where the evaluation call (scopes.eval_1img of bench.py: one image x 1000 proposals incl. post-processing) spends
its time -- ROIAlign + Res5 + predictor alone, the post-processing alone on fixed predictions, the whole call; HIP events, medians.
usage: python3 tools/eval_breakdown.py [n_images [proposals [classes]]]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench

n_img = int(sys.argv[1]) if len(sys.argv) > 1 else 1
extra = (["--proposals", sys.argv[2]] if len(sys.argv) > 2 else []) + (["--classes", sys.argv[3]] if len(sys.argv) > 3 else [])
args = bench.parse(["--no-cpu-baseline", "--images", str(max(n_img, 1))] + extra)
dev = torch.device("cuda:0")
torch.cuda.set_device(dev)
wl = bench.Workload(args, dev)
h = wl.eval_heads()
feats = {"res4": wl.features["res4"][:n_img]}
props = wl.proposals[:n_img]
boxes = wl.boxes[:n_img]


def med(fn, n=30, warm=5):
    for _ in range(warm):
        fn()
    ts = []
    for _ in range(n):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        e0.record()
        fn()
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    ts.sort()
    return ts[len(ts) // 2]


with torch.no_grad():
    def head_only():
        return h.box_predictor(h._shared_roi_transform([feats["res4"]], boxes, pooled=True))
    preds = head_only()
    t_head = med(head_only)
    t_post = med(lambda: h.box_predictor.inference(preds, props))
    t_all = med(lambda: h(None, feats, props, None))
    t_probs = med(lambda: h.box_predictor.predict_probs(preds, props))
    t_boxes = med(lambda: h.box_predictor.predict_boxes(preds, props))
print(f"{n_img} image(s) x {args.proposals} proposals x {args.classes} classes: ROIAlign + Res5 + mean + predictor {t_head:.3f} ms; "
      f"post-processing alone {t_post:.3f} ms (softmax {t_probs:.3f}, box decoding {t_boxes:.3f}); whole call {t_all:.3f} ms")
