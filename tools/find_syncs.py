"""Developer aid: host synchronisations inside one inference step / one training step (torch.cuda.set_sync_debug_mode('warn'))."""
import sys, os, warnings
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
mode = sys.argv[1] if len(sys.argv) > 1 else "infer"
args = bench.parse(["--mode", "train"] if mode not in ("infer", "eval", "eval_torch_chain") else [])
if mode == "stt":
    args.train_config = "stt"
dev = torch.device("cuda")
if mode == "infer":
    wl = bench.Workload(args, dev)
    step = lambda: wl.step_s2()
elif mode in ("eval", "eval_torch_chain"):          # the evaluation call (one image incl. post-processing); _torch_chain: round 5's path
    wl = bench.Workload(bench.parse([]), dev)
    if mode == "eval_torch_chain":
        from locov_amd.roi_heads import box_emb_head
        box_emb_head._FUSED_POSTPROCESS = False
    step = lambda: wl.step_eval(1)
else:
    tw = bench.TrainWorkload(args, dev, "hip", 1)
    step = tw.step
step(); step(); torch.cuda.synchronize()
torch.cuda.set_sync_debug_mode("warn")
with warnings.catch_warnings(record=True) as w:
    warnings.simplefilter("always")
    step()
torch.cuda.set_sync_debug_mode("default")
import collections
c = collections.Counter()
for x in w:
    if "synchroniz" in str(x.message):
        c[f"{x.filename.replace(os.path.dirname(os.path.dirname(os.path.abspath(__file__))) + '/', '')}:{x.lineno}"] += 1
print(mode, "syncs per step:", sum(c.values()))
for k, v in c.most_common():
    print(f"  {v:3d}  {k}")
