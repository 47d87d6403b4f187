"""Developer A/B: the LSM / STT training step of bench.py with the one-launch loss tails (ops.box_reg_loss, ops.grounding_ce) on / off."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench
from locov_amd import ops
from locov_amd.roi_heads import box_emb_head

args = bench.parse([])
dev = torch.device("cuda:0")
torch.cuda.set_device(dev)
for cfg in ("lsm", "stt"):
    for box, ce in ((True, True), (False, False), (True, False), (False, True)):
        ops.GROUNDING_CE_MAX_B = 64 if ce else 0
        box_emb_head._FUSED_BOX_LOSS = box
        tw = bench.TrainWorkload(args, dev, "hip", 1, config=cfg)
        losses = []
        for _ in range(8):
            tw.step()
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0 = time.time()
        a.record()
        for _ in range(20):
            tw.step()
        b.record(); torch.cuda.synchronize()
        wall = (time.time() - t0) / 20 * 1e3
        loss, _ = tw.forward_backward()
        print(f"{cfg}: box fused {box}, ce fused {ce}: {a.elapsed_time(b) / 20:.2f} ms/step (wall {wall:.2f}), loss after 28 steps {float(loss):.6f}", flush=True)
        del tw
        torch.cuda.empty_cache()
