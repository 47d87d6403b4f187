"""Developer aid: one training step of the ROI heads per backend (run on the GPU box)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from oracle import lsm_oracle as oracle
import tests.test_gpu_res5_train as T
import locov_amd

backend = sys.argv[1]
heads, c_in = T._train_heads(locov_amd, oracle, backend, "fp32")
feat = torch.randn(2, c_in, 50, 84, generator=torch.Generator().manual_seed(5)).cuda().requires_grad_(True)
props, targets = T._train_batch(locov_amd, oracle, 2, 60, 5, seed=31)
torch.manual_seed(77)
grid, box_feats, sampled, losses = heads(None, {"res4": feat}, props, targets)
print("forward ok", grid.shape, float(losses["loss_box_reg"]), flush=True)
for name, l in (("box", losses["loss_box_reg"]), ("grid", grid.square().mean()), ("feats", torch.cat(box_feats).square().mean())):
    l.backward(retain_graph=True)
    torch.cuda.synchronize()
    print("backward ok", name, flush=True)
