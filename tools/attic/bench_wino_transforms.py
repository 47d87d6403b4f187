"""Developer aid: one Winograd-domain convolution of the default inference path (split operands, split-layout output) repeated --
run under rocprofv3 --kernel-trace --stats for the per-kernel times of the two transforms and the batched GEMM (tools/attic/ab_wino.sh)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from locov_amd import ops
R = 8000
g = torch.Generator().manual_seed(0)
x = torch.relu(torch.randn(49 * R, 512, generator=g)).cuda()
u = ops.split_pack(ops.winograd_pack_weight((torch.randn(512, 512, 3, 3, generator=g) * 0.02).cuda()))
sc, sh = (torch.rand(512, generator=g) + 0.5).cuda(), torch.randn(512, generator=g).cuda()
for _ in range(12):
    ops.winograd_conv3x3(x, u, scale=sc, shift=sh, relu=True, roi_major=True, in_roi_major=True, out_split_scale=16.0)
torch.cuda.synchronize()
