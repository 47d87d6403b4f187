"""Developer aid: per-tensor gradient errors of the Res5 rows path against float64 autograd (run on the GPU box)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from oracle import lsm_oracle as oracle
import tests.test_gpu_res5_train as T
import locov_amd
from locov_amd import res5_train

dims, R = ((128, 64, 256), 21) if len(sys.argv) < 2 else ((1024, 512, 2048), 12)
in_ch, mid, out_ch = dims
res5, params = T._stage(locov_amd, oracle, in_ch, mid, out_ch, seed=R)
gen = torch.Generator().manual_seed(17)
x14 = torch.randn(R, in_ch, 14, 14, generator=gen)
gy = torch.randn(R, out_ch, generator=gen)
xd = x14.double().requires_grad_(True)
yd, pd = T._float64_stage(oracle, params, xd)
(yd.mean(dim=[2, 3]) * gy.double()).sum().backward()
x0 = x14[:, :, ::2, ::2].permute(0, 2, 3, 1).reshape(R * 49, in_ch).contiguous().cuda().requires_grad_(True)
out = res5_train.res5_rows(res5, x0, R, 7, 7, pooled=True, split=False)
(out * gy.cuda()).sum().backward(retain_graph=True)
want_x = xd.grad[:, :, ::2, ::2].permute(0, 2, 3, 1).reshape(R * 49, in_ch)
print("fwd", T.rel_err(out.detach(), yd.mean(dim=[2, 3]).detach()))
print("x0", T.rel_err(x0.grad, want_x))
sd = dict(res5.named_parameters())
for k in T._weight_keys(params):
    print(k, T.rel_err(sd[k].grad, pd[k].grad))

# ---- isolate the last block
from locov_amd import ops
saved = out.grad_fn.saved_tensors
x2, y1, y2, o2 = [t.detach() for t in saved[8:12]]
ref_out = yd.detach().permute(0, 2, 3, 1).reshape(R * 49, out_ch)
print("out rows err", T.rel_err(o2, ref_out), "mask mismatches", int(((o2.cpu() > 0) != (ref_out > 0)).sum()))
g = ops.spatial_mean_bwd(gy.cuda(), o2, 49)
g_ref = torch.where(ref_out > 0, (gy.double() / 49).repeat_interleave(49, 0), torch.zeros_like(ref_out))
print("g err", T.rel_err(g, g_ref))
blk = res5[2]
w3, s3, _ = res5._packed(blk.conv3)
dw3 = ops.gemm_tn(g, y2, s3)
dw3_same = (g.double().t() @ y2.double()) * s3.double()[:, None]
print("gemm_tn vs float64 on the same operands", T.rel_err(dw3, dw3_same))
print("dw3 vs autograd", T.rel_err(dw3, pd["2.conv3.weight"].grad.view(out_ch, mid)))
print("s3 vs oracle", T.rel_err(s3, params["2.conv3.norm.weight"].double() * (params["2.conv3.norm.running_var"].double() + 1e-5).rsqrt()))
