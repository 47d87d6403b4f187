"""The Res5 stage of the C4 ROI head, with Detectron2's module / checkpoint-key layout.

[D2-upstream] ResNet.make_stage(BottleneckBlock, 3, stride_per_block=[2,1,1], in=1024,
bottleneck=512, out=2048, stride_in_1x1=True, norm="FrozenBN") as built by
ovr/modeling/roi_heads/roi_emb_heads.py:217-241 and applied at :245 (per-region) and :323
(whole grid).  State-dict keys: res5.{0,1,2}.{conv1,conv2,conv3,shortcut}.{weight,norm.*}
(SURVEY.md 8b "Checkpoint keys"), so LocOV.pth loads unchanged.

Row a-3 is outside north_star's hand-written kernel list; the `backend`
switch selects how the convolutions run on the MI355X:
  "miopen" : torch conv2d (MIOpen) -- the stock library path
  "hip"    : the hand-written channels-last MFMA GEMM path of this package (ops_res5)
"""
from __future__ import annotations

from typing import Optional

import torch
import torch.nn.functional as F
from torch import nn


import os

# Block outputs in the split layout (conv3 writes it, the next conv1 stages it by LDS DMA, the next conv3 reads its residual
# from it).  What it buys is the 256 x 256 tile for conv1 (gemm_split_big.hip takes launches with BOTH operands pre-split): at
# 8 000 proposals conv1 goes from 2.17 ms (converting, 128 x 128) to 1.79 ms, each conv3 pays 0.05-0.25 ms for the extra epilogue
# arithmetic (split + pair exchange, decode of the residual): 371 k -> 383 k proposals/s end to end.  Launches too small for the
# big tile lose ~2 % to it -- accepted, because the choice must NOT depend on the batch: the residual then carries 22 instead of 24
# significant bits, and a row's result has to be the same whichever images share its launch (image sharding over ranks;
# tests/test_gpu_roi_heads.py::test_full_size_head_properties).  LOCOV_RES5_OUT_SPLIT=0 turns it off.
_OUT_SPLIT = os.environ.get("LOCOV_RES5_OUT_SPLIT", "1") != "0"
_FUSE12 = os.environ.get("LOCOV_RES5_FUSE12", "1") != "0"      # developer A/B: conv1 + conv2 through ops.conv1x1_winograd_conv3x3
_ONE_LAUNCH_PREP = os.environ.get("LOCOV_RES5_PREP", "1") != "0"    # developer A/B: a training step's operands from one launch (TrainOperands)
_ASYNC_REFRESH = os.environ.get("LOCOV_RES5_SCALE_REFRESH", "async") != "sync"   # developer A/B: "sync" = every 64th step re-chooses the scales on the host-read chain


class FrozenBatchNorm2d(nn.Module):
    """[D2-upstream] FrozenBatchNorm2d: fixed statistics and affine, y = x*scale + shift."""

    def __init__(self, num_features: int, eps: float = 1e-5):
        super().__init__()
        self.num_features = num_features
        self.eps = eps
        self.register_buffer("weight", torch.ones(num_features))
        self.register_buffer("bias", torch.zeros(num_features))
        self.register_buffer("running_mean", torch.zeros(num_features))
        self.register_buffer("running_var", torch.ones(num_features) - eps)

    def scale_shift(self):
        scale = self.weight * (self.running_var + self.eps).rsqrt()
        shift = self.bias - self.running_mean * scale
        return scale, shift

    def forward(self, x):
        scale, shift = self.scale_shift()
        return x * scale.reshape(1, -1, 1, 1) + shift.reshape(1, -1, 1, 1)


class Conv2d(nn.Conv2d):
    """[D2-upstream] detectron2.layers.Conv2d: conv -> norm (sub-module `norm`)."""

    def __init__(self, *args, norm: Optional[nn.Module] = None, **kwargs):
        super().__init__(*args, **kwargs)
        self.norm = norm

    def forward(self, x):
        x = F.conv2d(x, self.weight, self.bias, self.stride, self.padding, self.dilation, self.groups)
        if self.norm is not None:
            x = self.norm(x)
        return x


def get_norm(norm: str, ch: int):
    if norm in ("FrozenBN", "FrozenBatchNorm2d"):
        return FrozenBatchNorm2d(ch)
    if norm in ("", None):
        return None
    if norm == "BN":
        return nn.BatchNorm2d(ch)
    raise ValueError(f"unsupported norm {norm!r} for the Res5 head")


class BottleneckBlock(nn.Module):
    def __init__(self, in_channels, out_channels, *, bottleneck_channels, stride=1, num_groups=1,
                 norm="FrozenBN", stride_in_1x1=True):
        super().__init__()
        self.in_channels, self.out_channels, self.stride = in_channels, out_channels, stride
        self.stride_in_1x1 = stride_in_1x1
        if in_channels != out_channels:
            self.shortcut = Conv2d(in_channels, out_channels, kernel_size=1, stride=stride, bias=False,
                                   norm=get_norm(norm, out_channels))
        else:
            self.shortcut = None
        s1, s3 = (stride, 1) if stride_in_1x1 else (1, stride)
        self.conv1 = Conv2d(in_channels, bottleneck_channels, kernel_size=1, stride=s1, bias=False,
                            norm=get_norm(norm, bottleneck_channels))
        self.conv2 = Conv2d(bottleneck_channels, bottleneck_channels, kernel_size=3, stride=s3, padding=1,
                            bias=False, groups=num_groups, norm=get_norm(norm, bottleneck_channels))
        self.conv3 = Conv2d(bottleneck_channels, out_channels, kernel_size=1, bias=False,
                            norm=get_norm(norm, out_channels))
        for layer in (self.conv1, self.conv2, self.conv3, self.shortcut):
            if layer is not None:   # [D2-upstream] c2_msra_fill
                nn.init.kaiming_normal_(layer.weight, mode="fan_out", nonlinearity="relu")

    def forward(self, x):
        out = F.relu_(self.conv1(x))
        out = F.relu_(self.conv2(out))
        out = self.conv3(out)
        shortcut = self.shortcut(x) if self.shortcut is not None else x
        out = out + shortcut
        return F.relu_(out)


class Res5Stage(nn.Sequential):
    """nn.Sequential of the three bottlenecks (so the state-dict keys stay res5.{0,1,2}....) with an
    additional hand-written forward on channels-last pixel rows.

    forward(x)                    NCHW in / NCHW out through torch conv2d (MIOpen); differentiable; any norm.
    forward_rows(x0, H, W)        x0 [R*H*W, Cin] = the stage input ALREADY sub-sampled by block 0's
                                  stride (the even positions, STRIDE_IN_1X1=True) -> [R*H*W, Cout];
                                  every convolution is an MFMA GEMM of this package with FrozenBN, ReLU
                                  and the residual add fused into its epilogue.  Inference only.
    """

    def __init__(self, *blocks):
        super().__init__(*blocks)
        self._cache = {}
        self._scales = {}           # convolution -> (split-operand scale, uses): see _split
        self._guards = {}           # (kind, device, stream) -> ops.RangeGuard

    def range_guard(self, kind: str, device):
        """This stage's range-guard word of the split arithmetic for `kind` ("fwd": owned by whoever runs a guarded forward;
        "bwd": raised by the backward of res5_train.Res5BlockFn, never reset there) on `device` and the current stream."""
        from . import ops
        key = (kind, torch.device(device), torch.cuda.current_stream(device).cuda_stream)
        g = self._guards.get(key)
        if g is None:
            g = self._guards[key] = ops.RangeGuard(device, deferred=kind in self.DEFERRED_KINDS)
        return g

    # guards nobody reads where they are raised: "bwd" (res5_train.Res5BlockFn.backward) and "fwd_train" (the training forward of the ROI
    # heads) -- the pass zero-fills its results on the device when the word is set, the word is read with the NEXT host read
    DEFERRED_KINDS = ("bwd", "fwd_train")

    def deferred_guards(self, device):
        """[(kind, guard)] of every deferred guard of `device` (their words are read together with another host read)."""
        dev = torch.device(device)
        return [(kind, g) for (kind, d, _), g in self._guards.items() if kind in self.DEFERRED_KINDS and d == dev]

    def backward_guard_words(self, device):
        """Device words of every "bwd" guard of `device` (to be read together with another host read)."""
        return [g.word for kind, g in self.deferred_guards(device) if kind == "bwd"]

    def forget_scales(self) -> None:
        """Drop every remembered split-operand scale and everything packed with one (a range guard tripped: what no longer fits
        may be a remembered scale): the scales are chosen afresh at the next packing, also at unchanged weight versions."""
        self._scales.clear()
        self._cache.clear()
        self.__dict__.pop("_train_ops_by", None)
        self.__dict__.pop("_train_ops", None)

    def backward_guard_tripped(self) -> None:
        """A remembered weight scale stopped covering its weight during a backward: forget the scales (they are chosen
        afresh at the next packing) and clear the words."""
        self.forget_scales()
        for (kind, _, _), g in self._guards.items():
            if kind == "bwd":
                g.reset()

    def backward_guard_raised(self, device) -> bool:
        """One host read: did a backward since the last check trip the range guard?  (Clears it when it did.)"""
        words = self.backward_guard_words(device)
        hit = bool(words) and bool(int(torch.stack([w.reshape(()) for w in words]).max()))
        if hit:
            self.backward_guard_tripped()
        return hit

    def supports_rows_path(self) -> bool:
        b0 = self[0]
        return (all(isinstance(c.norm, FrozenBatchNorm2d) for blk in self for c in
                    (blk.conv1, blk.conv2, blk.conv3)) and b0.stride_in_1x1 and all(blk.conv2.groups == 1 for blk in self))

    def _packed(self, conv: Conv2d, winograd: bool = False):
        """(weight as GEMM operand, scale, shift), re-packed only when a tensor was modified in place or
        re-assigned (checkpoint load, optimizer step).  winograd: the 3x3 weight in the transform domain."""
        from . import ops
        n = conv.norm
        key = (id(conv), conv.weight.data_ptr(), conv.weight._version, n.weight._version, n.bias._version,
               n.running_mean._version, n.running_var._version, n.weight.data_ptr())
        slot = (id(conv), winograd)
        hit = self._cache.get(slot)
        if hit is not None and hit[0] == key:
            return hit[1]
        w = conv.weight.detach()
        if w.shape[2] == 3:
            wp = ops.winograd_pack_weight(w) if winograd else ops.pack_conv3x3_weight(w)
        else:
            wp = w.reshape(w.shape[0], w.shape[1])
        # the FrozenBN fold only depends on the (frozen) statistics: kept across optimizer steps, which only move conv.weight
        fkey = (n.weight._version, n.bias._version, n.running_mean._version, n.running_var._version, n.weight.data_ptr())
        fhit = self._cache.get(("fold", id(n)))
        if fhit is not None and fhit[0] == fkey:
            scale, shift = fhit[1]
        else:
            scale, shift = ops.frozen_bn_fold(n.weight, n.bias, n.running_mean, n.running_var, n.eps)
            self._cache[("fold", id(n))] = (fkey, (scale, shift))
        wp._locov_key = (id(conv), winograd)
        val = (wp, scale, shift)
        self._cache[slot] = (key, val)
        return val

    def _derived(self, conv: Conv2d, tag: str, fn):
        """A weight-sized tensor derived from conv.weight and its FrozenBN fold (transposed / flipped / re-packed filters of
        the backward pass), cached until either changes: the two Res5 calls of a training step (whole grid + sampled
        proposals) run the same backward kernels on the same weights, the second one reuses the first one's operands."""
        n = conv.norm
        key = (conv.weight.data_ptr(), conv.weight._version, n.weight._version, n.bias._version, n.running_mean._version,
               n.running_var._version)
        slot = ("derived", id(conv), tag)
        hit = self._cache.get(slot)
        if hit is not None and hit[0] == key:
            return hit[1]
        val = fn()
        self._cache[slot] = (key, val)
        return val

    def _fold(self, conv: Conv2d):
        """(scale, shift) of conv's FrozenBN (cached until the statistics change; no launch after the first call)."""
        from . import ops
        n = conv.norm
        fkey = (n.weight._version, n.bias._version, n.running_mean._version, n.running_var._version, n.weight.data_ptr())
        fhit = self._cache.get(("fold", id(n)))
        if fhit is not None and fhit[0] == fkey:
            return fhit[1]
        val = ops.frozen_bn_fold(n.weight, n.bias, n.running_mean, n.running_var, n.eps)
        self._cache[("fold", id(n))] = (fkey, val)
        return val

    def train_operands(self, split: bool, grid: bool = True, rois: bool = True) -> "TrainOperands":
        """The GEMM operands of one training step (forward and backward of every convolution), valid for the current weight
        versions: built once per step and FLAVOUR (arithmetic, which 3x3 forms are needed), by the first Res5 call that asks
        (see TrainOperands).  A step that calls the stage through different flavours (res5_grid and res5_rois separately instead
        of one Res5Step) gets one operand set per flavour, each in its own buffers: a later set never re-packs, in place, the
        buffers an earlier set's SplitWeight objects still point at for their backward."""
        versions = tuple((c.weight.data_ptr(), c.weight._version, c.norm.weight._version, c.norm.running_var._version)
                         for blk in self for c in (blk.conv1, blk.conv2, blk.conv3, blk.shortcut) if c is not None)
        flavour = (bool(split), bool(grid), bool(rois))
        by = self.__dict__.setdefault("_train_ops_by", {})
        hit = by.get(flavour)
        if hit is not None and hit[0] == versions:
            self.__dict__["_train_ops"] = (flavour + versions, hit[1])
            return hit[1]
        # a NEW step (the weights moved since the last operand set of any flavour): steps are counted and scale refreshes adopted
        # once per weight version, not once per construction
        new_step = self.__dict__.get("_train_ops_versions") != versions
        self.__dict__["_train_ops_versions"] = versions
        val = TrainOperands(self, split, grid, rois, new_step=new_step)
        by[flavour] = (versions, val)
        self.__dict__["_train_ops"] = (flavour + versions, val)      # (the latest set: tests read its `ready`)
        return val

    def _packed_block0_tail(self):
        """Block 0's conv3 and shortcut as ONE GEMM over the K-concatenated operand [conv2 output | stage
        input]:  relu(s3*(W3 y) + b3 + ss*(Ws x) + bs) = relu([y | x] . [s3*W3 | ss*Ws]^T + (b3 + bs)).
        The FrozenBN scales go into the weight rows (the sum of two differently scaled products cannot use
        the epilogue's single scale).  Returns (Wcat [Cout, mid + Cin], shift)."""
        b0 = self[0]
        w3, s3, b3 = self._packed(b0.conv3)
        ws, ss, bs = self._packed(b0.shortcut)
        key = (id(w3), id(ws), id(s3), id(ss))
        hit = self._cache.get("block0_tail")
        if hit is not None and hit[0] == key:
            return hit[1]
        wcat = torch.cat([w3 * s3[:, None], ws * ss[:, None]], dim=1).contiguous()
        val = (wcat, (b3 + bs).contiguous())
        self._cache["block0_tail"] = (key, val, (w3, ws, s3, ss))      # keep the keyed tensors alive (ids stay unique)
        return val

    def rows_input(self, M: int, device) -> torch.Tensor:
        """Destination for the stage input rows [M, Cin] (ROIAlign writes into it).  When block 0 has a
        projection shortcut it is the right column block of a [M, mid + Cin] matrix whose left block later
        receives block 0's conv2 output, so that conv3 + shortcut + add + ReLU run as one GEMM with
        K = mid + Cin (no separate shortcut tensor, no residual read)."""
        b0 = self[0]
        cin, mid = b0.conv1.in_channels, b0.conv1.out_channels
        if b0.shortcut is None or not self.supports_rows_path() or mid % 4 or cin % 4:
            return torch.empty((M, cin), dtype=torch.float32, device=device)
        buf = torch.empty((M, mid + cin), dtype=torch.float32, device=device)
        x0 = buf[:, mid:]
        x0._locov_cat = buf
        return x0

    def _bf16(self, t: torch.Tensor) -> torch.Tensor:
        """bf16 copy of a packed fp32 weight (cached per tensor object; the packed tensors are themselves cached)."""
        from . import ops
        hit = self._cache.get(("bf16", id(t)))
        if hit is not None and hit[0] is t:
            return hit[1]
        out = ops.to_bf16(t.contiguous())
        self._cache[("bf16", id(t))] = (t, out)
        return out

    def _split(self, t: torch.Tensor):
        """Split-operand packing (ops.split_pack) of a packed fp32 weight, cached per tensor object.
        The power-of-two operand scale of a weight is remembered per convolution (`_locov_key`, set by _packed) and
        re-used while the weights train -- choosing it needs max |w| on the host, i.e. a device sync per packing, ten per
        training step; weights drift slowly against the 8x headroom the scale leaves, the pack kernel raises the
        range-guard word if a re-used scale ever stops covering them (the caller then repeats the pass on the f32 MFMA and
        drops the remembered scales), and every 64 uses the scale is chosen afresh."""
        from . import ops
        hit = self._cache.get(("split", id(t)))
        if hit is not None and hit[0] is t:
            return hit[1]
        key = getattr(t, "_locov_key", None)
        scale = None
        if key is not None:
            rec = self._scales.get(key)
            if rec is not None and rec[1] < 64:
                scale, self._scales[key] = rec[0], (rec[0], rec[1] + 1)
            else:
                scale = ops.split_scale_for(t)
                self._scales[key] = (scale, 0)
        out = ops.split_pack(t.contiguous(), scale)
        if key is not None:
            # ONE live packing per (convolution, tag): a new tensor under the same key (the weights trained, or a per-backward
            # temporary such as the transposed / flipped filters) supersedes the old one, whose buffers are released here
            old = self._cache.pop(("split_of", key), None)
            if old is not None:
                self._cache.pop(("split", old), None)
            self._cache[("split_of", key)] = id(t)
        else:
            stale = [k for k in self._cache if isinstance(k, tuple) and k and k[0] == "split"]
            if len(stale) >= 64:                 # un-keyed tensors (tools, tests): bounded
                for k in stale:
                    del self._cache[k]
        self._cache[("split", id(t))] = (t, out)
        return out

    def _linear(self, split: bool, x, w, bias=None, **kw):
        """One 1x1 convolution / FC as a GEMM: fp32 MFMA, or (split) split-operand f16 MFMA when the shape allows.
        x_is_split=True: x is the split-layout output of the Winograd convolution in front (ACT_SPLIT_SCALE)."""
        from . import ops
        if split and w.shape[1] % 32 == 0 and w.shape[0] % 4 == 0:
            return ops.linear_split(x, self._split(w), bias, **kw)
        assert not (kw.get("x_is_split") or kw.get("out_split") or kw.get("residual_is_split")), \
            "a split-layout activation needs the split GEMM"
        for k in ("x_is_split", "x_scale", "out_split", "residual_is_split"):
            kw.pop(k, None)
        return ops.linear(x, w, bias, **kw)

    def _warn_once(self, key: str, text: str) -> None:
        seen = self.__dict__.setdefault("_warned", set())
        if key not in seen:
            seen.add(key)
            import warnings
            warnings.warn(text, RuntimeWarning, stacklevel=3)

    ACT_SPLIT_SCALE = 16.0       # operand scale of activations (|x| < 4094), also used when a producer writes them pre-split

    def _y2_split_ok(self, split: bool, c2, w3) -> bool:
        """conv2's output can leave the Winograd output transform already in the split layout when its only consumer is the
        split GEMM of conv3."""
        return bool(split) and c2.out_channels % 32 == 0 and w3.shape[1] % 32 == 0 and w3.shape[0] % 4 == 0

    def _out_split_ok(self, split: bool, winograd: bool, bi: int, pooled: bool) -> bool:
        """Block bi's output can leave its last 1x1 convolution in the split layout (never as fp32) when every reader is a split
        GEMM of the next block: its conv1 (pre-split A, staged by LDS DMA) and the identity shortcut in its conv3's epilogue
        (split-layout residual).  Inference only; the stage's final output always stays fp32."""
        if not (split and winograd) or bi + 1 >= len(self) or not _OUT_SPLIT:
            return False
        nxt, cur = self[bi + 1], self[bi]
        ch = cur.conv3.out_channels
        return (nxt.shortcut is None and ch % 32 == 0 and nxt.conv1.out_channels % 4 == 0 and nxt.conv3.out_channels % 8 == 0
                and nxt.conv2.in_channels % 32 == 0 and nxt.conv2.out_channels % 4 == 0)

    def _packed_block0_on_map(self):
        """Weights for running block 0's two 1x1 stride-2 convolutions on the feature MAP (see
        forward_from_map): Wmap = [W1 ; ss*Ws]  ([mid + Cout, Cin]; the shortcut's FrozenBN scale is folded into
        its rows, conv1's FrozenBN is applied after the pooling).  Returns (Wmap, s1, b1, shift_tail = b3 + bs)."""
        b0 = self[0]
        w1, s1, b1 = self._packed(b0.conv1)
        _, _, b3 = self._packed(b0.conv3)
        ws, ss, bs = self._packed(b0.shortcut)
        key = (id(w1), id(ws), id(ss), id(b3), id(bs))
        hit = self._cache.get("block0_map")
        if hit is not None and hit[0] == key:
            return hit[1]
        val = (torch.cat([w1, ws * ss[:, None]], dim=0).contiguous(), s1, b1, (b3 + bs).contiguous())
        self._cache["block0_map"] = (key, val, (w1, ws, ss, b3, bs))
        return val

    def map_path_pays(self, n_rois: int, n_pixels: int) -> bool:
        """forward_from_map runs block 0's 1x1 convolutions on n_pixels map pixels instead of 49*n_rois pooled
        rows, and pools 2.5x as many channels: worth it once the pooled rows clearly outnumber the pixels."""
        b0 = self[0]
        return (b0.shortcut is not None and self.supports_rows_path() and b0.conv1.in_channels % 32 == 0
                and b0.conv1.out_channels % 32 == 0 and 49 * n_rois >= 3 * n_pixels)

    @torch.no_grad()
    def forward_from_map(self, nhwc: torch.Tensor, rois: torch.Tensor, pooler_resolution: int, spatial_scale: float,
                         sampling_ratio: int = 0, aligned: bool = True, winograd: bool = True,
                         bf16: bool = False, split: bool = False, pooled: bool = False, roi_major: bool = False) -> torch.Tensor:
        """The whole stage from the channels-last res4 map [N,H,W,Cin] and the rois [R,5] -> position-major
        rows [49*R, Cout], with block 0's conv1 and projection shortcut moved IN FRONT of the pooler:

            W . ROIAlign(F) == ROIAlign(W . F)      (ROIAlign is linear, the convolutions are 1x1, their stride 2
                                                     is the even-bin grid)

        so both run once per map pixel (N*H*W rows) instead of once per pooled position (49*R rows; 11.7x
        more at 1000 proposals per 1333x800 image): 27 % of the stage's multiply-adds disappear.  conv1's
        FrozenBN + ReLU are applied by the pooling kernel to the pooled value, the pooled shortcut enters
        conv3's epilogue as the residual -- the same arithmetic as the reference up to fp32 re-association."""
        from . import ops
        assert self.supports_rows_path() and self[0].shortcut is not None and pooler_resolution == 14
        b0 = self[0]
        mid = b0.conv1.out_channels
        N, H, W, cin = nhwc.shape
        wmap, s1, b1, shift_tail = self._packed_block0_on_map()
        if bf16:
            g = ops.linear_bf16(ops.to_bf16(nhwc.reshape(N * H * W, cin)), self._bf16(wmap)).view(N, H, W, wmap.shape[0])
        else:
            g = self._linear(split, nhwc.reshape(N * H * W, cin), wmap).view(N, H, W, wmap.shape[0])
        # row order of every [49*R, C] tensor from here on: position-major (pos*R + r; what the direct 3x3 convolution's
        # tap skipping needs) or, with roi_major, ROI-major (r*49 + pos: a ROI's 49 rows are adjacent in memory, which
        # the Winograd transforms and the mean-fused last convolution prefer)
        pm = not (roi_major and not bf16)
        c2 = b0.conv2
        w3, s3, _ = self._packed(b0.conv3)
        use_wino = winograd and c2.in_channels % 32 == 0 and c2.out_channels % 4 == 0
        # pooler + FBN + ReLU + conv2 in one call (the pooled rows never leave the ROIAlign workgroup): split Winograd path, ROI-major
        fuse_pool = _FUSE12 and split and use_wino and not bf16 and not pm and g.dtype == torch.float32
        R = rois.shape[0]
        y = None
        if not fuse_pool:
            y = ops.roi_align_nhwc(g[..., :mid], rois, 14, spatial_scale, sampling_ratio, aligned, bin_stride=2,
                                   pos_major=pm, ch_scale=s1, ch_shift=b1, relu=True).view(49 * R, mid)   # conv1 + FBN + ReLU, pooled
        sc = ops.roi_align_nhwc(g[..., mid:], rois, 14, spatial_scale, sampling_ratio, aligned, bin_stride=2,
                                pos_major=pm).view(49 * R, -1)                               # ss * shortcut, pooled
        if bf16:
            w2, s2, b2 = self._packed(c2)
            y = ops.conv3x3_nhwc_bf16(ops.to_bf16(y), self._bf16(w2), 7, 7, scale=s2, shift=b2, relu=True, pos_major=True)
            x = ops.linear_bf16(ops.to_bf16(y), self._bf16(w3), shift_tail, scale=s3, residual=sc, relu=True)
            return self.forward_rows(x, 7, 7, pos_major=True, start_block=1, bf16=True)
        y_split = False
        if use_wino:
            u2, s2, b2 = self._packed(c2, winograd=True)
            y_split = self._y2_split_ok(split, c2, w3)
            if fuse_pool:
                y = ops.roi_align_winograd_conv3x3(g[..., :mid], rois, 14, spatial_scale, sampling_ratio, aligned, self._split(u2),
                                                   ch_scale=s1, ch_shift=b1, scale2=s2, shift2=b2, relu=True, roi_major=True,
                                                   out_split_scale=self.ACT_SPLIT_SCALE if y_split else None)
            else:
                y = ops.winograd_conv3x3(y, self._split(u2) if split else u2, scale=s2, shift=b2, relu=True,
                                         roi_major=not pm, in_roi_major=not pm,
                                         out_split_scale=self.ACT_SPLIT_SCALE if y_split else None)
        else:
            w2, s2, b2 = self._packed(c2)
            y = ops.conv3x3_nhwc(y, w2, 7, 7, scale=s2, shift=b2, relu=True, pos_major=pm)
        x_split = self._out_split_ok(split, winograd, 0, pooled)
        kw3 = {"x_is_split": True, "x_scale": self.ACT_SPLIT_SCALE} if y_split else {}
        if x_split:
            kw3.update(out_split=True, x_scale=self.ACT_SPLIT_SCALE)
        x = self._linear(split, y, w3, shift_tail, scale=s3, residual=sc, relu=True, **kw3)                 # conv3 + FBN + add + ReLU
        return self.forward_rows(x, 7, 7, pos_major=pm, winograd=winograd, start_block=1, split=split, pooled=pooled,
                                 x0_is_split=x_split)

    @torch.no_grad()
    def forward_rows(self, x0: torch.Tensor, H: int, W: int, pos_major: bool = False,
                     winograd: bool = True, start_block: int = 0, bf16: bool = False, split: bool = False,
                     pooled: bool = False, x0_is_split: bool = False) -> torch.Tensor:
        """Rows are ROI-major (r*H*W + pos) or position-major (pos*R + r); the 1x1 convolutions do not
        care.  The 3x3 one runs, on 7x7 position-major tiles, in the Winograd domain (121 instead of 361
        products per tile and channel pair; `winograd=False` keeps the direct implicit GEMM, which skips
        the zero-padding taps in the position-major order).
        pooled: return the spatial mean [R, Cout] of the stage output instead of its rows (what the box head consumes,
        roi_emb_heads.py:262,344,356).  With split arithmetic on the Winograd path the mean is fused into the last 1x1
        convolution (ops.linear_split_segmean: the [49*R, Cout] output is neither written nor re-read)."""
        from . import ops
        assert self.supports_rows_path(), "forward_rows needs FrozenBN, STRIDE_IN_1X1 and ungrouped convs"
        x = x0
        x_split = bool(x0_is_split)                      # x holds the previous block's output in the split layout (_out_split_ok)
        assert not x_split or (split and start_block > 0)
        cat = getattr(x0, "_locov_cat", None)            # rows_input(): x0 is the right block of [conv2 out | x0]
        for bi, blk in enumerate(self):
            if bi < start_block:
                continue
            w1, s1, b1 = self._packed(blk.conv1)
            w3, s3, b3 = self._packed(blk.conv3)
            c2 = blk.conv2
            if bf16:
                # opt-in reduced precision: bf16 GEMM operands (direct tap-skipping 3x3), fp32 accumulate, fp32
                # FrozenBN / ReLU / residual and an fp32 residual stream
                w2, s2, b2 = self._packed(c2)
                y = ops.linear_bf16(ops.to_bf16(x.contiguous()), self._bf16(w1), b1, scale=s1, relu=True)
                y = ops.conv3x3_nhwc_bf16(ops.to_bf16(y), self._bf16(w2), H, W, scale=s2, shift=b2, relu=True,
                                          pos_major=pos_major)
                if blk.shortcut is not None:
                    ws, ss, bs = self._packed(blk.shortcut)
                    sc = ops.linear_bf16(ops.to_bf16(x.contiguous()), self._bf16(ws), bs, scale=ss)
                else:
                    sc = x
                x = ops.linear_bf16(ops.to_bf16(y), self._bf16(w3), b3, scale=s3, residual=sc, relu=True)
                continue
            xs_kw = {"x_is_split": True, "x_scale": self.ACT_SPLIT_SCALE} if x_split else {}
            use_wino = winograd and H == 7 and W == 7 and c2.in_channels % 32 == 0 and c2.out_channels % 4 == 0
            assert use_wino or not x_split
            rm = not pos_major                     # the Winograd transforms read / write either row order
            # conv1 and conv2 in one call where the block input is already in the split layout: conv1's epilogue then writes the
            # Winograd-domain tensor itself (ops.conv1x1_winograd_conv3x3; the same bits as the two calls)
            fuse12 = _FUSE12 and x_split and use_wino and rm and w1.shape[0] % 32 == 0

            def conv12(**kw):
                u2, s2, b2 = self._packed(c2, winograd=True)
                if fuse12:
                    return ops.conv1x1_winograd_conv3x3(x, self._split(w1), b1, self._split(u2), scale1=s1, scale2=s2, shift2=b2,
                                                        relu=True, x_scale=self.ACT_SPLIT_SCALE, roi_major=kw["roi_major"],
                                                        out_split_scale=kw.get("out_split_scale"))
                return ops.winograd_conv3x3(y, self._split(u2) if split else u2, scale=s2, shift=b2, relu=True, in_roi_major=rm, **kw)

            y = None if fuse12 else self._linear(split, x, w1, b1, scale=s1, relu=True, **xs_kw)      # 1x1 (+stride via x0) + FBN + ReLU
            if use_wino and bi == 0 and cat is not None and blk.shortcut is not None:
                u2, s2, b2 = self._packed(c2, winograd=True)
                ops.winograd_conv3x3(y, self._split(u2) if split else u2, scale=s2, shift=b2, relu=True,
                                     out=cat[:, :c2.out_channels], roi_major=rm, in_roi_major=rm)
                wcat, bcat = self._packed_block0_tail()
                x = self._linear(split, cat, wcat, bcat, relu=True)               # conv3 + shortcut + add + ReLU, K-concatenated
                continue
            last = bi == len(self) - 1
            if use_wino and pooled and last and split and blk.shortcut is None and w3.shape[1] % 32 == 0 and w3.shape[0] % 4 == 0:
                ysp = self._y2_split_ok(True, c2, w3)
                if not ops.segmean_supported(x.shape[0], w3.shape[0], w3.shape[1], H * W, residual_roi_major=rm, x_is_split=ysp,
                                             residual_is_split=x_split):
                    # (the 128 x 128 form of the mean-fused convolution addresses its residual with 32-bit offsets; the 256 x 256
                    # form, which every call of thousands of proposals takes, does not)
                    self._warn_once("segmean", f"Res5Stage.forward_rows: the spatial mean is NOT fused into the last convolution for "
                                    f"{x.shape[0]} rows x {w3.shape[0]} channels (too large for the 128x128 mean-fused kernel and not a "
                                    "256x256 launch): the [rows, channels] tensor is written and read once more (~10 % slower)")
                    ysp = None
            else:
                ysp = None
            if ysp is not None:
                y = conv12(roi_major=True, out_split_scale=self.ACT_SPLIT_SCALE if ysp else None)
                return ops.linear_split_segmean(y, self._split(w3), b3, x, H * W, scale=s3, relu=True, residual_roi_major=rm,
                                                x_is_split=ysp, x_scale=self.ACT_SPLIT_SCALE, residual_is_split=x_split)
            y_split = False
            if use_wino:
                y_split = self._y2_split_ok(split, c2, w3)
                y = conv12(roi_major=rm, out_split_scale=self.ACT_SPLIT_SCALE if y_split else None)          # 3x3 + FBN + ReLU
            else:
                w2, s2, b2 = self._packed(c2)
                y = ops.conv3x3_nhwc(y, w2, H, W, scale=s2, shift=b2, relu=True, pos_major=pos_major)
            res_split = False
            if blk.shortcut is not None:
                ws, ss, bs = self._packed(blk.shortcut)
                sc = self._linear(split, x, ws, bs, scale=ss, **xs_kw)            # 1x1 shortcut + FBN
            else:
                sc, res_split = x, x_split
            out_split = use_wino and y_split and self._out_split_ok(split, winograd, bi, pooled)
            x = self._linear(split, y, w3, b3, scale=s3, residual=sc, relu=True,
                             **({"x_is_split": True, "x_scale": self.ACT_SPLIT_SCALE} if y_split else {}),
                             **({"out_split": True} if out_split else {}),
                             **({"residual_is_split": True} if res_split else {}))   # 1x1 + FBN + add + ReLU
            x_split = out_split
        if pooled:
            R = x.shape[0] // (H * W)
            return ops.spatial_mean(x.view(H, W, R, x.shape[1]), channels_last=2) if pos_major else \
                ops.spatial_mean(x.view(R, H, W, x.shape[1]), channels_last=1)
        return x


class TrainOperands:
    """Every weight-derived GEMM operand of ONE training step of the stage.

        get(conv, tag) -> ops.SplitWeight (split arithmetic, eligible shape) | fp32 tensor
        tags:  "plain"  W [N,K]                         forward of a 1x1 convolution
               "t"      (s W)^T [K,N]                   its data gradient
               "wino"   U = (G (x) G) w [121,N,Cin]     forward of a 3x3 convolution on 7x7 tiles
               "col"    [N, 9 Cin]                      ... on a general grid (im2col GEMM)
               "uflip"  (G (x) G) flip(s w) [121,Cin,N] data gradient on 7x7 tiles
               "flip9"  [Cin, 9 N] of flip(s w)         ... on a general grid

    In split arithmetic, once every operand's power-of-two scale is remembered (Res5Stage._scales: first chosen from max |.| on
    the host, 8x headroom; the pack kernel raises the range guard if one stops covering its data), ALL of them come out of ONE
    launch (ops.res5_weight_prep) into buffers the stage keeps -- enqueued by the step's first Res5 call.  Otherwise (first step,
    the f32 MFMA, odd shapes) each operand is built on demand by the multi-launch chain it replaces (_packed / _derived /
    _split), which also chooses the scales.
    Every REFRESH steps the scales are chosen again WITHOUT a host wait: max |w| of every weight and max |s| of every FrozenBN
    scale go to pinned memory behind an event (two launches), the step two later derives each operand's
    scale from the BOUND  max |operand| <= max |s| * max |w| (* WINO_GAIN in the Winograd domain)  and the steps in between keep
    the old scales, which the headroom still covers.  (The chain's refresh read the exact max of each operand on the host: 26
    waits and ~80 launches, a 5 ms step every 64 -- LOCOV_RES5_SCALE_REFRESH=sync keeps that form.)"""

    REFRESH = 64
    WINO_GAIN = 2.25          # max_f (sum_a |G[f][a]|)^2 of csrc/winograd_tables.h: |((G (x) G) w)[f]| <= 2.25 max |w|

    def __init__(self, stage: "Res5Stage", split: bool, grid: bool, rois: bool, new_step: bool = True):
        from . import ops
        self.stage, self.split = stage, bool(split)
        self.ready = {}
        if not self.split:
            return
        wanted = []
        for blk in stage:
            for conv in (blk.conv1, blk.conv3, blk.shortcut):
                if conv is not None:
                    wanted += [(conv, "plain"), (conv, "t")]
            c2 = blk.conv2
            if rois:
                wanted += [(c2, "wino"), (c2, "uflip")]
            if grid:
                wanted += [(c2, "col"), (c2, "flip9")]
        if _ASYNC_REFRESH and new_step:
            self._adopt_refresh()
        recs = [stage._scales.get(self.scale_key(conv, tag)) for conv, tag in wanted]
        # (async refresh: a remembered scale stays usable past REFRESH -- the new one is on its way; sync: it sends the step to the chain)
        fresh = [r is not None and (_ASYNC_REFRESH or r[1] < self.REFRESH) for r in recs]
        ok = _ONE_LAUNCH_PREP and any(fresh) and all(
            conv.weight.is_cuda and conv.weight.dtype == torch.float32 and conv.weight.is_contiguous()
            and conv.in_channels % 32 == 0 and conv.out_channels % 32 == 0 and conv.groups == 1 for conv, _ in wanted)
        if not ok:
            return
        if not all(fresh):
            # SOME scales are remembered: the others belong to operands the previous steps never asked for (block 0's data
            # gradient when the stage input needs none: a frozen backbone) -- on the on-demand chain they would stay unknown for
            # good and keep every step off the one launch.  They are built here by that chain (one host read each, once per
            # REFRESH steps).
            for (conv, tag), f in zip(wanted, fresh):
                if not f:
                    self.get(conv, tag)
            recs = [stage._scales.get(self.scale_key(conv, tag)) for conv, tag in wanted]
            if not all(r is not None and (_ASYNC_REFRESH or r[1] < self.REFRESH) for r in recs):
                return
        if _ASYNC_REFRESH and any(r[1] >= self.REFRESH for r in recs):
            self._start_refresh(wanted)
        bufs = stage.__dict__.setdefault("_prep_bufs", {})
        jobs = []
        for (conv, tag), rec in zip(wanted, recs):
            w = conv.weight.detach()
            shape = ops.prep_shape(tag, w)
            bk = (id(conv), tag)
            fk = (bool(grid), bool(rois)) + bk                # (per flavour: see Res5Stage.train_operands)
            buf = bufs.get(fk)
            if buf is None or tuple(buf.shape) != shape or buf.device != w.device:
                buf = bufs[fk] = torch.empty(shape, dtype=torch.float32, device=w.device)
            rs = stage._fold(conv)[0] if tag in ("t", "uflip", "flip9") else None
            jobs.append((tag, w, rs, buf, rec[0]))
            if new_step:
                stage._scales[self.scale_key(conv, tag)] = (rec[0], rec[1] + 1)
            self.ready[bk] = ops.SplitWeight(buf, rec[0])
        ops.res5_weight_prep(jobs)

    def _start_refresh(self, wanted) -> None:
        """Enqueue max |w| / max |s| of every convolution of `wanted` towards pinned memory (no wait); one refresh in flight."""
        stage = self.stage
        if stage.__dict__.get("_scale_refresh") is not None:
            return
        convs = list({id(c): c for c, _ in wanted}.values())
        ts = [c.weight.detach() for c in convs] + [stage._fold(c)[0] for c in convs]
        try:
            norms = torch._foreach_norm(ts, float("inf"))
        except (RuntimeError, TypeError):                    # (a torch without the foreach form of the max norm)
            norms = [t.abs().max() for t in ts]
        dev = torch.stack([n.reshape(()).to(torch.float32) for n in norms])
        host = stage.__dict__.get("_scale_refresh_host")     # (one refresh in flight, adopted before the next starts: one buffer)
        if host is None or host.numel() < dev.numel():
            host = stage.__dict__["_scale_refresh_host"] = torch.empty(max(dev.numel(), 64), dtype=torch.float32).pin_memory()
        host = host[:dev.numel()]
        host.copy_(dev, non_blocking=True)
        event = torch.cuda.Event()
        event.record(torch.cuda.current_stream(dev.device))
        stage.__dict__["_scale_refresh"] = (event, host, [id(c) for c in convs], list(wanted), stage.__dict__.get("_prep_steps", 0))

    def _adopt_refresh(self) -> None:
        """Two steps after a refresh was enqueued: every operand's scale from the bound on its max |.| (counts start again)."""
        import math
        stage = self.stage
        steps = stage.__dict__["_prep_steps"] = stage.__dict__.get("_prep_steps", 0) + 1
        pending = stage.__dict__.get("_scale_refresh")
        # adopted a FIXED two steps after it was enqueued (its event completed a step ago -- every step waits for the GPU once --
        # so the wait below is free): the schedule of scales does not depend on timing
        if pending is None or steps - pending[4] < 2:
            return
        pending[0].synchronize()
        stage.__dict__["_scale_refresh"] = None
        _, host, ids, wanted, _ = pending
        vals = host.tolist()
        n = len(ids)
        amax_w, amax_s = dict(zip(ids, vals[:n])), dict(zip(ids, vals[n:]))
        for conv, tag in wanted:
            key = self.scale_key(conv, tag)
            if key not in stage._scales:                     # (dropped in between -- a guard tripped: the chain chooses afresh)
                continue
            bound = amax_w[id(conv)] * (amax_s[id(conv)] if tag in ("t", "uflip", "flip9") else 1.0) * (
                self.WINO_GAIN if tag in ("wino", "uflip") else 1.0)
            scale = 2.0 ** (12 - math.floor(math.log2(bound))) if bound > 0 and math.isfinite(bound) else 1.0
            stage._scales[key] = (min(max(scale, 2.0 ** -100), 2.0 ** 100), 0)

    @staticmethod
    def scale_key(conv, tag):
        """The key Res5Stage._split remembers this operand's scale under (set by _packed / the backward's `keyed`)."""
        return (id(conv), {"plain": False, "col": False, "wino": True}.get(tag, tag))

    def get(self, conv, tag: str):
        hit = self.ready.get((id(conv), tag))
        if hit is not None:
            return hit
        from . import ops
        st = self.stage
        if tag in ("plain", "col", "wino"):
            t = st._packed(conv, winograd=tag == "wino")[0]
        else:
            s = st._fold(conv)[0]
            if tag == "t":
                w = st._packed(conv)[0]
                t = st._derived(conv, "wt", lambda: ops.weight_transpose_scale(w, s))
            else:
                wflip = st._derived(conv, "flip", lambda: ops.conv3x3_weight_flip(conv.weight.detach(), s))     # [Cin, Cout, 3, 3]
                if tag == "uflip":
                    t = st._derived(conv, "uflip", lambda: ops.winograd_pack_weight(wflip))
                else:
                    t = st._derived(conv, "flip9", lambda: ops.pack_conv3x3_weight(wflip))
            t._locov_key = (id(conv), tag)                  # remembered operand scale of a per-step packing (Res5Stage._split)
        if self.split and t.shape[-1] % 32 == 0 and t.shape[-2] % 4 == 0:
            return st._split(t)
        return t


def build_res5_block(cfg):
    """roi_emb_heads.py:217-241 _build_res5_block."""
    stage_channel_factor = 2 ** 3
    num_groups = cfg.MODEL.RESNETS.NUM_GROUPS
    width_per_group = cfg.MODEL.RESNETS.WIDTH_PER_GROUP
    bottleneck_channels = num_groups * width_per_group * stage_channel_factor
    out_channels = cfg.MODEL.RESNETS.RES2_OUT_CHANNELS * stage_channel_factor
    stride_in_1x1 = cfg.MODEL.RESNETS.STRIDE_IN_1X1
    norm = cfg.MODEL.RESNETS.NORM
    assert not cfg.MODEL.RESNETS.DEFORM_ON_PER_STAGE[-1], "Deformable conv is not yet supported in res5 head."
    blocks = []
    in_ch = out_channels // 2
    for stride in (2, 1, 1):
        blocks.append(BottleneckBlock(in_ch, out_channels, bottleneck_channels=bottleneck_channels, stride=stride,
                                      num_groups=num_groups, norm=norm, stride_in_1x1=stride_in_1x1))
        in_ch = out_channels
    return Res5Stage(*blocks), out_channels
