"""The Res5 stage under autograd on the hand-written gfx950 kernels (forward + data gradients + weight gradients).

The LSM configuration trains the Res5 convolution weights (configs/coco_lsm.yaml:8 `FREEZE_AT: 0`; FrozenBN only
freezes the statistics) through two call sites of ovr/modeling/roi_heads/roi_emb_heads.py:
    :323      visual_grid_features = self.res5(features)            -- the whole res4 grid
    :343-344  box_features = self._shared_roi_transform(...).mean() -- the sampled proposals (4 x 200 per GPU)
The reference differentiates both through cuDNN (torch autograd).  Here every tensor is a channels-last pixel-row
matrix [R*H*W, C] (ROI-major rows) and one torch.autograd.Function covers the whole stage:

  forward   per bottleneck: 1x1 (GEMM, FrozenBN + ReLU in the epilogue) -> 3x3 (Winograd domain for 7x7 tiles, implicit
            GEMM otherwise) -> 1x1 (+ projection shortcut / identity residual, ReLU); the post-ReLU activations are kept.
  backward  with g the gradient of a block's output already masked by (output > 0):
              dW3 = s3 * g^T y2                      TN GEMM over the pixel rows (gemm_tn.hip)
              g2  = (g . s3 W3) * [y2 > 0]           NT GEMM, mask fused into the epilogue (gemm_nt.hip)
              dW2 = s2 * wgrad3x3(y1, g2)            Winograd domain (121 TN GEMMs) or im2col + TN GEMM
              g1  = conv3x3(g2, flip(s2 W2)) * [y1 > 0]
              dW1 = s1 * g1^T x ;  dWs = ss * g^T x
              gx  = (g1 . s1 W1 + g [. ss Ws]) * [x > 0]      = the masked gradient of the previous block's output
            FrozenBN scales are folded into the transposed weights (data gradients) and applied to the rows of the
            weight gradients; the statistics receive no gradient (buffers), exactly as in the reference.

Arithmetic: with `split` (RES5_DTYPE "f16x2") the forward AND the backward GEMMs form every fp32 product from (hi, lo)
f16 pairs on the f16 matrix pipe with fp32 accumulation (gemm_split.hip, gemm_tn_split.hip); a gradient tensor has no
a-priori range, so its power-of-two operand scale is chosen on the device from its max |.| (one small reduction per
gradient tensor, no host read).  Without it everything runs on the f32 MFMA.  LOCOV_RES5_BWD_F32=1 keeps the backward on
the f32 MFMA while the forward uses split operands (developer A/B).
"""
from __future__ import annotations

import os
from typing import List, NamedTuple, Optional, Sequence, Tuple

import torch

from . import ops

__all__ = ["res5_rows", "res5_grid", "res5_rois", "roi_align_even_rows", "to_nhwc", "Res5BlockFn", "Res5OutputFn", "Res5Step", "Segment",
           "saved_activations"]

_BWD_STREAMS = bool(int(os.environ.get("LOCOV_RES5_BWD_STREAMS", "1")))    # the grid segment's 3x3 gradients on a side stream (0: one stream)
_SIDE_STREAMS = {}


def _side_stream(device) -> "torch.cuda.Stream":
    key = (torch.device(device), torch.cuda.current_stream(device).cuda_stream)
    st = _SIDE_STREAMS.get(key)
    if st is None:
        st = _SIDE_STREAMS[key] = torch.cuda.Stream(device=device)
    return st


_KEEP_V = bool(int(os.environ.get("LOCOV_RES5_KEEP_V", "1")))               # the forward's Winograd-domain input kept for the weight gradient (0: transformed again)
_NO_WINO_BWD = bool(int(os.environ.get("LOCOV_RES5_BWD_DIRECT", "0")))      # developer A/B: 3x3 gradients in the direct form
_BWD_F32 = bool(int(os.environ.get("LOCOV_RES5_BWD_F32", "0")))              # developer A/B: backward GEMMs on the f32 MFMA


def _wino_ok(H: int, W: int, cin: int, cout: int) -> bool:
    return H == 7 and W == 7 and cin % 32 == 0 and cout % 4 == 0


class Segment(NamedTuple):
    """The rows of one Res5 call inside a step's matrices: rows [row0, row0 + n * H * W) are n independent H x W grids (ROI tiles
    or whole image grids), ROI-major."""
    row0: int
    n: int
    H: int
    W: int

    @property
    def rows(self) -> int:
        return self.n * self.H * self.W

    @property
    def wino(self) -> bool:
        return self.H == 7 and self.W == 7


class Res5Step:
    """The Res5 work of ONE training step.  The LSM step calls the stage twice with the same weights -- on the whole res4
    grid (roi_emb_heads.py:323, 4 200 rows per GPU) and on the sampled proposals (:343-344, 39 200 rows) -- and every 1x1
    convolution's data gradient and weight gradient is a GEMM over pixel rows that does not care which call a row came
    from.  So the rows of all calls of a step live in ONE matrix per activation (a `Segment` each); the forwards run per
    segment as soon as their input exists, or -- when every input is there before anything has to wait for the host
    (the ROI heads form the sample on the device) -- together, with the 1x1 convolutions sharing launches
    (forward_segments); and the backward runs ONCE over the joint rows: one launch per 1x1 data gradient and weight
    gradient instead of one per call (the whole-grid launches are too small to fill the chip: 2.7 ms for a tenth of the
    rows), one weight gradient per parameter instead of two that autograd has to add.  Only the 3x3 convolutions differ per
    segment (Winograd domain for 7x7 tiles, im2col GEMM on a general grid).

        step = Res5Step(stage, split, device, capacity_rows)
        x = step.input_rows(n * H * W)        # where the producer (ROIAlign, rows_stride2) writes the segment's stage input
        step.forward(n, H, W)                 # the stage on that segment (no autograd), outputs stay in the step's matrices
        ...                                   # (or: input_rows() of several segments, then forward_segments([(n, H, W), ...]))
        outs = step.outputs([x_with_grad, ...], [pooled, ...])       # ONE autograd node for all segments
    """

    def __init__(self, stage, split: bool, device, capacity_rows: int, grid: bool = True, rois: bool = True):
        self.stage, self.split, self.device = stage, bool(split), torch.device(device)
        # every weight-derived operand of the step's forward AND backward (one launch in split arithmetic: Res5Stage.train_operands)
        self.operands = stage.train_operands(self.split, grid=grid, rois=rois)
        self.capacity = int(capacity_rows)
        self.segments: List[Segment] = []
        self.filled = 0
        self._pending: List[int] = []
        new = lambda c: torch.empty((self.capacity, c), dtype=torch.float32, device=self.device)
        self.x0 = new(stage[0].conv1.in_channels)
        # per block: y1 (conv1 + FBN + ReLU), y2 (conv2 + FBN + ReLU), out (conv3 + FBN + shortcut + ReLU) -- the post-ReLU
        # activations the backward needs as masks and as weight-gradient operands
        self.act = [(new(b.conv1.out_channels), new(b.conv2.out_channels), new(b.conv3.out_channels)) for b in stage]
        self.cols = {}                                # (block, segment index) -> im2col patches of y1 (general-grid segments)
        # (block, segment index) -> the workspace of that 3x3 convolution's Winograd-domain forward (split arithmetic): it starts
        # with the transformed y1 in the split layout, which the weight gradient's TN GEMMs take as it is
        self.wino_ws = {}

    def input_rows(self, rows: int) -> torch.Tensor:
        """Where the producer of the NEXT segment's stage input writes it (several segments may be reserved before one forward)."""
        row0 = self.filled + sum(self._pending)
        if row0 + rows > self.capacity:
            raise ValueError(f"Res5Step: {row0} + {rows} rows exceed the capacity of {self.capacity}")
        self._pending.append(rows)
        return self.x0[row0:row0 + rows]

    @torch.no_grad()
    def forward(self, n: int, H: int, W: int, on_range_final=None) -> Segment:
        """The stage on the segment whose input was just written into input_rows(n * H * W).
        on_range_final: called (split arithmetic, 7x7 tiles) in front of the LAST convolution's launch, at which point every
        value a split GEMM of this segment will read has been range-checked (the last block's conv2 checks its own output) --
        a caller that waits for the range guard records its event there instead of behind the stage."""
        return self.forward_segments([(n, H, W)], on_range_final)[0]

    @torch.no_grad()
    def forward_segments(self, geoms: Sequence[Tuple[int, int, int]], on_range_final=None) -> List[Segment]:
        """The stage on ALL the segments whose inputs are waiting in input_rows (geoms: their (n, H, W) in reservation order) in
        ONE pass: the 1x1 convolutions run over the joint rows (one launch each instead of one per segment -- the whole-grid
        call's 4 200 rows alone fill half the chip), the 3x3 convolutions per segment.  The last block's last convolution stays
        per segment, the general-grid segments first, so that on_range_final (see forward) can still be called in front of the
        7x7 segment's."""
        assert [g[0] * g[1] * g[2] for g in geoms] == self._pending, "Res5Step.forward: input_rows() of every segment first"
        new = []
        for (n, H, W) in geoms:
            seg = Segment(self.filled, n, H, W)
            new.append((len(self.segments), seg))
            self.segments.append(seg)
            self.filled += seg.rows
        self._pending = []
        live = [(si, seg) for si, seg in new if seg.rows > 0]
        if not live:
            return [seg for _, seg in new]
        stage, T = self.stage, self.operands
        lo, hi = live[0][1].row0, live[-1][1].row0 + live[-1][1].rows
        joint = slice(lo, hi)
        x = self.x0[joint]
        nb = len(stage)
        for bi, blk in enumerate(stage):
            Y1, Y2, OUT = self.act[bi]
            s1, b1 = stage._fold(blk.conv1)
            s2, b2 = stage._fold(blk.conv2)
            s3, b3 = stage._fold(blk.conv3)
            c2 = blk.conv2
            _linear(x, T.get(blk.conv1, "plain"), b1, scale=s1, relu=True, out=Y1[joint])
            if blk.shortcut is not None:
                ss, bs = stage._fold(blk.shortcut)
                sc = _linear(x, T.get(blk.shortcut, "plain"), bs, scale=ss)
            else:
                sc = x
            last = bi == nb - 1
            w3 = T.get(blk.conv3, "plain")
            # (last block: 7x7 segments behind the general-grid ones, each followed by ITS last convolution)
            order = sorted(live, key=lambda it: self.seg_wino(it[1], c2)) if last else live
            for si, seg in order:
                sl = slice(seg.row0, seg.row0 + seg.rows)
                early = False
                if self.seg_wino(seg, c2):
                    u2 = T.get(c2, "wino")
                    early = (on_range_final is not None and last and blk.shortcut is None
                             and isinstance(u2, ops.SplitWeight) and isinstance(w3, ops.SplitWeight))
                    keep = None
                    if _KEEP_V and isinstance(u2, ops.SplitWeight) and c2.in_channels % 8 == 0 and seg.n > 0:
                        # (the whole workspace of this convolution -- V and M, 121 * n * (Cin + N) * 4 bytes: 0.4 GB per block at 800
                        # proposals -- stays alive until the backward although only V is read again: the kernels take ONE
                        # pointer; trivial against 288 GB.  Only when the forward runs in ONE pass: with the developer knob
                        # LOCOV_WINO_CHUNK the workspace holds V per chunk as [121][chunk][Cin], which the weight gradient's
                        # TN GEMMs would misread as [121][R][Cin] -- then the activation is transformed again in the backward)
                        ws_bytes = ops.winograd_workspace_bytes(seg.n, c2.in_channels, c2.out_channels)
                        if ws_bytes == 121 * seg.n * (c2.in_channels + c2.out_channels) * 4 + 16:
                            keep = self.wino_ws[(bi, si)] = torch.empty(ws_bytes, dtype=torch.uint8, device=self.device)
                    ops.winograd_conv3x3(Y1[sl], u2, scale=s2, shift=b2, relu=True, roi_major=True, in_roi_major=True, out=Y2[sl],
                                         range_check_scale=16.0 if early else None, workspace=keep)
                    if early:
                        on_range_final()
                        on_range_final = None
                else:
                    # general grid (the whole-grid call): 3x3 as a GEMM over im2col patches -- K = 9 Cin columns in the order of
                    # the packed weight -- so that it runs in the stage's arithmetic; the patches are kept for the weight gradient
                    col = self.cols[(bi, si)] = ops.im2col3x3(Y1[sl], seg.H, seg.W)
                    _linear(col, T.get(c2, "col"), b2, scale=s2, relu=True, out=Y2[sl])
                if last and len(live) > 1:
                    _linear(Y2[sl], w3, b3, scale=s3, residual=sc[sl.start - lo:sl.stop - lo], relu=True, out=OUT[sl])
            if not (last and len(live) > 1):
                _linear(Y2[joint], w3, b3, scale=s3, residual=sc, relu=True, out=OUT[joint])
            x = OUT[joint]
        return [seg for _, seg in new]

    def seg_wino(self, seg: Segment, c2) -> bool:
        return seg.wino and _wino_ok(seg.H, seg.W, c2.in_channels, c2.out_channels)

    def outputs(self, inputs: Sequence[torch.Tensor], pooled: Sequence[bool]) -> List[torch.Tensor]:
        """The segments' stage outputs as differentiable tensors ([rows, Cout] pixel rows, or with pooled[i] the per-tile mean
        [n, Cout]) of `inputs` (the tensors the producers wrote into input_rows, carrying the graph) and the stage's weights.
        The graph is TWO autograd nodes per bottleneck (Res5BlockFn "head": conv1 + shortcut, "tail": conv2 + conv3) behind an output
        node (Res5OutputFn): a half's weight gradients reach their AccumulateGrad nodes -- and DistributedDataParallel's bucket
        hooks -- as soon as that half's backward kernels are enqueued, while everything in front of it is still to come
        (ovr/engine/trainer.py:61-66: the reference gets the same overlap from Detectron2's per-convolution autograd nodes)."""
        assert len(inputs) == len(pooled) == len(self.segments) and not self._pending
        # a DEFERRED guard held by the caller (the ROI heads' training forward): nothing reads it before the backward runs, so
        # the backward must not turn the inf / NaN activations of an out-of-range forward into gradients -- every node zero-fills
        # what it produced on the device when that word is set (the caller does the same with this forward's outputs)
        # (its per-forward copy `step_word`, which the caller fills at the end of ITS forward: a later forward's labelling read
        # may clear the guard's own word before this backward runs)
        active = ops.active_guard(self.device) if self.split else None
        self.skip_words = [getattr(active, "step_word", active.word)] if active is not None and getattr(active, "deferred", False) else []
        self.bwd_split = self.split and not _BWD_F32
        self._carry = None                            # ("tail" | "head", gradient, its operand-scale slot[, ...]) on its way to the next node
        self._prefetched = False
        self.need_x0 = any(t.requires_grad for t in inputs)
        h = None
        for bi, blk in enumerate(self.stage):
            head_w = (blk.conv1.weight,) + ((blk.shortcut.weight,) if blk.shortcut is not None else ())
            y1 = Res5BlockFn.apply(self, bi, "head", len(inputs), *inputs, *head_w) if bi == 0 else Res5BlockFn.apply(self, bi, "head", 1, h, *head_w)
            h = Res5BlockFn.apply(self, bi, "tail", 1, y1, blk.conv2.weight, blk.conv3.weight)
        outs = Res5OutputFn.apply(self, tuple(bool(p) for p in pooled), h)
        return list(outs) if isinstance(outs, tuple) else [outs]

    # -- pieces of the backward shared by the nodes ---------------------------------------------------------------------------
    def _bwd_operands(self):
        # (LOCOV_RES5_BWD_F32: a split forward with the backward on the f32 MFMA takes the fp32 operand set)
        sp = self.bwd_split
        return self.operands if self.operands.split == bool(sp) else self.stage.train_operands(bool(sp))

    def _bwd_guarded(self, fn):
        """Run one node's backward.  Split arithmetic: what can leave fp16's range here is not data -- the activations passed the
        forward's guard (the weight-gradient GEMMs and the Winograd transforms see the same tensors at the same scales) and every
        gradient's operand scale is chosen on the device from its own max |g|.  Only a REMEMBERED weight scale (Res5Stage._split:
        chosen again every 64 steps, 8x headroom) that stopped covering a weight could trip the guard -- the pack kernel raises
        it.  That is recorded in the stage's "bwd" guard and nothing is read inside autograd (a host read here would stall DDP's
        overlapped all-reduce).  Instead every node ends with a GradScaler-style skip decided ON THE DEVICE: when the word is set
        every gradient the node produced is zero-filled (locov_zero_if_raised: one launch that exits after a scalar load when
        the word is clear), so no inf / NaN can reach the optimizer or the all-reduce whether or not another step follows.  The
        operands of ALL blocks are packed by the first node that runs (_prefetch_bwd_operands), so a stale scale of any block is
        known before the first weight gradient is handed to autograd: a skipped pass leaves zeros everywhere, never the
        gradients of some blocks.  The ROI heads look at the word together with the ONE host read of the next step's labelling
        (SampleAllROIHeads.label_and_sample_proposals), warn, and drop the remembered scales; `stage.backward_guard_raised()`
        reads it on demand (e.g. next to a trainer's loss logging)."""
        if not self.bwd_split:
            grads = fn()
            for w in self.skip_words:
                ops.zero_if_raised(grads, w)
            return grads
        guard = self.stage.range_guard("bwd", self.device)
        with ops.range_guard(guard):
            self._prefetch_bwd_operands()
            grads = fn()
        ops.zero_if_raised(grads, guard.word)          # (every gradient here is a freshly written contiguous tensor or a row slice of one)
        for w in self.skip_words:
            ops.zero_if_raised(grads, w)
        return grads

    def _prefetch_bwd_operands(self) -> None:
        """Split arithmetic, operands not out of the step's one preparation launch (the steps that choose scales): pack every
        operand the backward of ANY block will ask for now, under the guard, in front of the first gradient."""
        if self._prefetched:
            return
        self._prefetched = True
        T = self._bwd_operands()
        if not T.split:
            return
        live = [seg for seg in self.segments if seg.rows > 0]
        for bi, blk in enumerate(self.stage):
            c2 = blk.conv2
            want = [(blk.conv3, "t")]
            if bi > 0 or self.need_x0:                # (block 0's input gradient only when the stage input asks for one)
                want.append((blk.conv1, "t"))
                if blk.shortcut is not None:
                    want.append((blk.shortcut, "t"))
            for seg in live:
                want.append((c2, "uflip" if _wino_ok(seg.H, seg.W, c2.out_channels, c2.in_channels) and not _NO_WINO_BWD else "flip9"))
            for conv, tag in want:
                if (id(conv), tag) not in T.ready:
                    T.get(conv, tag)


def _linear(x, W, bias=None, **kw):
    """One 1x1 convolution / im2col GEMM: split-operand f16 MFMA for an ops.SplitWeight, the f32 MFMA for a plain tensor."""
    if isinstance(W, ops.SplitWeight):
        return ops.linear_split(x, W, bias, **kw)
    return ops.linear(x, W, bias, **kw)


# test / measurement hook: called as _BACKWARD_MARK(kind, block) with kind "backward_begin", "block_begin", "block_mid", "block_end" from inside the
# backward (host side, kernels of everything before it are enqueued) -- tests/test_gpu_multirank.py and bench.py record HIP events
# here to place DDP's bucket-ready points on the Res5 backward's timeline
_BACKWARD_MARK = None


def _mark(kind: str, bi: int = -1) -> None:
    if _BACKWARD_MARK is not None:
        _BACKWARD_MARK(kind, bi)


class Res5OutputFn(torch.autograd.Function):
    """The tail of a Res5Step's graph: hands out the segments' outputs (rows, or their per-tile mean) of the last block's joint
    rows; backward gathers the segments' output gradients into ONE matrix, masked by the last block's ReLU, and chooses its
    operand scale (split arithmetic) from the small tensors it was derived from."""

    @staticmethod
    def forward(ctx, step, pooled, h):
        ctx.step, ctx.pooled = step, pooled
        last = step.act[-1][2]
        ctx.save_for_backward(last[:step.filled])
        outs = []
        for seg, pool in zip(step.segments, pooled):
            o = last[seg.row0:seg.row0 + seg.rows]
            if pool:
                o = ops.spatial_mean(o.view(seg.n, seg.H, seg.W, o.shape[1]), channels_last=1) if seg.rows else o.new_zeros((0, o.shape[1]))
            outs.append(o)
        return tuple(outs)

    @staticmethod
    def backward(ctx, *grad_outs):
        step = ctx.step
        _mark("backward_begin")
        (out_last,) = ctx.saved_tensors
        rows = step.filled
        sp = step.bwd_split
        g = torch.empty((rows, out_last.shape[1]), dtype=torch.float32, device=step.device)
        if rows == 0:
            step._carry = ("tail", g, None)
            return None, None, g
        # gradient of the last block's output, masked by its ReLU, every segment into its rows of ONE matrix
        # (the element-wise kernels at the head of the chain keep the separate reduction: their waves all finish together,
        # so every one of them would issue its atomic -- measured +110 us on the grid's relu_mask against a 10 us reduction)
        bound_of, bound_mul = [], []
        for seg, pool, go in zip(step.segments, ctx.pooled, grad_outs):
            if seg.rows == 0:
                continue
            sl = slice(seg.row0, seg.row0 + seg.rows)
            if go is None:
                g[sl].zero_()
                continue
            go = ops._dev(go, "grad_out")
            if pool:
                ops.spatial_mean_bwd(go, out_last[sl], seg.H * seg.W, out=g[sl])
            else:
                ops.relu_mask(go, out_last[sl], out=g[sl])
            bound_of.append(go)
            bound_mul.append(1.0 / (seg.H * seg.W) if pool else 1.0)
        # the operand scale of g: its range is bounded by the SMALL tensors it was just derived from (a broadcast of the pooled
        # gradient / 49, a masked copy of the grid's) -- one launch over those instead of a pass over g's 43 400 x 2 048 values
        if not sp:
            sg = None
        elif bound_of and len(bound_of) <= ops.AMAX_BOUND_MAX and all(t.numel() % 4 == 0 for t in bound_of):
            sg = ops.amax_bound(bound_of, bound_mul)
        else:
            sg = ops.split_scale_from_amax(g)
        for w in step.skip_words:
            ops.zero_if_raised([g], w)
        step._carry = ("tail", g, sg)
        return None, None, g


class Res5BlockFn(torch.autograd.Function):
    """One HALF of a bottleneck of a Res5Step under autograd -- "head": conv1 (+ the projection shortcut), "tail": conv2 + conv3 --
    forward hands out what the step already computed (the half's output rows: y1 / the block's output); backward is the half's
    part of the joint pass

        tail, with g the gradient of the block's output already masked by (output > 0), over the rows of ALL segments:
          dW3 = s3 * g^T y2                      TN GEMM over the pixel rows (gemm_tn.hip)
          g2  = (g . s3 W3) * [y2 > 0]           NT GEMM, mask fused into the epilogue
          dW2 = s2 * wgrad3x3(y1, g2)            per segment: Winograd domain (121 TN GEMMs) or im2col + TN GEMM
          g1  = conv3x3(g2, flip(s2 W2)) * [y1 > 0]         per segment
        head:
          dW1 = s1 * g1^T x ;  dWs = ss * g^T x
          gx  = (g1 . s1 W1 + g [. ss Ws]) * [x > 0]      = the masked gradient of the previous block's output

    Two nodes per block because autograd hands a node's weight gradients on when the node RETURNS: conv3's and conv2's leave half a
    block earlier than they would with the block as one node (block 0's 13.6 MB: while its conv1 / shortcut gradients are still
    being formed, instead of behind the stage's last kernel).
    Inputs: (step, block index, half, number of data inputs, the data inputs -- head of block 0: x0 of every segment; head of any
    other block: the previous block's joint output rows; tail: the head's y1 --, the half's convolution weights: conv1[, shortcut] /
    conv2, conv3, passed as inputs so that autograd routes their gradients).  The tensors between two nodes are INTERNAL to
    Res5Step.outputs: the gradient one node returns for its input is the next node's operand -- already masked by the ReLU in
    front of it -- and travels with its operand-scale slot (and, from a tail to its head, with the block's output gradient, which
    the shortcut path needs) in step._carry.
    saved_tensors: head (x, y1), tail (y1, y2, out) over the joint rows (tests read the active sets through saved_activations)."""

    @staticmethod
    def forward(ctx, step, bi, half, n_in, *args):
        rows = step.filled
        ctx.step, ctx.bi, ctx.half, ctx.n_in = step, bi, half, n_in
        ctx.nw = len(args) - n_in
        y1, y2, out = step.act[bi]
        if half == "head":
            x = step.x0 if bi == 0 else step.act[bi - 1][2]
            ctx.save_for_backward(x[:rows], y1[:rows])
            return y1[:rows]
        ctx.save_for_backward(y1[:rows], y2[:rows], out[:rows])
        return out[:rows]

    @staticmethod
    def backward(ctx, g):
        step = ctx.step
        if ctx.half == "tail":
            _mark("block_begin", ctx.bi)
        fn = Res5BlockFn._backward_tail if ctx.half == "tail" else Res5BlockFn._backward_head
        grads = step._bwd_guarded(lambda: fn(ctx, g))
        _mark("block_mid" if ctx.half == "tail" else "block_end", ctx.bi)
        return (None, None, None, None) + tuple(grads)

    @staticmethod
    def _take_carry(step, g, what):
        carry, step._carry = step._carry, None
        if carry is None or carry[0] != what or carry[1].data_ptr() != g.data_ptr() or carry[1].shape != g.shape:
            raise RuntimeError("Res5BlockFn.backward: the gradient of a node's output must come from the node behind it "
                               "(the tensors between two Res5 nodes are internal to Res5Step.outputs)")
        return carry[1:]

    @staticmethod
    def _helpers(ctx):
        step = ctx.step
        sp = step.bwd_split
        T = step._bwd_operands()
        # Operand scales of the gradients (split arithmetic): every kernel that WRITES a gradient folds max |.| into a zeroed
        # 16-byte slot (ops.scale_slot) on its way out, and the GEMMs that read the gradient derive its power-of-two scale from
        # the slot -- no separate pass over the tensor, no host read.
        slot = (lambda ref: ops.scale_slot(ref)) if sp else (lambda ref: None)

        def wgrad_1x1(g_, sg_, x_, s_):                # dW = s * g^T x
            if sp and g_.shape[1] % 4 == 0 and x_.shape[1] % 4 == 0 and g_.shape[0] > 0:
                return ops.gemm_tn_split(g_, x_, s_, sg_, 16.0)
            return ops.gemm_tn(g_, x_, s_)

        def dgrad_1x1(g_, sg_, conv, tag="t", amax_out=None, wt=None, **kw):   # (g . wt^T [+ residual]) [mask]; amax_out: slot of the result
            if wt is None:
                wt = T.get(conv, tag)
            if isinstance(wt, ops.SplitWeight):
                return ops.linear_split_ex(g_, wt, x_scale_dev=sg_, amax_out=amax_out, **kw), amax_out
            y_ = ops.linear_ex(g_, wt, **kw)
            return y_, (ops.split_scale_from_amax(y_) if sp and amax_out is not None else None)

        return sp, T, slot, wgrad_1x1, dgrad_1x1

    @staticmethod
    def _backward_tail(ctx, g):
        """conv3 and conv2 of block bi: (dW2, dW3) and g1, the masked gradient of y1."""
        step, bi = ctx.step, ctx.bi
        stage, segs = step.stage, step.segments
        rows = step.filled
        blk = stage[bi]
        need_w = ctx.needs_input_grad[5:]
        gw: List[Optional[torch.Tensor]] = [None, None]
        if rows == 0:
            g1 = g.new_empty((0, blk.conv1.out_channels))
            step._carry = ("head", g1, None, g, None)
            return (g1,) + tuple(torch.zeros_like(w) if need else None for w, need in zip((blk.conv2.weight, blk.conv3.weight), need_w))
        g, sg = Res5BlockFn._take_carry(step, g, "tail")
        sp, T, slot, wgrad_1x1, dgrad_1x1 = Res5BlockFn._helpers(ctx)
        y1, y2, _ = ctx.saved_tensors
        s2, s3 = stage._fold(blk.conv2)[0], stage._fold(blk.conv3)[0]
        c2 = blk.conv2
        # conv3: dW3 = s3 * g^T y2 ; g2 = (g . s3 W3) [y2 > 0]         -- all segments, one launch each
        if need_w[1]:
            gw[1] = wgrad_1x1(g, sg, y2, s3).view_as(blk.conv3.weight)
        g2, sg2 = dgrad_1x1(g, sg, blk.conv3, amax_out=slot(g), mask=y2)
        # conv2 (3x3), per segment: dW2 = s2 * wgrad(y1, g2) ; g1 = conv3x3(g2, flip(s2 W2)) [y1 > 0]
        g1 = torch.empty((rows, y1.shape[1]), dtype=torch.float32, device=step.device)
        sg1 = slot(g2)
        # (on the f32 MFMA nothing fills sg1; in split arithmetic every segment's kernel folds its max into the ONE slot)
        # The segments' 3x3 gradients are independent of each other: with two of them (whole grid + proposals) the general-grid
        # segment's chain -- im2col GEMMs of 4 200 rows that fill half the chip -- runs on a SIDE stream next to the proposals'
        # Winograd-domain chain (between two joins with the main stream per block)
        live = [(si, seg) for si, seg in enumerate(segs) if seg.rows > 0]
        side = _side_stream(step.device) if _BWD_STREAMS and len(live) == 2 and sum(step.seg_wino(sg_, c2) for _, sg_ in live) == 1 else None
        main = torch.cuda.current_stream(step.device)
        # the flipped 3x3 filters of BOTH forms, taken here on the main stream, in front of the fork: on the on-demand chain (first
        # step, the f32 MFMA, a step behind forget_scales) the Winograd-domain form is derived from the flipped weight the im2col
        # form derives too -- built inside the side stream's block it would be read by the main stream's segment without an order
        # between the two (found by tests/test_gpu_multirank.py: garbage gradients from block 2's conv1 on under DDP's timing)
        flipped = {}
        for _, seg in live:
            tag = "uflip" if _wino_ok(seg.H, seg.W, c2.out_channels, c2.in_channels) and not _NO_WINO_BWD else "flip9"
            if tag not in flipped:
                flipped[tag] = T.get(c2, tag)
        parts = []
        for si, seg in live:
            sl = slice(seg.row0, seg.row0 + seg.rows)
            wino = step.seg_wino(seg, c2)
            on_side = side is not None and not wino
            if on_side:
                side.wait_stream(main)
                for t_ in (g1, g2, y1):
                    t_.record_stream(side)
            with torch.cuda.stream(side if on_side else main):
                part = None
                if need_w[0]:
                    if wino and c2.out_channels % 4 == 0 and not _NO_WINO_BWD:
                        part = ops.winograd_wgrad(y1[sl], g2[sl], s2, roi_major=True, split=sp, v_split=step.wino_ws.get((bi, si)) if sp else None)
                    else:
                        colw = step.cols.get((bi, si))
                        if colw is None:
                            colw = ops.im2col3x3(y1[sl], seg.H, seg.W)
                        elif on_side:
                            colw.record_stream(side)
                        if sp:
                            part = ops.conv3x3_wgrad_unpack(ops.gemm_tn_split(g2[sl], colw, None, sg2, 16.0), s2)
                        else:
                            part = ops.conv3x3_wgrad_unpack(ops.gemm_tn(g2[sl], colw), s2)
                    if on_side:
                        part.record_stream(main)
                    parts.append(part)
                if _wino_ok(seg.H, seg.W, c2.out_channels, c2.in_channels) and not _NO_WINO_BWD:
                    uflip = flipped["uflip"]
                    if isinstance(uflip, ops.SplitWeight):
                        ops.winograd_conv3x3_split_ex(g2[sl], uflip, mask=y1[sl], roi_major=True, amax_out=sg1, out=g1[sl])
                    else:
                        ops.winograd_conv3x3_ex(g2[sl], uflip, mask=y1[sl], roi_major=True, out=g1[sl])
                        if sp and g1[sl].numel() % 4 == 0:
                            ops.amax_bound([g1[sl]], [1.0], slot=sg1)
                else:
                    # data gradient on the general grid: the same im2col GEMM with the flipped filter (im2col only copies and
                    # zero-pads: the patches have g2's range)
                    _, got = dgrad_1x1(ops.im2col3x3(g2[sl], seg.H, seg.W), sg2, c2, "flip9", amax_out=sg1, wt=flipped["flip9"],
                                       mask=y1[sl], out=g1[sl])
                    if sp and got is not sg1 and g1[sl].numel() % 4 == 0:
                        ops.amax_bound([g1[sl]], [1.0], slot=sg1)      # (an operand the split GEMM cannot take: its f32 result's range by a pass)
        if side is not None:
            main.wait_stream(side)
        if need_w[0]:
            dw2 = parts[0]
            for part in parts[1:]:
                dw2 = dw2.add_(part)
            gw[0] = dw2
        step._carry = ("head", g1, sg1, g, sg)
        return (g1,) + tuple(gw)

    @staticmethod
    def _backward_head(ctx, g1):
        """conv1 (+ shortcut) of block bi: (dW1[, dWs]) and gx, the masked gradient of the previous block's output."""
        step, bi, n_in = ctx.step, ctx.bi, ctx.n_in
        stage, segs = step.stage, step.segments
        rows = step.filled
        blk = stage[bi]
        has_sc = blk.shortcut is not None
        first = bi == 0
        need_x = any(ctx.needs_input_grad[4:4 + n_in])
        need_w = ctx.needs_input_grad[4 + n_in:]
        gw: List[Optional[torch.Tensor]] = [None] * ctx.nw
        weights = (blk.conv1.weight,) + ((blk.shortcut.weight,) if has_sc else ())
        if rows == 0:
            step._carry = None
            if not first:
                gx = torch.empty((0, blk.conv1.in_channels), dtype=torch.float32, device=step.device)
                step._carry = ("tail", gx, None)
                return (gx,) + tuple(torch.zeros_like(w) if need else None for w, need in zip(weights, need_w))
            return tuple(None for _ in range(n_in)) + tuple(torch.zeros_like(w) if need else None for w, need in zip(weights, need_w))
        g1, sg1, g, sg = Res5BlockFn._take_carry(step, g1, "head")
        sp, T, slot, wgrad_1x1, dgrad_1x1 = Res5BlockFn._helpers(ctx)
        x, _ = ctx.saved_tensors
        s1 = stage._fold(blk.conv1)[0]
        # conv1 (+ shortcut): dW1 = s1 * g1^T x ; gx = (g1 . s1 W1 + shortcut path) [x > 0]
        if need_w[0]:
            gw[0] = wgrad_1x1(g1, sg1, x, s1).view_as(blk.conv1.weight)
        if has_sc:
            ss = stage._fold(blk.shortcut)[0]
            if need_w[1]:
                gw[1] = wgrad_1x1(g, sg, x, ss).view_as(blk.shortcut.weight)
        if not need_x:
            return tuple(None for _ in range(n_in)) + tuple(gw)
        # the input of block 0 is the stage input (no ReLU in front of it); every other block's input is the
        # post-ReLU output of its predecessor, whose mask turns gx into that block's masked output gradient
        mask = None if first else x
        # (the slot of gx is only needed when a block below reads it: not for the stage input's gradient)
        gx, sgx = dgrad_1x1(g1, sg1, blk.conv1, amax_out=None if (first or has_sc) else slot(g1),
                            residual=None if has_sc else g, mask=None if has_sc else mask)
        if has_sc:
            gx, sgx = dgrad_1x1(g, sg, blk.shortcut, amax_out=None if first else slot(g), residual=gx, mask=mask)
        if not first:
            step._carry = ("tail", gx, sgx)
            return (gx,) + tuple(gw)
        gxs = [None] * n_in
        for i, seg in enumerate(segs):
            if ctx.needs_input_grad[4 + i]:
                gxs[i] = gx[seg.row0:seg.row0 + seg.rows]
        return tuple(gxs) + tuple(gw)


def saved_activations(out: torch.Tensor) -> List[torch.Tensor]:
    """The activations the Res5 nodes behind `out` (an output of Res5Step.outputs, possibly behind view / layout nodes) saved
    for their backward, per block (x, y1, y2, out) over the joint rows, block 0 first -- the device's own active sets."""
    found, stack, seen = {}, [out.grad_fn], set()
    while stack:
        n = stack.pop()
        if n is None or id(n) in seen:
            continue
        seen.add(id(n))
        if "Res5BlockFn" in type(n).__name__:
            found[(n.bi, n.half)] = n
        stack += [f for f, _ in n.next_functions]
    saved = []
    for bi in sorted({k[0] for k in found}):
        x, y1 = found[(bi, "head")].saved_tensors
        _, y2, o = found[(bi, "tail")].saved_tensors
        saved += [x, y1, y2, o]
    return saved


def _guarded(stage, split, guard, device, build):
    """build(split) -> a finished Res5Step.  With `guard` ((on_overflow,)) and no caller-held range guard: check the split
    arithmetic's range-guard word behind the forward and rebuild the step on the f32 MFMA when an activation left fp16's
    range (calls on_overflow() first).  Inside a caller's own `ops.range_guard` block the launches raise THAT guard and
    nothing is read here: the caller checks once for all the calls it made."""
    own = None
    if split and guard is not None and ops.active_guard(device) is None:
        own = stage.range_guard("fwd", device)
        own.reset()
    with ops.range_guard(own):
        step = build(split)
    if own is not None and own.raised():
        # (what left the range may be a remembered operand scale that stopped covering its weight -- the step's operands are
        # packed under this guard: the scales are chosen afresh at the next packing)
        stage.forget_scales()
        if guard[0] is not None:
            guard[0]()
        del step
        step = build(False)
    return step


def res5_rows(stage, x0: torch.Tensor, R: int, H: int, W: int, pooled: bool = False, split: bool = True,
              overflow_check: bool = True, on_overflow=None) -> torch.Tensor:
    """Differentiable Res5 on ROI-major pixel rows: out = Res5(x0), x0 [R*H*W, Cin] the stage input already sub-sampled by
    block 0's stride (the even positions: STRIDE_IN_1X1) -> the rows [R*H*W, Cout], or with `pooled` their per-ROI mean
    [R, Cout].  A one-segment Res5Step."""
    assert stage.supports_rows_path(), "the rows path needs FrozenBN, STRIDE_IN_1X1 and ungrouped convolutions"
    xd = ops._dev(x0.detach(), "x0")

    def build(sp):
        step = Res5Step(stage, sp, xd.device, R * H * W, grid=not (H == 7 and W == 7), rois=H == 7 and W == 7)
        step.input_rows(R * H * W).copy_(xd)
        step.forward(R, H, W)
        return step
    step = _guarded(stage, split, (on_overflow,) if overflow_check else None, xd.device, build)
    return step.outputs([x0], [pooled])[0]


class _ToNHWC(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        return ops.nchw_to_nhwc(x.detach())

    @staticmethod
    def backward(ctx, g):
        return ops.nhwc_to_nchw(g)


class _ToNCHW(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        return ops.nhwc_to_nchw(x.detach())

    @staticmethod
    def backward(ctx, g):
        return ops.nchw_to_nhwc(g)


def to_nhwc(x: torch.Tensor) -> torch.Tensor:
    """[N,C,H,W] -> [N,H,W,C], differentiable."""
    return _ToNHWC.apply(x)


class _RoiAlignEvenRows(torch.autograd.Function):
    """Even-grid ROIAlign of a channels-last map -> ROI-major rows [oh*ow*R, C] (oh = P/2), differentiable in the map.
    step: a Res5Step whose next segment's input rows the result is written into (no copy)."""

    @staticmethod
    def forward(ctx, nhwc, rois, P, scale, sampling_ratio, aligned, step=None):
        o = (P + 1) // 2
        dst = step.input_rows(o * o * rois.shape[0]) if step is not None else None
        out = ops.roi_align_nhwc(nhwc.detach(), rois, P, scale, sampling_ratio, aligned, bin_stride=2, pos_major=False, out=dst)
        ctx.save_for_backward(rois)
        ctx.args = (tuple(nhwc.shape), P, scale, sampling_ratio, aligned)
        return out.view(-1, nhwc.shape[3])

    @staticmethod
    def backward(ctx, g):
        (rois,) = ctx.saved_tensors
        shape, P, scale, sampling_ratio, aligned = ctx.args
        return ops.roi_align_nhwc_bwd(g, shape, rois, P, scale, sampling_ratio, aligned, bin_stride=2, pos_major=False), \
            None, None, None, None, None, None


def roi_align_even_rows(nhwc, rois, P, scale, sampling_ratio, aligned, step=None) -> torch.Tensor:
    return _RoiAlignEvenRows.apply(nhwc, rois, int(P), float(scale), int(sampling_ratio), bool(aligned), step)


class _Stride2Rows(torch.autograd.Function):
    @staticmethod
    def forward(ctx, nhwc, step=None):
        N, H, W, _ = nhwc.shape
        ctx.dims = (N, H, W)
        dst = step.input_rows(N * ((H + 1) // 2) * ((W + 1) // 2)) if step is not None else None
        return ops.rows_stride2(nhwc.detach(), N, H, W, True, out=dst)

    @staticmethod
    def backward(ctx, g):
        N, H, W = ctx.dims
        return ops.rows_stride2(g, N, H, W, False), None


def grid_rows(nhwc: torch.Tensor, step=None) -> torch.Tensor:
    """The even pixels of a channels-last map as pixel rows [N * ceil(H/2) * ceil(W/2), C] (block 0's stride-2 1x1 convolutions
    read nothing else), differentiable; with `step` written into that Res5Step's next input rows."""
    return _Stride2Rows.apply(nhwc, step)


def grid_segment(step: "Res5Step", nhwc: torch.Tensor) -> torch.Tensor:
    """Enqueue the whole-grid call of roi_emb_heads.py:323 as the next segment of `step`; returns its input rows (graph)."""
    N, H, W, _ = nhwc.shape
    rows = grid_rows(nhwc, step)
    step.forward(N, (H + 1) // 2, (W + 1) // 2)
    return rows


def roi_segment(step: "Res5Step", nhwc: torch.Tensor, rois: torch.Tensor, P: int, scale: float, sampling_ratio: int, aligned: bool,
                on_range_final=None) -> torch.Tensor:
    """Enqueue the proposals' call of roi_emb_heads.py:343 (even-grid ROIAlign + the stage) as the next segment of `step`."""
    x0 = roi_align_even_rows(nhwc, rois, P, scale, sampling_ratio, aligned, step)
    o = (int(P) + 1) // 2
    step.forward(rois.shape[0], o, o, on_range_final=on_range_final)
    return x0


def grid_and_roi_segments(step: "Res5Step", nhwc: torch.Tensor, rois: torch.Tensor, P: int, scale: float, sampling_ratio: int, aligned: bool,
                          on_range_final=None):
    """Both calls of roi_emb_heads.py:323 and :343 as the two segments of `step`, forwarded TOGETHER (their 1x1 convolutions
    share launches): possible when the sampled proposals are known without a host wait.  Returns (grid input rows, proposals'
    input rows) carrying the graph."""
    N, H, W, _ = nhwc.shape
    rows = grid_rows(nhwc, step)
    x0 = roi_align_even_rows(nhwc, rois, P, scale, sampling_ratio, aligned, step)
    o = (int(P) + 1) // 2
    step.forward_segments([(N, (H + 1) // 2, (W + 1) // 2), (rois.shape[0], o, o)], on_range_final=on_range_final)
    return rows, x0


def grid_capacity(nhwc: torch.Tensor) -> int:
    N, H, W, _ = nhwc.shape
    return N * ((H + 1) // 2) * ((W + 1) // 2)


def to_nchw(rows: torch.Tensor, N: int, OH: int, OW: int) -> torch.Tensor:
    """Pixel rows [N*OH*OW, C] -> logical NCHW [N, C, OH, OW], differentiable."""
    return _ToNCHW.apply(rows.view(N, OH, OW, rows.shape[1]))


def res5_grid(stage, nhwc: torch.Tensor, split: bool = True, overflow_check: bool = True, on_overflow=None) -> torch.Tensor:
    """roi_emb_heads.py:323 -- the stage applied to the whole channels-last res4 map [N,H,W,Cin] -> logical NCHW
    [N, Cout, ceil(H/2), ceil(W/2)], differentiable in the map and the convolution weights.  Block 0's stride-2 1x1
    convolutions read the even pixels; the 3x3 convolutions run as implicit GEMMs over the (H/2 x W/2) grid."""
    N, H, W, _ = nhwc.shape
    assert stage.supports_rows_path() and stage[0].stride == 2 and stage[0].stride_in_1x1
    OH, OW = (H + 1) // 2, (W + 1) // 2
    made = {}

    def build(sp):
        step = Res5Step(stage, sp, nhwc.device, N * OH * OW, rois=False)
        made["rows"] = grid_segment(step, nhwc)
        return step
    step = _guarded(stage, split, (on_overflow,) if overflow_check else None, nhwc.device, build)
    return to_nchw(step.outputs([made["rows"]], [False])[0], N, OH, OW)


def res5_rois(stage, nhwc: torch.Tensor, rois: torch.Tensor, P: int, scale: float, sampling_ratio: int, aligned: bool,
              pooled: bool = False, split: bool = True, overflow_check: bool = True, on_overflow=None) -> torch.Tensor:
    """roi_emb_heads.py:243-245 under autograd: even-grid ROIAlign of the channels-last map + the stage on the proposals' 7x7
    tiles -> rows [o*o*R, Cout] (o = P/2), or with `pooled` their per-proposal mean [R, Cout]."""
    assert stage.supports_rows_path() and stage[0].stride == 2 and stage[0].stride_in_1x1
    o = (int(P) + 1) // 2
    made = {}

    def build(sp):
        step = Res5Step(stage, sp, nhwc.device, o * o * rois.shape[0], grid=False, rois=o == 7)
        made["x0"] = roi_segment(step, nhwc, rois, P, scale, sampling_ratio, aligned)
        return step
    step = _guarded(stage, split, (on_overflow,) if overflow_check else None, nhwc.device, build)
    return step.outputs([made["x0"]], [pooled])[0]
