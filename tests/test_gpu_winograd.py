"""Winograd-domain 3x3 convolution (locov_winograd_conv3x3_f32) and the batched NT GEMM under it.
Reference semantics: BottleneckBlock.conv2 + FrozenBN + ReLU of the Res5 stage
(roi_emb_heads.py:217-241, applied :245,:323); the oracle is the direct convolution."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ops():
    if not torch.cuda.is_available():
        pytest.fail("GPU tests need a ROCm device")
    from locov_amd import _lib, ops
    _lib.load()
    return ops


@pytest.mark.parametrize("B,M,N,K", [(3, 70, 48, 64), (5, 300, 256, 96), (121, 130, 128, 32)])
def test_batched_gemm(ops, B, M, N, K):
    g = torch.Generator().manual_seed(B * 1000 + M)
    x = torch.randn(B, M, K, generator=g)
    w = torch.randn(B, N, K, generator=g)
    got = ops.gemm_nt_batched(x.cuda(), w.cuda()).cpu()
    want = torch.bmm(x.double(), w.double().transpose(1, 2))
    assert (got.double() - want).abs().max().item() < 1e-4


def _rows(x):          # [R,C,7,7] -> position-major rows [49*R, C]
    R, C = x.shape[:2]
    return x.permute(2, 3, 0, 1).reshape(49 * R, C).contiguous()


def _unrows(y, R):     # [49*R, N] -> [R,N,7,7]
    return y.reshape(7, 7, R, -1).permute(2, 3, 0, 1).contiguous()


@pytest.mark.parametrize("R,Cin,N,relu", [(1, 32, 4, False), (37, 64, 48, True), (300, 128, 256, True), (129, 512, 512, True)])
def test_winograd_conv_matches_direct(ops, R, Cin, N, relu):
    g = torch.Generator().manual_seed(R)
    x = torch.randn(R, Cin, 7, 7, generator=g)
    w = torch.randn(N, Cin, 3, 3, generator=g) * (2.0 / (9 * Cin)) ** 0.5
    scale = 0.5 + torch.rand(N, generator=g)
    shift = torch.randn(N, generator=g) * 0.1
    want = F.conv2d(x.double(), w.double(), padding=1) * scale.double().view(1, -1, 1, 1) + shift.double().view(1, -1, 1, 1)
    if relu:
        want = want.relu()
    U = ops.winograd_pack_weight(w.cuda())
    got = _unrows(ops.winograd_conv3x3(_rows(x).cuda(), U, scale=scale.cuda(), shift=shift.cuda(), relu=relu).cpu(), R)
    err = (got.double() - want).abs().max().item()
    assert err < 2e-5 * max(1.0, want.abs().max().item()), err
    # and against the direct HIP form on the same rows
    direct = ops.conv3x3_nhwc(_rows(x).cuda(), ops.pack_conv3x3_weight(w.cuda()), 7, 7, scale=scale.cuda(),
                              shift=shift.cuda(), relu=relu, pos_major=True).cpu()
    assert (_unrows(direct, R) - got).abs().max().item() < 2e-5 * max(1.0, want.abs().max().item())


def test_winograd_pack_weight_is_fp64_transform(ops):
    import sys, os
    sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tools"))
    import gen_winograd_tables as T
    G = torch.tensor([[float(v) for v in r] for r in T.build()[1]], dtype=torch.float64)
    w = torch.randn(8, 32, 3, 3, generator=torch.Generator().manual_seed(0))
    want = torch.einsum("ai,kcij,bj->abkc", G, w.double(), G).reshape(121, 8, 32).float()
    got = ops.winograd_pack_weight(w.cuda()).cpu()
    np.testing.assert_allclose(got.numpy(), want.numpy(), rtol=2e-7, atol=1e-9)


def test_winograd_empty(ops):
    U = ops.winograd_pack_weight(torch.randn(8, 32, 3, 3).cuda())
    assert ops.winograd_conv3x3(torch.empty(0, 32).cuda(), U).shape == (0, 8)


def test_row_strided_operands_and_outputs(ops):
    """The K-concatenation of Res5 block 0 (res5.Res5Stage.rows_input): ROIAlign and the Winograd conv
    write column blocks of a wider matrix, the GEMM reads a column block as its A operand."""
    g = torch.Generator().manual_seed(11)
    R, C, N = 19, 64, 32
    feat = torch.randn(2, 20, 24, C, generator=g).cuda()
    rois = torch.tensor([[i % 2, 8.0 + i, 6.0 + 2 * i, 150.0 + 5 * i, 120.0 + 3 * i] for i in range(R)]).cuda()
    want = ops.roi_align_nhwc(feat, rois, 14, 1.0 / 16, 0, True, bin_stride=2, pos_major=True).view(49 * R, C)
    buf = torch.full((49 * R, N + C), 7.0, device="cuda")
    got = ops.roi_align_nhwc(feat, rois, 14, 1.0 / 16, 0, True, bin_stride=2, pos_major=True, out=buf[:, N:])
    assert got.data_ptr() == buf[:, N:].data_ptr() and torch.equal(buf[:, N:], want) and bool((buf[:, :N] == 7.0).all())
    w = torch.randn(N, C, 3, 3, generator=g).cuda() * 0.05
    U = ops.winograd_pack_weight(w)
    y_want = ops.winograd_conv3x3(want, U, relu=True)
    ops.winograd_conv3x3(buf[:, N:], U, relu=True, out=buf[:, :N])          # strided input AND output
    assert torch.equal(buf[:, :N], y_want) and torch.equal(buf[:, N:], want)
    wl = torch.randn(40, C, generator=g).cuda()
    assert torch.equal(ops.linear(buf[:, N:], wl), ops.linear(want.contiguous(), wl))


def test_block0_k_concatenation_matches_separate_gemms(ops, oracle):
    from locov_amd.config import get_cfg
    from locov_amd.res5 import build_res5_block
    cfg = get_cfg()
    cfg.MODEL.RESNETS.RES2_OUT_CHANNELS = 32
    cfg.MODEL.RESNETS.WIDTH_PER_GROUP = 8
    res5, out_ch = build_res5_block(cfg)
    res5.load_state_dict(oracle.make_res5_params(3, in_ch=128, mid=64, out_ch=256))
    res5 = res5.cuda().eval()
    x = torch.randn(49 * 23, 128, generator=torch.Generator().manual_seed(5)).cuda()
    plain = res5.forward_rows(x, 7, 7, pos_major=True, winograd=True)
    x0 = res5.rows_input(49 * 23, x.device)
    assert x0.stride(0) == 64 + 128
    x0.copy_(x)
    fused = res5.forward_rows(x0, 7, 7, pos_major=True, winograd=True)
    # same arithmetic up to where the FrozenBN scales are applied (weights vs epilogue): rounding-level agreement
    assert (fused - plain).abs().max().item() <= 1e-5 * plain.abs().max().item()


def test_concurrent_streams_do_not_share_the_workspace(ops):
    """Two Winograd convolutions enqueued on two HIP streams may run concurrently: each stream owns its
    transform-domain workspace (ops._WINO_WS is keyed by device and stream)."""
    g = torch.Generator().manual_seed(2)
    xs = [torch.randn(49 * 600, 128, generator=g).cuda() for _ in range(2)]
    U = ops.winograd_pack_weight((torch.randn(128, 128, 3, 3, generator=g) * 0.05).cuda())
    want = [ops.winograd_conv3x3(x, U, relu=True) for x in xs]
    torch.cuda.synchronize()
    streams = [torch.cuda.Stream(), torch.cuda.Stream()]
    for _ in range(5):
        got = []
        for st, x in zip(streams, xs):
            st.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(st):
                got.append(ops.winograd_conv3x3(x, U, relu=True))
        torch.cuda.synchronize()
        assert torch.equal(got[0], want[0]) and torch.equal(got[1], want[1])


def test_output_transform_past_32_bit_offsets(ops):
    """wino_output_kernel<..., WIDE>: the flat-address instance the launcher picks when a ROI-major result needs byte offsets
    past 2^32 (R * 49 * row pitch * 4).  The result of 8 600 ROIs is written into a column block of a [49 R, 2560] matrix
    (4.3 GB: the pitch of block 0's K-concatenated operand is what makes rows this far apart) and must equal, bit for bit, the
    same call into a compact buffer (32-bit offsets); the other columns of the wide matrix stay untouched."""
    R, Cin, N, pitch = 8600, 32, 64, 2560
    assert R * 49 * pitch * 4 > 2 ** 32 > R * 49 * N * 4
    g = torch.Generator().manual_seed(86)
    x = torch.randn(49 * R, Cin, generator=g).cuda()
    u = ops.winograd_pack_weight((torch.randn(N, Cin, 3, 3, generator=g) * 0.05).cuda())
    s, b = (torch.rand(N, generator=g) + 0.5).cuda(), (torch.randn(N, generator=g) * 0.1).cuda()
    want = ops.winograd_conv3x3(x, u, scale=s, shift=b, relu=True, roi_major=True, in_roi_major=True)
    wide = torch.full((49 * R, pitch), -7.0, device="cuda")
    col0 = 1024
    ops.winograd_conv3x3(x, u, scale=s, shift=b, relu=True, roi_major=True, in_roi_major=True, out=wide[:, col0:col0 + N])
    assert torch.equal(wide[:, col0:col0 + N], want)
    assert bool((wide[:, :col0] == -7.0).all()) and bool((wide[:, col0 + N:] == -7.0).all())
    del wide, want
    # ... and the split-layout result (a dense destination only): 10 800 ROIs x 2 048 output channels = 4.3 GB, against the same
    # convolution of the two halves (32-bit offsets each)
    R, N = 10800, 2048
    assert R * 49 * N * 4 > 2 ** 32 > (R // 2) * 49 * N * 4
    x = torch.randn(49 * R, Cin, generator=g).cuda()
    us = ops.split_pack(ops.winograd_pack_weight((torch.randn(N, Cin, 3, 3, generator=g) * 0.05).cuda()))
    s, b = (torch.rand(N, generator=g) + 0.5).cuda(), (torch.randn(N, generator=g) * 0.1).cuda()
    kw = dict(scale=s, shift=b, relu=True, roi_major=True, in_roi_major=True, out_split_scale=16.0)
    got = ops.winograd_conv3x3(x, us, **kw)
    half = 49 * (R // 2)
    for lo in (0, half):
        want = ops.winograd_conv3x3(x[lo:lo + half], us, **kw)
        assert torch.equal(got[lo:lo + half].view(torch.int32), want.view(torch.int32)), lo
        del want
