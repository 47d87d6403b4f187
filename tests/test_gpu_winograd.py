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
