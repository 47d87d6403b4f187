"""locov_amd.distill_losses against the reference's own MultiDistillLoss* outputs
(tests/golden/g5_distill_losses.npz, produced by tests/golden/make_golden.py from
ovr/modeling/meta_arch/distill_mmss_gcnn.py:211-433)."""
import os

import numpy as np
import pytest
import torch

from locov_amd import distill_losses as dl

G5 = np.load(os.path.join(os.path.dirname(__file__), "golden", "g5_distill_losses.npz"))


@pytest.mark.parametrize("name", ["MultiDistillLoss", "MultiDistillLossJS", "MultiDistillLossL2"])
@pytest.mark.parametrize("tt", [True, False])
def test_matches_reference_values(name, tt):
    for c in range(int(G5["num_cases"])):
        trans, w2r, r2w = (torch.from_numpy(G5[f"c{c}_{k}"]) for k in ("trans", "w2r", "r2w"))
        mod = getattr(dl, name)(float(G5[f"c{c}_temp"]), loss_weight=0.7, detach_teacher=True, transformer_teacher=tt)
        got = float(mod(trans, w2r, r2w))
        want = float(G5[f"c{c}_{name}_tt{int(tt)}"])
        assert abs(got - want) <= 2e-6 * max(1.0, abs(want)), (c, got, want)


@pytest.mark.parametrize("cls", [dl.MultiDistillLoss, dl.MultiDistillLossJS, dl.MultiDistillLossL2])
def test_teacher_detach_controls_the_gradient_path(cls):
    g = torch.Generator().manual_seed(3)
    for tt in (True, False):
        trans, w2r, r2w = (torch.randn(4, 4, generator=g, requires_grad=True) for _ in range(3))
        cls(2.0, detach_teacher=True, transformer_teacher=tt)(trans, w2r, r2w).backward()
        teacher_grads = [trans.grad] if tt else [w2r.grad, r2w.grad]
        student_grads = [w2r.grad, r2w.grad] if tt else [trans.grad]
        assert all(x is None or float(x.abs().sum()) == 0.0 for x in teacher_grads)
        assert all(x is not None and float(x.abs().sum()) > 0.0 for x in student_grads)
        # without the detach every input receives a gradient
        trans, w2r, r2w = (torch.randn(4, 4, generator=g, requires_grad=True) for _ in range(3))
        cls(2.0, detach_teacher=False, transformer_teacher=tt)(trans, w2r, r2w).backward()
        assert all(float(x.grad.abs().sum()) > 0.0 for x in (trans, w2r, r2w))


def test_identical_costs_give_zero_divergence():
    c = torch.randn(5, 5, generator=torch.Generator().manual_seed(1))
    for cls in (dl.MultiDistillLoss, dl.MultiDistillLossL2):
        assert abs(float(cls(1.5)(c, c.clone(), c.clone()))) < 1e-6
