"""Distillation losses over the [B,B] caption x image cost matrices.

Same class names, constructor arguments and `forward(trans_pw_cost, pw_cost_w2r, pw_cost_r2w)` as
ovr/modeling/meta_arch/distill_mmss_gcnn.py:211-433 (selected in distill_prop_mmss_gcnn.py:127-149);
the two student matrices are the "w2r" / "r2w" outputs of GroundingHead.forward (grounding_head.py).
B is the per-GPU batch (4 at 8 GPUs), so this is scalar-sized torch arithmetic -- no kernel.

A cost matrix is read two ways: per image over the captions (softmax over dim 0, "cap") and per caption
over the images (softmax over dim 1, transposed, "img").  Every loss is the sum over
{cap, img} x {w2r, r2w} of one divergence between the teacher view and the student view.
"""
from __future__ import annotations

import torch
from torch import nn

__all__ = ["MultiDistillLoss", "MultiDistillLossJS", "MultiDistillLossL2"]


def _views(cost: torch.Tensor, temp: float, log: bool):
    """(cap, img) distributions of -cost/temp; img is transposed so both are [B,B] with the softmax axis first."""
    z = -cost / temp
    f = torch.log_softmax if log else torch.softmax
    return f(z, dim=0), f(z, dim=1).t()


def _kl(log_q: torch.Tensor, p: torch.Tensor) -> torch.Tensor:
    """nn.KLDivLoss(reduction="batchmean")(log_q, p) = sum p*(log p - log_q) / rows (0*log 0 := 0)."""
    return torch.nn.functional.kl_div(log_q, p, reduction="batchmean")


class _DistillBase(nn.Module):
    def __init__(self, temperature, loss_weight=1.0, detach_teacher=False, transformer_teacher=True):
        super().__init__()
        self.temp = temperature
        self.loss_weight = loss_weight
        self.detach_teacher = detach_teacher
        self.transformer_teacher = transformer_teacher

    def _detach(self, trans, w2r, r2w):
        if self.detach_teacher:
            if self.transformer_teacher:
                trans = trans.detach()
            else:
                w2r, r2w = w2r.detach(), r2w.detach()
        return trans, w2r, r2w


class MultiDistillLoss(_DistillBase):
    """KD: T^2 * KL(teacher || student) per view (distill_mmss_gcnn.py:211-290).  With
    transformer_teacher=False the roles swap: the grounding costs teach the transformer's."""

    def forward(self, trans_pw_cost, pw_cost_w2r, pw_cost_r2w):
        trans, w2r, r2w = self._detach(trans_pw_cost, pw_cost_w2r, pw_cost_r2w)
        t2 = self.temp * self.temp
        total = 0.0
        if self.transformer_teacher:
            teacher = _views(trans, self.temp, log=False)
            for student_cost in (w2r, r2w):
                student = _views(student_cost, self.temp, log=True)
                total = total + sum(_kl(s, p) for s, p in zip(student, teacher)) * t2
        else:
            student = _views(trans, self.temp, log=True)
            for teacher_cost in (w2r, r2w):
                teacher = _views(teacher_cost, self.temp, log=False)
                total = total + sum(_kl(s, p) for s, p in zip(student, teacher)) * t2
        return total * self.loss_weight


class MultiDistillLossJS(_DistillBase):
    """Jensen-Shannon form (distill_mmss_gcnn.py:293-376): 1/2 KL(P||M) + 1/2 KL(Q||M), M = (P+Q)/2.
    As in the reference, the per-caption ("img") terms are measured against the per-image ("cap")
    mixtures M (lines 357-366 reuse m_cap_*): kept for identical loss values."""

    def forward(self, trans_pw_cost, pw_cost_w2r, pw_cost_r2w):
        trans, w2r, r2w = self._detach(trans_pw_cost, pw_cost_w2r, pw_cost_r2w)
        t2 = self.temp * self.temp
        p_cap, _ = _views(trans, self.temp, log=False)
        logp = _views(trans, self.temp, log=True)
        total = 0.0
        for cost in (w2r, r2w):
            q_cap, _ = _views(cost, self.temp, log=False)
            logq = _views(cost, self.temp, log=True)
            m_cap = 0.5 * (p_cap + q_cap)
            for lp, lq in zip(logp, logq):                 # cap view, then img view -- both against m_cap
                total = total + 0.5 * _kl(lp, m_cap) * t2 + 0.5 * _kl(lq, m_cap) * t2
        return total * self.loss_weight


class MultiDistillLossL2(_DistillBase):
    """Mean-squared error between the raw cost matrices, counted once per view (distill_mmss_gcnn.py:379-433)."""

    def forward(self, trans_pw_cost, pw_cost_w2r, pw_cost_r2w):
        trans, w2r, r2w = self._detach(trans_pw_cost, pw_cost_w2r, pw_cost_r2w)
        mse = torch.nn.functional.mse_loss
        total = 0.0
        for cost in (w2r, r2w):
            total = total + mse(trans, cost) + mse(trans.t(), cost.t())
        return total * self.loss_weight
