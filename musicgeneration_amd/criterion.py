"""Mirror of mg/model/MusicTransformer/criterion.py: SmoothCrossEntropyLoss + CustomSchedule."""
from __future__ import annotations

import torch
from torch.nn.modules.loss import _Loss

from . import ops


def _rows_with_stride(x: torch.Tensor) -> torch.Tensor:
    """The model returns logits as a [B,L,V] view of a [B,L,Vp] buffer (vocabulary rows padded to the GEMM
    tile).  The kernels take a row stride, so hand them the padded base tensor instead of copying."""
    b = x._base
    if (b is not None and b.is_contiguous() and b.dim() == x.dim() and b.shape[:-1] == x.shape[:-1]
            and b.data_ptr() == x.data_ptr() and x.stride(-1) == 1 and b.dtype == x.dtype):
        return b
    return x.contiguous()


class SmoothCrossEntropyLoss(_Loss):
    """criterion.py:28-67.  ``forward(input [.., V] logits, target [..])`` -> scalar mean over
    non-``ignore_index`` targets of -sum q' log softmax, q' = (1-eps) onehot + eps/V.
    Runs the fused libmgx kernel (closed form, no one-hot tensor); logits are consumed in bf16."""
    __constants__ = ['label_smoothing', 'vocab_size', 'ignore_index', 'reduction']

    def __init__(self, label_smoothing, vocab_size, ignore_index=-100, reduction='mean', is_logits=True):
        assert 0.0 <= label_smoothing <= 1.0
        super().__init__(reduction=reduction)
        self.label_smoothing = label_smoothing
        self.vocab_size = vocab_size
        self.ignore_index = ignore_index
        self.input_is_logits = is_logits

    def fused(self, input, target):
        """-> (loss, stats f32[4] = [loss_sum, n_nonpad, n_correct, n_rows], argmax int32[rows])"""
        if input.dtype != torch.bfloat16:
            input = input.to(torch.bfloat16)
        return ops.smooth_ce(_rows_with_stride(input), target.to(torch.int32).contiguous(), self.vocab_size,
                             self.label_smoothing, self.ignore_index)

    def forward(self, input, target):
        loss, stats, _ = self.fused(input, target)
        if self.reduction == 'mean':
            return loss
        elif self.reduction == 'sum':
            return loss * stats[1]
        raise NotImplementedError


class CustomSchedule:
    """Noam learning-rate schedule that also drives the optimiser (criterion.py:70-96):
    ``lr(n) = d_model^-0.5 * min(n^-0.5, n * warmup^-1.5)`` with n = number of ``step()`` calls so far."""

    def __init__(self, d_model, warmup_steps=4000, optimizer=None):
        self.d_model, self.warmup_steps, self.optimizer = d_model, warmup_steps, optimizer
        self._step = 0
        self._rate = 0

    def rate(self, step=None):
        n = self._step if step is None else step
        decay = n ** (-0.5)
        ramp = n * (self.warmup_steps ** -1.5)
        return self.d_model ** (-0.5) * min(decay, ramp)

    def step(self):
        """advance the schedule, set every parameter group's lr, then take the optimiser step"""
        self._step += 1
        self._rate = lr = self.rate()
        for group in self.optimizer.param_groups:
            group['lr'] = lr
        self.optimizer.step()
