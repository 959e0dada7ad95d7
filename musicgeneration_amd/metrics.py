"""Mirror of mg/model/MusicTransformer/metrics.py (Accuracy, CategoricalAccuracy, LogitsBucketting,
MetricsSet).  The reference runs softmax+argmax, the loss and another argmax as three passes over
the logits; here ``MetricsSet`` recognises that trio and serves all three from ONE fused kernel
launch (loss + accuracy + argmax), while each metric also works stand-alone."""
from __future__ import annotations

from typing import Dict

import torch

from . import ops
from .criterion import SmoothCrossEntropyLoss, _rows_with_stride


class _Metric(torch.nn.Module):
    def forward(self, input: torch.Tensor, target: torch.Tensor):
        raise NotImplementedError()


class Accuracy(_Metric):
    def forward(self, input: torch.Tensor, target: torch.Tensor):
        """input [B,L] predicted ids, target [B,L] -> mean over ALL positions (metrics.py:22-29)"""
        bool_acc = input.long() == target.long()
        return bool_acc.sum().to(torch.float) / bool_acc.numel()


class MockAccuracy(Accuracy):
    pass


def _fused_stats(input, target):
    V = input.shape[-1]
    x = input if input.dtype == torch.bfloat16 else input.to(torch.bfloat16)
    return ops.smooth_ce_fwd(_rows_with_stride(x), target.to(torch.int32).contiguous(), V, 0.0, -1)


class CategoricalAccuracy(Accuracy):
    def forward(self, input: torch.Tensor, target: torch.Tensor):
        """input [B,T,V] logits; argmax is softmax-invariant (metrics.py:40-52)"""
        stats, _, _ = _fused_stats(input, target)
        return stats[2] / stats[3]


class LogitsBucketting(_Metric):
    def __init__(self, vocab_size):
        super().__init__()

    def forward(self, input: torch.Tensor, target: torch.Tensor):
        _, argmax, _ = _fused_stats(input, target)
        return argmax


class MetricsSet(object):
    def __init__(self, metric_dict: Dict):
        super().__init__()
        self.metrics = metric_dict

    def __call__(self, input: torch.Tensor, target: torch.Tensor):
        return self.forward(input=input, target=target)

    def forward(self, input: torch.Tensor, target: torch.Tensor):
        losses = [k for k, m in self.metrics.items() if isinstance(m, SmoothCrossEntropyLoss)]
        if len(losses) == 1 and self.metrics[losses[0]].reduction == 'mean':
            loss, stats, argmax = self.metrics[losses[0]].fused(input, target)
            self.last_nonpad = stats[1]          # device scalar: non-pad targets of this batch (dp.loss_weight)
            out = {}
            for k, m in self.metrics.items():
                if k == losses[0]:
                    out[k] = loss
                elif type(m) is CategoricalAccuracy:
                    out[k] = stats[2] / stats[3]
                elif type(m) is LogitsBucketting:
                    out[k] = argmax
                else:
                    out[k] = m(input, target)
            return out
        self.last_nonpad = None
        return {k: metric(input, target) for k, metric in self.metrics.items()}
