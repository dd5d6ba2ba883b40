"""CrossEntropyLoss -- mirrors loss/cross_entropy_loss.py:5-18 of the reference: a thin wrapper over
`nn.CrossEntropyLoss` applied to the (already soft-maxed) network output with `target.squeeze(1).long()`.
Not named by the hot path: it stays a stock torch op on the device."""
import torch
import torch.nn as nn


class CrossEntropyLoss(nn.Module):

    def __init__(self, weight=None, size_average=None, ignore_index=-100, reduce=None, reduction='mean'):
        super(CrossEntropyLoss, self).__init__()
        self.func = nn.CrossEntropyLoss(weight=weight, ignore_index=ignore_index, reduction=reduction)

    def forward(self, input, target):
        assert isinstance(input, torch.Tensor)
        assert isinstance(target, torch.Tensor)
        target = torch.squeeze(target, dim=1).long()
        return self.func(input, target)
