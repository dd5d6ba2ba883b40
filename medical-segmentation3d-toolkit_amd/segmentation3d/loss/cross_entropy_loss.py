"""CrossEntropyLoss plugin (reference: loss/cross_entropy_loss.py:5-18; selected by `loss.name = 'CE'`,
core/seg_train.py:99).

The reference feeds the network's soft-max OUTPUT to `nn.CrossEntropyLoss`, i.e. the loss applies a second log-softmax
to probabilities; that quirk is kept because checkpoints trained with it depend on it.  The label map arrives as
float [N, 1, D, H, W] and is squeezed to integer [N, D, H, W].  This loss is not on the accelerated path (SURVEY.md
section 8a row 13): it runs as stock torch ops on the device.
"""
import torch
import torch.nn as nn


class CrossEntropyLoss(nn.Module):

    def __init__(self, weight=None, size_average=None, ignore_index=-100, reduce=None, reduction='mean'):
        super(CrossEntropyLoss, self).__init__()
        # size_average / reduce are accepted for signature compatibility only (deprecated in torch)
        self.func = nn.CrossEntropyLoss(weight=weight, ignore_index=ignore_index, reduction=reduction)

    def forward(self, input, target):
        if not (isinstance(input, torch.Tensor) and isinstance(target, torch.Tensor)):
            raise TypeError('CrossEntropyLoss expects tensors')
        labels = target.squeeze(1).long()
        return self.func(input, labels)
