"""BinaryDiceLoss -- drop-in for loss/binary_dice_loss.py:5-36 of the reference.

input [N, 2, ...] probabilities, target [N, 1, ...] (used as floats).  The reference takes (value, index) = max over the
two channels and multiplies value by index, i.e. keeps channel 1 where it is STRICTLY larger than channel 0 (a tie takes
index 0) and 0 elsewhere, then computes 1 - (2 sum(pred * t) + 1e-6) / (sum(pred^2) + sum(t^2) + 1e-6) per sample and
the batch mean; the gradient reaches channel 1 only, and only where it won.

Runs on the HIP engine (seg3d_binary_dice_fwd / _bwd, csrc/loss.hip): one fused reduction pass, one elementwise backward.
MultiDiceLoss does not route through this class: its kernel folds the constant 1/C channel in as a threshold.
"""
import torch.nn as nn

from segmentation3d import _ops


class BinaryDiceLoss(nn.Module):
    """ Dice Loss for binary segmentation """

    def forward(self, input, target):
        if input.dim() < 2 or input.shape[1] != 2:
            raise ValueError('BinaryDiceLoss expects a 2-channel input, got {}'.format(tuple(input.shape)))
        return _ops.BinaryDiceLossFunction.apply(input, target)
