"""MultiDiceLoss -- drop-in for loss/multi_dice_loss.py:6-43 of the reference.

Reference algorithm: for every class i (background included) build the 2-channel tensor [1/C, p_i], binarise it
with BinaryDiceLoss's arg-max gate (=> p_hat = p_i * [p_i > 1/C], ties -> 0), compute
1 - (2 sum(p_hat t_i) + 1e-6) / (sum(p_hat^2) + sum(t_i) + 1e-6) per sample, average over the batch and sum with
normalised class weights.  Here all 3*N*C spatial sums come from ONE pass over the probabilities
(seg3d_dice_fwd) instead of ~4*C full-size temporaries, and the backward is one elementwise kernel.
"""
import torch
import torch.nn as nn

from segmentation3d import _ops


class MultiDiceLoss(nn.Module):
    """ Dice Loss for multi-class segmentation """

    def __init__(self, weights, num_class, use_gpu):
        """
        :param weights: weight for each class dice loss
        :param num_class: the number of class
        :param use_gpu: kept for signature compatibility; the weights follow the input's device
        """
        super(MultiDiceLoss, self).__init__()
        self.num_class = num_class
        assert len(weights) == self.num_class, "the length of weight must equal to num_class"
        self.weights = torch.FloatTensor(weights)
        self.weights = self.weights / self.weights.sum()
        self.use_gpu = use_gpu

    def forward(self, input_tensor, target):
        """
        :param input_tensor: network output probabilities [N, C, D, H, W] (any trailing spatial shape)
        :param target: ground truth class ids, float, [N, 1, D, H, W]
        :return: weighted dice loss (0-dim tensor, differentiable w.r.t. input_tensor)
        """
        if input_tensor.shape[1] != self.num_class:
            raise ValueError('input has {} channels but num_class is {}'.format(input_tensor.shape[1], self.num_class))
        if self.weights.device != input_tensor.device:
            self.weights = self.weights.to(input_tensor.device)
        return _ops.DiceLossFunction.apply(input_tensor, target, self.weights)
