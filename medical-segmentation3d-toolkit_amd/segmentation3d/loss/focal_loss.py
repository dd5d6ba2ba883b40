"""FocalLoss -- drop-in for loss/focal_loss.py:5-61 of the reference.

loss = mean_v( -alpha[t_v] * (1 - p_v)^gamma * log(p_v) ),  p_v = input[v, t_v] + 1e-10, alpha normalised to sum 1
(uniform 1/C by default).  The reference permutes the input to [voxels, C], gathers a one-hot mask and alpha rows;
here one kernel reads the target id and the single probability it selects (seg3d_focal_fwd), and the backward writes
the planar gradient directly.
"""
import torch
from torch import nn

from segmentation3d import _ops


class FocalLoss(nn.Module):

    def __init__(self, class_num, alpha=None, gamma=2, size_average=True, use_gpu=True):
        super(FocalLoss, self).__init__()
        if alpha is None:
            self.alpha = torch.ones(class_num, 1) / class_num
        else:
            assert len(alpha) == class_num
            self.alpha = torch.FloatTensor(alpha)
            self.alpha = self.alpha.unsqueeze(1)
            self.alpha = self.alpha / self.alpha.sum()
        self.gamma = gamma
        self.class_num = class_num
        self.size_average = size_average
        self.use_gpu = use_gpu

    def forward(self, input, target):
        # accepted shapes (focal_loss.py:28-38): [sample, class], [batch, class, y, x], [batch, class, z, y, x]
        assert input.dim() == 2 or input.dim() == 4 or input.dim() == 5
        C = self.class_num
        if input.dim() == 2:
            if input.shape[1] != C:
                raise ValueError('input has {} classes, expected {}'.format(input.shape[1], C))
            N, S = 1, input.shape[0]
            sn, sc, ss = 0, 1, C
        else:
            if input.shape[1] != C:
                raise ValueError('input has {} channels, expected {}'.format(input.shape[1], C))
            N = input.shape[0]
            S = input[0, 0].numel()
            sn, sc, ss = C * S, S, 1
        if self.alpha.device != input.device:
            self.alpha = self.alpha.to(input.device)
        alpha = self.alpha.reshape(-1).contiguous()
        return _ops.FocalLossFunction.apply(input, target, alpha, self.gamma, self.size_average, N, C, S, sn, sc, ss)
