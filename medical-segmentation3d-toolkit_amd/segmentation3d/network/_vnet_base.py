"""Shared skeleton of the two network plugins.

Both reference plugins (network/vnet.py:20-51, network/vbnet.py:21-53) are the same encoder/decoder: a 16-channel
stem, four stride-2 stages that double the width, four transposed stages that take a skip connection, and a class
head.  They differ only in which stages use bottleneck residual blocks.  The stage table below is the single source
of truth; attribute names match the reference so checkpoints keep their keys.
"""
import torch
import torch.nn as nn

from segmentation3d import _ops

from segmentation3d.network.module.vnet_inblock import InputBlock
from segmentation3d.network.module.vnet_outblock import OutputBlock
from segmentation3d.network.module.vnet_upblock import UpBlock
from segmentation3d.network.module.vnet_downblock import DownBlock

STEM_WIDTH = 16
# (attribute, input width, number of residual convs)
ENCODER_STAGES = (('down_32', 16, 1), ('down_64', 32, 2), ('down_128', 64, 3), ('down_256', 128, 3))
# (attribute, input width, output width incl. skip, number of residual convs, encoder feature used as skip)
DECODER_STAGES = (('up_256', 256, 256, 3, 'down_128'), ('up_128', 256, 128, 3, 'down_64'),
                  ('up_64', 128, 64, 2, 'down_32'), ('up_32', 64, 32, 1, 'in_block'))
MAX_STRIDE = 2 ** len(ENCODER_STAGES)
PREFILL_SKIP_SLOTS = True   # inference: encoder features written straight into the decoder's concatenated buffers (forward())


class VNetBase(nn.Module):
    """volumetric segmentation network; `bottleneck` lists the stage names that use BottResidualBlock3"""

    def __init__(self, in_channels, out_channels, bottleneck=()):
        super(VNetBase, self).__init__()
        self.in_block = InputBlock(in_channels, STEM_WIDTH)
        for name, width, convs in ENCODER_STAGES:
            setattr(self, name, DownBlock(width, convs, compression=name in bottleneck))
        for name, cin, cout, convs, _ in DECODER_STAGES:
            setattr(self, name, UpBlock(cin, cout, convs, compression=name in bottleneck))
        self.out_block = OutputBlock(DECODER_STAGES[-1][2], out_channels)

    def forward(self, input):
        if not isinstance(input, torch.Tensor):
            raise TypeError('input must be a torch.Tensor')
        if input.dim() != 5 or any(int(s) % MAX_STRIDE for s in input.shape[2:]):
            raise ValueError('input must be [N, C, D, H, W] with D, H, W divisible by {} (got {})'.format(
                MAX_STRIDE, tuple(input.shape)))
        # Inference (no autograd, fp32 activations): every encoder feature that a decoder stage concatenates is written by its
        # producer straight into the second half of that stage's concatenated buffer (vnet_upblock.py:21 without the copy:
        # 4 x 27 us per 4-patch forward, 2.7 % of a whole-volume job); its other consumer, the next DownBlock's stride-2
        # conv, reads it there with a row stride (seg3d_conv3d_k2s2_mfma_fwd_ld).
        slots = {}
        if PREFILL_SKIP_SLOTS and not torch.is_grad_enabled() and input.is_cuda and _ops.activation_dtype_name() == 'fp32':
            N, _, D, H, W = input.shape
            widths = {'in_block': STEM_WIDTH}
            widths.update({name: 2 * width for name, width, _ in ENCODER_STAGES})
            scale = {'in_block': 1}
            for k, (name, _, _) in enumerate(ENCODER_STAGES):
                scale[name] = 2 ** (k + 1)
            for _, _, _, _, skip in DECODER_STAGES:
                C, f = widths[skip], scale[skip]
                if C % 4 == 0:
                    slots[skip] = torch.empty((N, D // f, H // f, W // f, 2 * C), dtype=torch.float32, device=input.device)

        def slot_of(name):
            buf = slots.get(name)
            return None if buf is None else buf[..., buf.shape[4] // 2:]
        feats = {'in_block': self.in_block(input, out_slot=slot_of('in_block'))}
        x, source = feats['in_block'], 'in_block'
        # every encoder feature but the deepest has two consumers: the next DownBlock and a decoder skip.  The link lets the
        # DownBlock's data-gradient kernel add the skip gradient instead of autograd summing two full tensors.
        links = {}
        for name, _, _ in ENCODER_STAGES:
            links[source] = _ops.ResidualLink() if (x.requires_grad and torch.is_grad_enabled()) else None
            x = getattr(self, name)(x, skip_link=links[source], out_slot=slot_of(name))
            feats[name], source = x, name
        for name, _, _, _, skip in DECODER_STAGES:
            x = getattr(self, name)(x, feats[skip], skip_link=links.get(skip), cat_buf=slots.get(skip))
        return self.out_block(x)

    def max_stride(self):
        return MAX_STRIDE


def init_parameters(net, initializer):
    """apply a per-module initialiser (kaiming / gaussian) to every sub-module, as `net.apply` does"""
    net.apply(initializer)
    return net
