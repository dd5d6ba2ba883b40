"""ORACLE (test infrastructure, not product code) -- CPU restatement of the reference's hot path.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.  The product package
(medical-segmentation3d-toolkit_amd/segmentation3d) never does; it has no CPU path at all.

What is restated: the reference's networks and losses are ~300 lines of Python composing stock torch.nn operators
(Conv3d / ConvTranspose3d / GroupNorm / ReLU / Softmax; README pins torch 1.3, semantics unchanged in 2.10).  This file
re-expresses the same operator sequence functionally over a state_dict, citing the reference lines it follows, and
the closed forms of the losses.  Pinning: tests/test_oracle_golden.py checks every function here against fixtures
produced by importing the real reference modules in the build container (oracle/gen_golden.py -> tests/golden/).
The reference itself ships no golden vectors for this path (only shape asserts), so the
fixtures generated from its own code are the pin.

All citations are relative to /root/reference/segmentation3d.
"""
import torch
import torch.nn.functional as F

GN_EPS = 1e-5

VNET_ENCODER = (('down_32', 16, 1), ('down_64', 32, 2), ('down_128', 64, 3), ('down_256', 128, 3))  # network/vnet.py:26-29
VNET_DECODER = (('up_256', 256, 256, 3, 'down_128'), ('up_128', 256, 128, 3, 'down_64'),           # network/vnet.py:30-33
                ('up_64', 128, 64, 2, 'down_32'), ('up_32', 64, 32, 1, 'in_block'))
VBNET_BOTTLENECK = ('down_64', 'down_128', 'down_256', 'up_256', 'up_128')                          # network/vbnet.py:27-32


def _gn(x, sd, prefix):
    # nn.GroupNorm(1, C): network/module/conv_gn_relu3.py:11
    return F.group_norm(x, 1, sd[prefix + '.weight'], sd[prefix + '.bias'], GN_EPS)


def conv_gn_relu3(x, sd, prefix, do_act=True):
    """ConvGnRelu3.forward: act?(gn(conv(x))), k3 s1 p1 -- network/module/conv_gn_relu3.py:16-20"""
    y = F.conv3d(x, sd[prefix + '.conv.weight'], sd.get(prefix + '.conv.bias'), stride=1, padding=1)
    y = _gn(y, sd, prefix + '.gn')
    return F.relu(y) if do_act else y


def bott_conv_gn_relu3(x, sd, prefix, do_act=True):
    """BottConvGnRelu3.forward: conv3(conv2(conv1(x))) -- network/module/conv_gn_relu3.py:32-34"""
    y = conv_gn_relu3(x, sd, prefix + '.conv1', True)
    y = conv_gn_relu3(y, sd, prefix + '.conv2', True)
    return conv_gn_relu3(y, sd, prefix + '.conv3', do_act)


def residual_block(x, sd, prefix, num_convs, bottleneck):
    """ResidualBlock3 / BottResidualBlock3: act(input + ops(input)), last op without activation --
    network/module/residual_block3.py:21-26, 44-46"""
    unit = bott_conv_gn_relu3 if bottleneck else conv_gn_relu3
    y = x
    for i in range(num_convs):
        y = unit(y, sd, '{}.ops.{}'.format(prefix, i), do_act=(i != num_convs - 1))
    return F.relu(x + y)


def input_block(x, sd, prefix='in_block'):
    """network/module/vnet_inblock.py:13-15"""
    y = F.conv3d(x, sd[prefix + '.conv.weight'], sd[prefix + '.conv.bias'], padding=1)
    return F.relu(_gn(y, sd, prefix + '.gn'))


def down_block(x, sd, prefix, num_convs, bottleneck):
    """network/module/vnet_downblock.py:19-22"""
    y = F.conv3d(x, sd[prefix + '.down_conv.weight'], sd[prefix + '.down_conv.bias'], stride=2)
    y = F.relu(_gn(y, sd, prefix + '.down_gn'))
    return residual_block(y, sd, prefix + '.rblock', num_convs, bottleneck)


def up_block(x, skip, sd, prefix, num_convs, bottleneck):
    """network/module/vnet_upblock.py:19-23 (cat order: up first, then skip)"""
    y = F.conv_transpose3d(x, sd[prefix + '.up_conv.weight'], sd[prefix + '.up_conv.bias'], stride=2)
    y = F.relu(_gn(y, sd, prefix + '.up_gn'))
    y = torch.cat((y, skip), 1)
    return residual_block(y, sd, prefix + '.rblock', num_convs, bottleneck)


def output_block(x, sd, prefix='out_block'):
    """network/module/vnet_outblock.py:20-24"""
    y = F.conv3d(x, sd[prefix + '.conv1.weight'], sd[prefix + '.conv1.bias'], padding=1)
    y = F.relu(_gn(y, sd, prefix + '.gn1'))
    y = F.conv3d(y, sd[prefix + '.conv2.weight'], sd[prefix + '.conv2.bias'])
    y = _gn(y, sd, prefix + '.gn2')
    return F.softmax(y, dim=1)


def segmentation_net(x, sd, name='vnet'):
    """SegmentationNet.forward -- network/vnet.py:36-48 / network/vbnet.py:36-50"""
    bott = VBNET_BOTTLENECK if name == 'vbnet' else ()
    feats = {'in_block': input_block(x, sd)}
    y = feats['in_block']
    for stage, _, convs in VNET_ENCODER:
        y = down_block(y, sd, stage, convs, stage in bott)
        feats[stage] = y
    for stage, _, _, convs, skip in VNET_DECODER:
        y = up_block(y, feats[skip], sd, stage, convs, stage in bott)
    return output_block(y, sd)


# ---- losses ---------------------------------------------------------------------------------------------------------
def multi_dice_loss(probs, target, weights):
    """loss/multi_dice_loss.py:24-43 + loss/binary_dice_loss.py:9-36 in closed form:
    sum_c w_c * mean_n [1 - (2 sum(ph t_c) + eps) / (sum(ph^2) + sum(t_c^2) + eps)], ph = p_c * [p_c > 1/C]"""
    C = probs.shape[1]
    w = torch.as_tensor(weights, dtype=torch.float32)
    w = w / w.sum()                                                   # multi_dice_loss.py:18-19
    n = probs.shape[0]
    total = 0
    thr = (1.0 / C + torch.zeros(1, dtype=probs.dtype))               # multi_dice_loss.py:36 (float32 constant)
    for c in range(C):
        p = probs[:, c].reshape(n, -1)
        ph = p * (p > thr).to(p.dtype)                                # binary_dice_loss.py:13-14 (ties -> index 0 -> 0)
        t = (target == c).float().reshape(n, -1)                      # multi_dice_loss.py:37
        inter = (ph * t).sum(1)
        area = (ph * ph).sum(1) + (t * t).sum(1)
        eps = torch.tensor(1e-6)
        loss_c = (torch.tensor(1.0) - (torch.tensor(2.0) * inter + eps) / (area + eps)).mean()  # binary_dice_loss.py:33-34
        total = total + loss_c * w[c]
    return total


def focal_loss(probs, target, class_num, alpha=None, gamma=2, size_average=True):
    """loss/focal_loss.py:27-61"""
    if alpha is None:
        a = torch.ones(class_num) / class_num                         # focal_loss.py:11
    else:
        a = torch.as_tensor(alpha, dtype=torch.float32)
        a = a / a.sum()                                               # focal_loss.py:14-16
    if probs.dim() > 2:
        p = probs.movedim(1, -1).reshape(-1, class_num)               # focal_loss.py:33-38
    else:
        p = probs
    t = target.long().reshape(-1)
    pt = p.gather(1, t[:, None])[:, 0] + 1e-10                        # focal_loss.py:46-48
    logp = pt.log()
    at = a.to(p.dtype)[t]
    if gamma > 0:
        batch = -at * torch.pow(1 - pt, gamma) * logp                 # focal_loss.py:51-52
    else:
        batch = -at * logp
    return batch.mean() if size_average else batch.sum()              # focal_loss.py:56-59


def binary_dice_loss(probs, target):
    """loss/binary_dice_loss.py:9-36 on a 2-channel input: (value, index) = max over channels, value *= index, i.e.
    pred = p1 where p1 > p0 STRICTLY (a tie takes index 0) and 0 elsewhere"""
    n = probs.shape[0]
    pred = (probs[:, 1] * (probs[:, 1] > probs[:, 0]).to(probs.dtype)).reshape(n, -1)      # :13-14
    t = target.float().reshape(n, -1)
    inter = (pred * t).sum(1)
    area = (pred * pred).sum(1) + (t * t).sum(1)
    eps = torch.tensor(1e-6)
    return (torch.tensor(1.0) - (torch.tensor(2.0) * inter + eps) / (area + eps)).mean()   # :33-34


def cross_entropy_on_probs(probs, target):
    """loss/cross_entropy_loss.py:13-18: nn.CrossEntropyLoss applied to the network's soft-max OUTPUT (a second
    log-softmax over probabilities -- the reference's quirk), labels = squeeze(target, 1).long()"""
    return F.cross_entropy(probs, target.squeeze(1).long())


# ---- train step -------------------------------------------------------------------------------------------------------
def train_step(sd_params, opt, x, target, name, loss_name, loss_kwargs):
    """core/seg_train.py:119-127: zero_grad -> forward -> loss -> backward -> Adam step; returns the loss value"""
    opt.zero_grad()
    probs = segmentation_net(x, sd_params, name)
    if loss_name == 'Dice':
        loss = multi_dice_loss(probs, target, **loss_kwargs)
    else:
        loss = focal_loss(probs, target, **loss_kwargs)
    loss.backward()
    opt.step()
    return float(loss.detach()), probs.detach()
