"""Generate tests/golden/* by running the REAL reference code (build container only).

    PYTHONPATH=/root/reference python oracle/gen_golden.py

The reference (qinliuliuqin/Medical-Segmentation3d-Toolkit, mounted read-only at /root/reference) is imported, fed
deterministic inputs/weights from oracle/detgen.py, and only its OUTPUTS are stored (small .npz / .json).  The GPU box
never sees the reference: tests regenerate the same inputs with detgen and compare against these fixtures.
Nothing from the reference's source is copied; `image_partition_by_fixed_size` cannot be imported (its module
imports SimpleITK), so its function body is executed in place via `ast` on a duck-typed image object.
"""
import ast
import json
import os
import sys

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = '/root/reference'
sys.path.insert(0, REPO)
if REF not in sys.path:
    sys.path.insert(0, REF)
from oracle import detgen  # noqa: E402

OUT = os.path.join(REPO, 'tests', 'golden')
torch.set_num_threads(8)


def _load_sd(module, seed):
    shapes = {k: tuple(v.shape) for k, v in module.state_dict().items()}
    sd = detgen.state_dict_like(shapes, seed)
    module.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
    return shapes


def _grads(module):
    return {k: p.grad.detach().numpy().copy() for k, p in module.named_parameters()}


def gen_blocks():
    from segmentation3d.network.module.conv_gn_relu3 import ConvGnRelu3, BottConvGnRelu3
    from segmentation3d.network.module.residual_block3 import ResidualBlock3, BottResidualBlock3
    from segmentation3d.network.module.vnet_inblock import InputBlock
    from segmentation3d.network.module.vnet_downblock import DownBlock
    from segmentation3d.network.module.vnet_upblock import UpBlock
    from segmentation3d.network.module.vnet_outblock import OutputBlock

    cases = {
        # name: (constructor, input shapes)
        'convgnrelu_k3_1_16': (lambda: ConvGnRelu3(1, 16, 3, 1, 1), [(2, 1, 8, 12, 16)]),
        'convgnrelu_k3_16_16_noact': (lambda: ConvGnRelu3(16, 16, 3, 1, 1, do_act=False), [(2, 16, 8, 8, 8)]),
        'convgnrelu_k3_32_32': (lambda: ConvGnRelu3(32, 32, 3, 1, 1), [(1, 32, 8, 8, 16)]),
        'convgnrelu_k2s2_16_32': (lambda: ConvGnRelu3(16, 32, 2, 2, 0), [(2, 16, 8, 8, 8)]),
        'bottconv_32': (lambda: BottConvGnRelu3(32, 32, 3, 1, 1, 4), [(1, 32, 8, 8, 8)]),
        'resblock_16_x2': (lambda: ResidualBlock3(16, 3, 1, 1, 2), [(2, 16, 8, 8, 8)]),
        'bottresblock_32_x2': (lambda: BottResidualBlock3(32, 3, 1, 1, 4, 2), [(1, 32, 8, 8, 8)]),
        'inblock_1': (lambda: InputBlock(1, 16), [(2, 1, 16, 16, 16)]),
        'inblock_4': (lambda: InputBlock(4, 16), [(1, 4, 8, 8, 8)]),
        'downblock_16_x1': (lambda: DownBlock(16, 1), [(2, 16, 8, 8, 8)]),
        'downblock_32_x2_bott': (lambda: DownBlock(32, 2, compression=True), [(1, 32, 8, 8, 8)]),
        'upblock_64_32_x1': (lambda: UpBlock(64, 32, 1), [(1, 64, 4, 4, 4), (1, 16, 8, 8, 8)]),
        'upblock_64_64_x2_bott': (lambda: UpBlock(64, 64, 2, compression=True), [(1, 64, 4, 4, 4), (1, 32, 8, 8, 8)]),
        'outblock_32_2': (lambda: OutputBlock(32, 2), [(1, 32, 8, 8, 8)]),
        'outblock_32_5': (lambda: OutputBlock(32, 5), [(2, 32, 8, 8, 8)]),
    }
    for name, (ctor, in_shapes) in cases.items():
        m = ctor()
        _load_sd(m, seed=11)
        ins = [torch.from_numpy(detgen.normal(12, '{}/in{}'.format(name, i), s)).requires_grad_(True)
               for i, s in enumerate(in_shapes)]
        out = m(*ins)
        gout = torch.from_numpy(detgen.normal(13, name + '/gout', tuple(out.shape)))
        out.backward(gout)
        data = {'out': out.detach().numpy()}
        for i, t in enumerate(ins):
            data['din{}'.format(i)] = t.grad.numpy()
        for k, g in _grads(m).items():
            data['dparam/' + k] = g
        np.savez_compressed(os.path.join(OUT, 'block_{}.npz'.format(name)), **data)
        print('block', name, tuple(out.shape))


def gen_nets():
    from segmentation3d.network import vnet, vbnet
    from segmentation3d.loss.multi_dice_loss import MultiDiceLoss
    from segmentation3d.loss.focal_loss import FocalLoss
    for plugin_name, plugin in (('vnet', vnet), ('vbnet', vbnet)):
        for cin, ncls in ((1, 2), (1, 5), (4, 4)):
            net = plugin.SegmentationNet(cin, ncls)
            _load_sd(net, seed=21)
            tag = '{}_{}_{}'.format(plugin_name, cin, ncls)
            x = torch.from_numpy(detgen.normal(22, tag + '/x', (1, cin, 32, 32, 32)))
            t = torch.from_numpy(detgen.labels(23, tag + '/t', (1, 1, 32, 32, 32), ncls))
            data = {}
            for loss_name in ('dice', 'focal'):
                net.zero_grad()
                probs = net(x)
                if loss_name == 'dice':
                    w = [1.0 + 0.5 * i for i in range(ncls)]
                    loss = MultiDiceLoss(weights=w, num_class=ncls, use_gpu=False)(probs, t)
                else:
                    loss = FocalLoss(class_num=ncls, alpha=None, gamma=2, use_gpu=False)(probs, t)
                loss.backward()
                data['loss_' + loss_name] = np.float32(loss.item())
                g = _grads(net)
                data['gradnorm_' + loss_name] = np.array([np.sqrt((g[k].astype(np.float64) ** 2).sum()) for k in sorted(g)],
                                                         dtype=np.float64)
                data['grad_' + loss_name + '/in_block.conv.weight'] = g['in_block.conv.weight']
                data['grad_' + loss_name + '/out_block.conv2.weight'] = g['out_block.conv2.weight']
                data['grad_' + loss_name + '/up_32.up_conv.weight'] = g['up_32.up_conv.weight']
                data['grad_' + loss_name + '/down_32.down_conv.bias'] = g['down_32.down_conv.bias']
            data['probs'] = probs.detach().numpy()
            data['param_names'] = np.array(sorted(g))
            np.savez_compressed(os.path.join(OUT, 'net_{}.npz'.format(tag)), **data)
            print('net', tag, float(data['loss_dice']), float(data['loss_focal']))


def gen_losses():
    from segmentation3d.loss.multi_dice_loss import MultiDiceLoss
    from segmentation3d.loss.focal_loss import FocalLoss
    cases = []
    for C, shape in ((2, (2, 2, 6, 8, 10)), (5, (2, 5, 4, 8, 8)), (3, (1, 3, 8, 8, 8))):
        logits = detgen.normal(31, 'loss/logits{}'.format(C), shape, std=2.0)
        e = np.exp(logits - logits.max(1, keepdims=True))
        probs = (e / e.sum(1, keepdims=True)).astype(np.float32)
        target = detgen.labels(32, 'loss/t{}'.format(C), (shape[0], 1) + shape[2:], C)
        cases.append(('rand{}'.format(C), C, probs, target))
    # tie / threshold cases: probabilities exactly 1/C, an empty class, an all-background target
    C = 2
    p = np.full((1, 2, 4, 4, 4), 0.5, dtype=np.float32)
    p[0, 0, :2] = 0.75
    p[0, 1, :2] = 0.25
    t = np.zeros((1, 1, 4, 4, 4), dtype=np.float32)
    t[0, 0, 2:] = 1.0
    cases.append(('ties2', 2, p, t))
    C = 4
    p = np.full((2, 4, 4, 4, 4), 0.25, dtype=np.float32)
    p[:, 0, 1] = 0.4
    p[:, 1, 1] = 0.3
    p[:, 2, 1] = 0.2
    p[:, 3, 1] = 0.1
    t = np.zeros((2, 1, 4, 4, 4), dtype=np.float32)     # all background: classes 1..3 empty
    cases.append(('allbg4', 4, p, t))
    for name, C, probs, target in cases:
        data = {'probs': probs, 'target': target}
        for wname, w in (('uniform', [1.0] * C), ('ramp', [1.0 + i for i in range(C)])):
            pt = torch.from_numpy(probs).clone().requires_grad_(True)
            loss = MultiDiceLoss(weights=w, num_class=C, use_gpu=False)(pt, torch.from_numpy(target))
            loss.backward()
            data['dice_{}'.format(wname)] = np.float32(loss.item())
            data['dice_{}_grad'.format(wname)] = pt.grad.numpy()
        for gname, gamma, alpha in (('g2', 2, None), ('g0', 0, None), ('g1p5_alpha', 1.5, [1.0 + i for i in range(C)])):
            pt = torch.from_numpy(probs).clone().requires_grad_(True)
            loss = FocalLoss(class_num=C, alpha=alpha, gamma=gamma, use_gpu=False)(pt, torch.from_numpy(target))
            loss.backward()
            data['focal_{}'.format(gname)] = np.float32(loss.item())
            data['focal_{}_grad'.format(gname)] = pt.grad.numpy()
        # size_average=False and the 2-D [sample, class] form
        p2 = torch.from_numpy(probs).movedim(1, -1).reshape(-1, C).clone().requires_grad_(True)
        loss = FocalLoss(class_num=C, gamma=2, size_average=False, use_gpu=False)(p2, torch.from_numpy(target).reshape(-1))
        loss.backward()
        data['focal_2d_sum'] = np.float32(loss.item())
        data['focal_2d_sum_grad'] = p2.grad.numpy()
        np.savez_compressed(os.path.join(OUT, 'loss_{}.npz'.format(name)), **data)
        print('loss', name, {k: float(v) for k, v in data.items() if np.ndim(v) == 0})


class _DuckImage(object):
    def __init__(self, size, spacing):
        self._size, self._spacing = tuple(size), tuple(spacing)

    def GetSize(self):
        return self._size

    def GetSpacing(self):
        return self._spacing

    def GetOrigin(self):
        return (0.0, 0.0, 0.0)


def gen_partitions():
    src = open(os.path.join(REF, 'segmentation3d', 'utils', 'image_tools.py')).read()
    tree = ast.parse(src)
    fn = [n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name == 'image_partition_by_fixed_size'][0]
    ns = {'np': np}
    exec(compile(ast.Module(body=[fn], type_ignores=[]), 'image_tools.py', 'exec'), ns)  # runs in memory only
    ref_fn = ns['image_partition_by_fixed_size']
    cases = {
        'vol512x512x400_96_48': ((512, 512, 400), (1.0, 1.0, 1.0), None, (96, 96, 96), (48, 48, 48)),
        'vol128_96_48': ((128, 128, 128), (1.0, 1.0, 1.0), None, (96, 96, 96), (48, 48, 48)),
        'vol128_mm_spacing0p8': ((128, 160, 96), (0.8, 0.8, 0.8), None, (64.0, 64.0, 48.0), (32.0, 32.0, 24.0)),
        'vol_aniso': ((112, 96, 80), (0.5, 1.0, 2.0), None, (40.0, 50.0, 96.0), (20.0, 25.0, 48.0)),
        'bbox_inside': ((256, 256, 192), (1.0, 1.0, 1.0), ((37, 50, 20), (200, 190, 150)), (96, 96, 96), (48, 48, 48)),
        'bbox_near_edge': ((128, 128, 128), (1.0, 1.0, 1.0), ((60, 70, 90), (128, 128, 128)), (64, 64, 64), (32, 32, 32)),
        'box_larger_than_volume': ((64, 64, 64), (1.0, 1.0, 1.0), None, (96, 96, 96), (48, 48, 48)),
        'stride_larger_than_box': ((160, 160, 160), (1.0, 1.0, 1.0), None, (64, 64, 64), (80, 80, 80)),
    }
    out = {}
    for name, (size, spacing, bbox, psize, pstride) in cases.items():
        s0 = [0, 0, 0] if bbox is None else list(bbox[0])
        e0 = list(size) if bbox is None else list(bbox[1])
        starts, ends = ref_fn(_DuckImage(size, spacing), list(s0), list(e0), list(psize), list(pstride), 16)
        out[name] = {'size': list(size), 'spacing': list(spacing), 'bbox_start': s0, 'bbox_end': e0,
                     'partition_size': list(psize), 'partition_stride': list(pstride), 'max_stride': 16,
                     'starts': [[int(v) for v in s] for s in starts], 'ends': [[int(v) for v in e] for e in ends]}
        print('partition', name, len(starts))
    with open(os.path.join(OUT, 'partitions.json'), 'w') as f:
        json.dump(out, f)


def gen_metrics():
    src = open(os.path.join(REF, 'segmentation3d', 'utils', 'metrics.py')).read()
    tree = ast.parse(src)
    fn = [n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name == 'cal_dsc'][0]

    class _NoSitk(object):      # the function only asks `isinstance(x, sitk.Image)`; numpy arrays never are
        class Image(object):
            pass
    ns = {'np': np, 'sitk': _NoSitk}
    exec(compile(ast.Module(body=[fn], type_ignores=[]), 'metrics.py', 'exec'), ns)  # runs in memory only
    ref_fn = ns['cal_dsc']
    out = {}
    for name, gt, seg, labels, threshold in detgen.metric_label_cases():
        res = []
        for l in labels:
            dsc, typ = ref_fn(gt, seg, l, threshold)
            res.append([int(l), float(dsc), typ])
        out[name] = {'threshold': int(threshold), 'results': res}
        print('metric', name, res)
    with open(os.path.join(OUT, 'metrics.json'), 'w') as f:
        json.dump(out, f)


def gen_sampling():
    """crop-centre helpers of the reference executed through `ast` (their modules import SimpleITK): the two methods
    SegmentationDataset.global_sample / center_sample on a duck-typed image, and
    select_random_voxels_in_multi_class_mask on a numpy mask -- with numpy's seeded global RNG"""
    import types
    src = open(os.path.join(REF, 'segmentation3d', 'dataloader', 'dataset.py')).read()
    cls = [n for n in ast.parse(src).body if isinstance(n, ast.ClassDef) and n.name == 'SegmentationDataset'][0]
    fns = [n for n in cls.body if isinstance(n, ast.FunctionDef) and n.name in ('global_sample', 'center_sample')]

    class _Duck(object):
        def __init__(self, size, spacing, origin, direction):
            self._s, self._sp, self._o, self._d = size, spacing, origin, np.asarray(direction, dtype=np.float64).reshape(3, 3)

        def GetSize(self):
            return self._s

        def GetSpacing(self):
            return self._sp

        def GetOrigin(self):
            return self._o

        def TransformIndexToPhysicalPoint(self, index):
            return tuple(np.asarray(self._o) + self._d @ (np.asarray(self._sp) * np.asarray(index, dtype=np.float64)))

    class _Sitk(object):
        Image = _Duck

        @staticmethod
        def GetArrayFromImage(image):
            return image
    ns = {'np': np, 'sitk': _Sitk}
    exec(compile(ast.Module(body=fns, type_ignores=[]), 'dataset.py', 'exec'), ns)
    src2 = open(os.path.join(REF, 'segmentation3d', 'utils', 'image_tools.py')).read()
    fn2 = [n for n in ast.parse(src2).body if isinstance(n, ast.FunctionDef) and n.name == 'select_random_voxels_in_multi_class_mask'][0]

    class _Arr(np.ndarray):
        pass
    _Sitk.Image = (_Duck, np.ndarray)          # isinstance(mask, sitk.Image) must accept the numpy mask
    ns2 = {'np': np, 'sitk': _Sitk}
    exec(compile(ast.Module(body=[fn2], type_ignores=[]), 'image_tools.py', 'exec'), ns2)
    out = {}
    for name, size, spacing, origin, direction, crop_size, crop_spacing, seed in detgen.sampling_cases():
        me = types.SimpleNamespace(crop_size=np.array(crop_size, dtype=np.int32), spacing=np.array(crop_spacing, dtype=np.double))
        img = _Duck(size, spacing, origin, direction)
        np.random.seed(seed)
        g = [[float(v) for v in ns['global_sample'](me, img)] for _ in range(3)]
        c = [float(v) for v in ns['center_sample'](me, img)]
        out[name] = {'global': g, 'center': c}
        print('sampling', name, g[0], c)
    mask = detgen.labels(601, 'sampling/mask', (20, 24, 28), 4).astype(np.int8)
    np.random.seed(21)
    picks = []
    for label in (1, 2, 3, 1, 9):
        sel = ns2['select_random_voxels_in_multi_class_mask'](mask, 1, label)
        picks.append([int(label), [int(v) for v in sel[0]] if len(sel) else None])
    out['select_voxels'] = {'seed': 21, 'picks': picks, 'after': float(np.random.uniform())}
    print('select', picks)
    with open(os.path.join(OUT, 'sampling.json'), 'w') as f:
        json.dump(out, f)


class _DuckSitkImage(object):
    """what the reference's helpers ask of a `sitk.Image`: a [z, y, x] array plus a frame.  `GetArrayFromImage` COPIES,
    as SimpleITK does (add_image_region relies on writing the copy back through GetImageFromArray)"""

    def __init__(self, array, spacing=(1.0, 1.0, 1.0), origin=(0.0, 0.0, 0.0), direction=(1.0, 0, 0, 0, 1.0, 0, 0, 0, 1.0)):
        self.a = np.array(array)
        self.sp, self.o, self.d = tuple(spacing), tuple(origin), tuple(direction)

    def GetSize(self):
        z, y, x = self.a.shape
        return (x, y, z)

    def GetSpacing(self):
        return self.sp

    def GetOrigin(self):
        return self.o

    def GetDirection(self):
        return self.d

    def SetSpacing(self, v):
        self.sp = tuple(float(x) for x in v)

    def SetOrigin(self, v):
        self.o = tuple(float(x) for x in v)

    def SetDirection(self, v):
        self.d = tuple(float(x) for x in v)

    def CopyInformation(self, other):
        self.sp, self.o, self.d = other.sp, other.o, other.d

    def GetPixelID(self):
        return self.a.dtype


class _DuckSitk(object):
    Image = _DuckSitkImage
    sitkFloat32, sitkInt8, sitkInt16 = np.dtype(np.float32), np.dtype(np.int8), np.dtype(np.int16)

    @staticmethod
    def GetArrayFromImage(image):
        return np.array(image.a)

    @staticmethod
    def GetImageFromArray(array):
        return _DuckSitkImage(np.array(array))

    @staticmethod
    def Cast(image, pixel_id):
        out = _DuckSitkImage(image.a.astype(pixel_id), image.sp, image.o, image.d)
        return out


def _ref_defs(rel_path, names, ns):
    """execute the named top-level functions / classes of one reference file in the namespace `ns` (in memory only)"""
    tree = ast.parse(open(os.path.join(REF, 'segmentation3d', rel_path)).read())
    nodes = [n for n in tree.body if isinstance(n, (ast.FunctionDef, ast.ClassDef)) and n.name in names]
    assert len(nodes) == len(names), (rel_path, names)
    exec(compile(ast.Module(body=nodes, type_ignores=[]), rel_path, 'exec'), ns)


def gen_normalizers():
    """normalize_image / get_mean_std_from_image / add_image_region / add_image_value / convert_* of the reference's
    utils/image_tools.py and the two normaliser classes of utils/normalizer.py, executed through `ast` on a duck-typed
    sitk (their modules import SimpleITK).  Inputs come from detgen (see detgen.normalizer_cases), only outputs are
    stored."""
    ns = {'np': np, 'sitk': _DuckSitk, 'torch': torch,
          'type_conversion_from_numpy_to_sitk': {np.int8: _DuckSitk.sitkInt8, np.int16: _DuckSitk.sitkInt16,
                                                 np.float32: _DuckSitk.sitkFloat32}}
    _ref_defs('utils/image_tools.py', ['get_image_frame', 'set_image_frame', 'normalize_image', 'get_mean_std_from_image',
                                       'add_image_region', 'add_image_value', 'convert_image_to_tensor',
                                       'convert_tensor_to_image'], ns)
    _ref_defs('utils/normalizer.py', ['FixedNormalizer', 'AdaptiveNormalizer'], ns)
    data = {}
    for name, roi, kind, params in detgen.normalizer_cases():
        img = _DuckSitkImage(roi, (0.5, 0.75, 1.25), (3.0, -2.0, 7.0))
        if kind == 'fixed':
            norm = ns['FixedNormalizer'](params['mean'], params['stddev'], params['clip'])
        else:
            norm = ns['AdaptiveNormalizer'](params['clip_sigma'])
        out = norm(img)
        assert isinstance(out, _DuckSitkImage) and out.a.dtype == roi.dtype
        data['norm/' + name] = out.a
        d = norm.to_dict()
        data['normdict/' + name] = np.array(json.dumps({k: (bool(v) if isinstance(v, bool) else float(v)) for k, v in d.items()}))
        m, s = ns['get_mean_std_from_image'](img)
        data['meanstd/' + name] = np.array([m, s], dtype=np.float64)
        print('normalizer', name, kind, float(out.a.min()), float(out.a.max()), float(m), float(s))
    # accumulate: the reference's loop core/seg_infer.py:313-323 over a small partition (overlaps 1..8), per class
    vol_shape, patches = detgen.accumulate_case()
    C = 3
    acc = [_DuckSitkImage(np.zeros(vol_shape, dtype=np.float32)) for _ in range(C)]
    cnt = _DuckSitkImage(np.zeros(vol_shape, dtype=np.float32))
    for k, (start, end) in enumerate(patches):
        size = [end[d] - start[d] for d in range(3)]
        for c in range(C):
            patch = _DuckSitkImage(detgen.uniform(71, 'acc/p{}c{}'.format(k, c), (size[2], size[1], size[0])).astype(np.float32))
            acc[c] = ns['add_image_region'](acc[c], list(start), list(end), patch)
        cnt = ns['add_image_value'](cnt, list(start), list(end), 1.0)
    data['acc/sum'] = np.stack([a.a for a in acc])
    data['acc/count'] = cnt.a
    print('accumulate', data['acc/sum'].shape, float(cnt.a.min()), float(cnt.a.max()))
    # tensor <-> image conversion (layout contract: tensor dims = (C, z, y, x))
    a0 = detgen.normal(72, 'conv/a0', (5, 6, 7))
    a1 = detgen.normal(72, 'conv/a1', (5, 6, 7))
    t1 = ns['convert_image_to_tensor'](_DuckSitkImage(a0))
    t2 = ns['convert_image_to_tensor']([_DuckSitkImage(a0), _DuckSitkImage(a1)])
    data['convert/single'] = t1.numpy()
    data['convert/list'] = t2.numpy()
    im3 = ns['convert_tensor_to_image'](torch.from_numpy(a0 * 10), np.int8)
    im4 = ns['convert_tensor_to_image'](torch.from_numpy(np.stack([a0, a1])), None)
    data['convert/to_image_int8'] = im3.a
    data['convert/to_image_list'] = np.stack([im.a for im in im4])
    np.savez_compressed(os.path.join(OUT, 'normalizers.npz'), **data)


def gen_small_losses():
    """BinaryDiceLoss on its own (loss/binary_dice_loss.py:9-36) and the CrossEntropyLoss plugin applied to soft-max
    OUTPUT (the reference's double soft-max, loss/cross_entropy_loss.py:13-18), values + input gradients"""
    from segmentation3d.loss.binary_dice_loss import BinaryDiceLoss
    from segmentation3d.loss.cross_entropy_loss import CrossEntropyLoss
    data = {}
    for name, shape in (('b2', (2, 2, 6, 8, 10)), ('b3_ties', (3, 2, 4, 4, 8))):
        logits = detgen.normal(81, 'bdice/' + name, shape, std=1.5)
        e = np.exp(logits - logits.max(1, keepdims=True))
        probs = (e / e.sum(1, keepdims=True)).astype(np.float32)
        if 'ties' in name:
            probs[:, :, 0] = 0.5          # exact ties: arg-max takes channel 0 -> pred 0
            probs[1] = np.float32(0.25)   # a sample where nothing is foreground
            probs[1, 0] = np.float32(0.75)
        target = detgen.labels(82, 'bdice/t' + name, (shape[0], 1) + shape[2:], 2)
        pt = torch.from_numpy(probs).clone().requires_grad_(True)
        loss = BinaryDiceLoss()(pt * 1.0, torch.from_numpy(target))    # (the module writes into its max() result, not the input)
        loss.backward()
        data['bdice_{}/probs'.format(name)] = probs
        data['bdice_{}/target'.format(name)] = target
        data['bdice_{}/loss'.format(name)] = np.float32(loss.item())
        data['bdice_{}/grad'.format(name)] = pt.grad.numpy()
        print('binary dice', name, float(loss))
    for C, shape in ((2, (2, 2, 4, 6, 8)), (5, (1, 5, 4, 4, 8))):
        logits = detgen.normal(83, 'ce/logits{}'.format(C), shape, std=2.0)
        e = np.exp(logits - logits.max(1, keepdims=True))
        probs = (e / e.sum(1, keepdims=True)).astype(np.float32)
        target = detgen.labels(84, 'ce/t{}'.format(C), (shape[0], 1) + shape[2:], C)
        pt = torch.from_numpy(probs).clone().requires_grad_(True)
        loss = CrossEntropyLoss()(pt, torch.from_numpy(target))
        loss.backward()
        data['ce{}/probs'.format(C)] = probs
        data['ce{}/target'.format(C)] = target
        data['ce{}/loss'.format(C)] = np.float32(loss.item())
        data['ce{}/grad'.format(C)] = pt.grad.numpy()
        print('cross entropy on probabilities', C, float(loss))
    np.savez_compressed(os.path.join(OUT, 'small_losses.npz'), **data)


def gen_shapes():
    from segmentation3d.network import vnet, vbnet
    out = {}
    for plugin_name, plugin in (('vnet', vnet), ('vbnet', vbnet)):
        for cin, ncls in ((1, 2), (1, 5), (4, 4)):
            net = plugin.SegmentationNet(cin, ncls)
            out['{}_{}_{}'.format(plugin_name, cin, ncls)] = {
                'keys': [[k, list(v.shape)] for k, v in net.state_dict().items()],
                'num_params': int(sum(p.numel() for p in net.parameters())), 'max_stride': int(net.max_stride())}
    with open(os.path.join(OUT, 'state_dict_shapes.json'), 'w') as f:
        json.dump(out, f)
    print('shapes', {k: v['num_params'] for k, v in out.items()})


if __name__ == '__main__':
    os.makedirs(OUT, exist_ok=True)
    which = sys.argv[1:] or ['blocks', 'nets', 'losses', 'partitions', 'shapes', 'metrics', 'sampling', 'normalizers',
                              'small_losses']
    if 'blocks' in which:
        gen_blocks()
    if 'nets' in which:
        gen_nets()
    if 'losses' in which:
        gen_losses()
    if 'partitions' in which:
        gen_partitions()
    if 'shapes' in which:
        gen_shapes()
    if 'metrics' in which:
        gen_metrics()
    if 'sampling' in which:
        gen_sampling()
    if 'normalizers' in which:
        gen_normalizers()
    if 'small_losses' in which:
        gen_small_losses()
