"""CPU-side checks of the plugin API mirror: import paths, constructor signatures, state_dict contract, error
behaviour (no CPU fallback), config/partition host logic."""
import importlib

import numpy as np
import pytest
import torch

from conftest import golden_json


@pytest.mark.parametrize('plugin', ['vnet', 'vbnet'])
@pytest.mark.parametrize('cin,ncls', [(1, 2), (1, 5), (4, 4)])
def test_state_dict_contract(plugin, cin, ncls):
    ref = golden_json('state_dict_shapes')['{}_{}_{}'.format(plugin, cin, ncls)]
    mod = importlib.import_module('segmentation3d.network.' + plugin)       # core/seg_train.py:72
    net = mod.SegmentationNet(cin, ncls)
    sd = net.state_dict()
    assert [[k, list(v.shape)] for k, v in sd.items()] == ref['keys']       # same keys, order and shapes
    assert sum(p.numel() for p in net.parameters()) == ref['num_params']
    assert net.max_stride() == ref['max_stride'] == 16
    assert all(v.dtype == torch.float32 for v in sd.values())


def test_kaiming_init_matches_reference_semantics():
    from segmentation3d.network import vnet
    net = vnet.SegmentationNet(1, 2)
    torch.manual_seed(0)
    vnet.parameters_kaiming_init(net)
    sd = net.state_dict()
    assert float(sd['in_block.conv.bias'].abs().max()) == 0.0              # conv biases zeroed (weight_init.py:9-10)
    assert torch.all(sd['in_block.gn.weight'] == 1) and torch.all(sd['in_block.gn.bias'] == 0)  # GN untouched
    w = sd['up_32.rblock.ops.0.conv.weight']
    assert abs(float(w.std()) - (2.0 / (32 * 27)) ** 0.5) < 5e-4            # kaiming_normal_, fan_in, gain sqrt(2)
    vnet.parameters_gaussian_init(net)
    assert abs(float(net.state_dict()['up_32.rblock.ops.0.conv.weight'].std()) - 0.01) < 1e-3


def test_module_prefix_checkpoints_load():
    """DataParallel checkpoints carry a 'module.' prefix (core/seg_infer.py:130-142)"""
    from segmentation3d.network import vbnet
    from segmentation3d.utils.model_io import strip_module_prefix
    net = vbnet.SegmentationNet(1, 2)
    sd = {'module.' + k: v.clone() for k, v in net.state_dict().items()}
    net2 = vbnet.SegmentationNet(1, 2)
    net2.load_state_dict(strip_module_prefix(sd))
    for k, v in net.state_dict().items():
        assert torch.equal(v, net2.state_dict()[k])


def test_no_cpu_fallback():
    """the product path must fail loudly on CPU tensors instead of silently using torch CPU ops"""
    from segmentation3d.network import vnet
    from segmentation3d.loss.multi_dice_loss import MultiDiceLoss
    from segmentation3d.loss.focal_loss import FocalLoss
    from segmentation3d._engine import Seg3dEngineError
    net = vnet.SegmentationNet(1, 2)
    with pytest.raises(Seg3dEngineError):
        net(torch.zeros(1, 1, 16, 16, 16))
    p = torch.full((1, 2, 4, 4, 4), 0.5)
    t = torch.zeros(1, 1, 4, 4, 4)
    with pytest.raises(Seg3dEngineError):
        MultiDiceLoss([1, 1], 2, use_gpu=False)(p, t)
    with pytest.raises(Seg3dEngineError):
        FocalLoss(2, use_gpu=False)(p, t)


def test_input_validation():
    from segmentation3d.network import vnet
    net = vnet.SegmentationNet(1, 2)
    with pytest.raises(ValueError):
        net(torch.zeros(1, 1, 20, 16, 16))          # not divisible by max_stride
    with pytest.raises(AssertionError):
        from segmentation3d.loss.multi_dice_loss import MultiDiceLoss
        MultiDiceLoss([1, 1, 1], 2, use_gpu=False)  # len(weights) != num_class (multi_dice_loss.py:16)


def test_reference_unit_test_shapes_are_constructible():
    """network/module/conv_gn_relu3_test.py:19-38 builds ConvGnRelu3 with (k3,s1,p1) and (k2,s2,p0)"""
    from segmentation3d.network.module.conv_gn_relu3 import ConvGnRelu3
    a = ConvGnRelu3(1, 16, 3, 1, 1, do_act=True)
    b = ConvGnRelu3(1, 16, 2, 2, 0, do_act=False)
    assert tuple(a.conv.weight.shape) == (16, 1, 3, 3, 3) and tuple(b.conv.weight.shape) == (16, 1, 2, 2, 2)
    with pytest.raises(ValueError):
        ConvGnRelu3(1, 16, 5, 1, 2)                 # 5x5x5 does not exist in the reference networks


def test_partition_matches_reference_tables():
    from segmentation3d.utils.image_tools import image_partition_by_fixed_size
    for name, case in golden_json('partitions').items():
        starts, ends = image_partition_by_fixed_size(
            (case['size'], case['spacing']), list(case['bbox_start']), list(case['bbox_end']), case['partition_size'],
            case['partition_stride'], case['max_stride'])
        assert starts == case['starts'], name
        assert ends == case['ends'], name
    c = golden_json('partitions')['vol512x512x400_96_48']
    assert len(c['starts']) == 800


def test_image_io_roundtrips_nifti_and_compressed_mha(tmp_path):
    """utils/image_io: NIfTI-1 (.nii / .nii.gz) write -> read round trip keeps voxels and the ITK-convention frame
    (RAS <-> LPS sign flips), header fields sit at the NIfTI-1 offsets, the qform path agrees with the sform path, and a
    zlib-compressed MetaImage (what sitk.WriteImage(..., True) produces) is readable"""
    import gzip
    import struct
    import zlib
    from segmentation3d.utils.image3d import Image3d
    from segmentation3d.utils.image_io import read_image, write_image
    rng = np.random.RandomState(3)
    arr = rng.randn(5, 7, 9).astype(np.float32)
    direction = [0.0, -1.0, 0.0, 1.0, 0.0, 0.0, 0.0, 0.0, 1.0]            # a 90 degree in-plane rotation
    img = Image3d(arr, (0.8, 1.25, 2.5), (-12.5, 30.0, 4.0), direction)
    for name in ('a.nii.gz', 'b.nii'):
        path = str(tmp_path / name)
        write_image(img, path)
        back = read_image(path, dtype=None)
        assert back.array.dtype == np.float32 and np.array_equal(back.array, arr)
        assert np.allclose(back.GetSpacing(), img.GetSpacing(), atol=1e-6)
        assert np.allclose(back.GetOrigin(), img.GetOrigin(), atol=1e-5)
        assert np.allclose(back.GetDirection(), direction, atol=1e-6)
    raw = gzip.open(str(tmp_path / 'a.nii.gz'), 'rb').read()
    assert struct.unpack('<i', raw[:4])[0] == 348 and raw[344:348] == b'n+1\x00'
    assert struct.unpack('<8h', raw[40:56])[:4] == (3, 9, 7, 5) and struct.unpack('<h', raw[70:72])[0] == 16
    # RAS affine: x and y rows carry the opposite sign of the LPS frame
    srow_x = struct.unpack('<4f', raw[280:296])
    assert abs(srow_x[3] - 12.5) < 1e-5
    # the same geometry written as a qform only
    hdr = bytearray(raw[:352])
    struct.pack_into('<2h', hdr, 252, 1, 0)
    R = np.diag([-1.0, -1.0, 1.0]) @ np.array(direction).reshape(3, 3)
    a = 0.5 * np.sqrt(max(0.0, 1.0 + R[0, 0] + R[1, 1] + R[2, 2]))
    b, c, d = (R[2, 1] - R[1, 2]) / (4 * a), (R[0, 2] - R[2, 0]) / (4 * a), (R[1, 0] - R[0, 1]) / (4 * a)
    struct.pack_into('<6f', hdr, 256, b, c, d, 12.5, -30.0, 4.0)
    qpath = str(tmp_path / 'q.nii')
    open(qpath, 'wb').write(bytes(hdr) + raw[352:])
    q = read_image(qpath)
    assert np.allclose(q.GetDirection(), direction, atol=1e-6) and np.allclose(q.GetOrigin(), img.GetOrigin(), atol=1e-5)
    # compressed MetaImage
    lab = rng.randint(0, 4, size=(4, 6, 8)).astype(np.int16)
    header = ('ObjectType = Image\nNDims = 3\nBinaryData = True\nBinaryDataByteOrderMSB = False\nCompressedData = True\n'
              'TransformMatrix = 1 0 0 0 1 0 0 0 1\nOffset = 1 2 3\nElementSpacing = 0.5 0.5 2\nDimSize = 8 6 4\n'
              'ElementType = MET_SHORT\nElementDataFile = LOCAL\n')
    cpath = str(tmp_path / 'c.mha')
    open(cpath, 'wb').write(header.encode('ascii') + zlib.compress(lab.tobytes()))
    c_img = read_image(cpath, dtype=None)
    assert np.array_equal(c_img.array, lab) and c_img.GetSpacing() == (0.5, 0.5, 2.0) and c_img.GetOrigin() == (1.0, 2.0, 3.0)
    with pytest.raises(ValueError):
        read_image(str(tmp_path / 'x.dcm'))


def test_tensor_on_another_device_is_refused(monkeypatch):
    """the engine launches on the CURRENT device's stream: a tensor living on another GPU must raise at the operator
    entry instead of being launched on the wrong queue (checked here with a stand-in tensor, no GPU needed)"""
    import torch
    from segmentation3d import _engine

    class OnGpu1(torch.Tensor):
        @property
        def is_cuda(self):
            return True

        @property
        def device(self):
            return torch.device('cuda', 1)
    t = torch.zeros(4).as_subclass(OnGpu1)
    monkeypatch.setattr(torch.cuda, 'current_device', lambda: 0)
    with pytest.raises(_engine.Seg3dEngineError, match='current device'):
        _engine.require_device(t)
    monkeypatch.setattr(torch.cuda, 'current_device', lambda: 1)
    _engine.require_device(t)          # same device: accepted


def test_epoch_accounting_counts_the_global_batch():
    from segmentation3d.core.seg_train import epoch_of_batch
    assert epoch_of_batch(10, 4, 40) == 1 and epoch_of_batch(9, 4, 40) == 0           # the reference's formula at world 1
    assert epoch_of_batch(5, 4, 40, world_size=2) == 1 and epoch_of_batch(10, 4, 40, world_size=8) == 8


_REF_CONFIG_DIR = '/root/reference/segmentation3d/config'


@pytest.mark.skipif(not __import__('os').path.isdir(_REF_CONFIG_DIR), reason='the reference tree is only present in the build container')
def test_reference_config_files_load_unchanged():
    """north_star: `config/train_config.py` drops in unchanged.  The reference's OWN two config files are loaded through
    this package's load_config (utils/file_io.py:8-28 of the reference) with this package as `segmentation3d` on the path:
    every section / key the engines read is there, and the values are the reference's.  Nothing of the reference travels:
    the test reads it in place and is skipped where /root/reference does not exist (the GPU box)."""
    import os
    import segmentation3d
    from segmentation3d.utils.file_io import load_config
    from segmentation3d.utils.normalizer import AdaptiveNormalizer
    assert 'medical-segmentation3d-toolkit_amd' in segmentation3d.__file__
    tc = load_config(os.path.join(_REF_CONFIG_DIR, 'train_config.py'))
    assert tc.net.name == 'vbnet' and tc.loss.name == 'Focal' and tc.loss.focal_gamma == 2
    assert tc.train.batchsize == 4 and tc.train.lr == 1e-4 and tuple(tc.train.betas) == (0.9, 0.999)
    assert tc.dataset.num_classes == 2 and list(tc.dataset.crop_size) == [64, 64, 64] and tc.dataset.sampling_method == 'HYBRID'
    assert isinstance(tc.dataset.crop_normalizers[0], AdaptiveNormalizer)          # the config imports OUR normalizer module
    assert tc.general.resume_epoch == -1 and tc.general.num_gpus == 1 and tc.debug.save_inputs is False
    ic = load_config(os.path.join(_REF_CONFIG_DIR, 'infer_config.py'))
    assert ic.general.single_scale == 'DISABLE'
    assert ic.coarse.partition_type == 'DISABLE' and ic.fine.partition_type == 'SIZE'
    assert list(ic.fine.partition_size) == [89.6] * 3 and list(ic.fine.partition_stride) == [89.6] * 3
    # the shipped defaults have the same sections and keys (values may differ where documented), and the same default mode
    mine = load_config(os.path.join(os.path.dirname(segmentation3d.__file__), 'config', 'infer_config.py'))
    assert mine.general.single_scale == ic.general.single_scale == 'DISABLE'
    for section in ('general', 'coarse', 'fine'):
        assert sorted(mine[section].keys()) == sorted(ic[section].keys()), section
    mine_t = load_config(os.path.join(os.path.dirname(segmentation3d.__file__), 'config', 'train_config.py'))
    for section in ('general', 'dataset', 'loss', 'net', 'train'):
        assert set(tc[section].keys()) <= set(mine_t[section].keys()), (section, set(tc[section].keys()) - set(mine_t[section].keys()))


def test_image_readers_against_independent_fixtures(tmp_path):
    """f3: the product readers (utils/mha_io.py, utils/image_io.py -- they replace sitk.ReadImage / sitk.WriteImage,
    core/seg_infer.py:414,467-481) on files they did NOT write: tests/golden/images/* are assembled byte by byte by
    tests/golden/make_image_fixtures.py from the MetaImage / NIfTI-1 format definitions (raw, zlib and big-endian MetaImage
    payloads, .mhd + separate data file, oblique TransformMatrix; NIfTI sform with scl_slope, qform with qfac = -1 in a
    .nii.gz, big-endian header).  Array, spacing, origin and direction must come out as the definitions say; then each
    image is written by the product writers and read back (the writers against the now independently checked readers)."""
    import os
    from conftest import GOLDEN
    from segmentation3d.utils.image_io import read_image, write_image
    folder = os.path.join(GOLDEN, 'images')
    expected = golden_json(os.path.join('images', 'expected'))
    assert len(expected) == 6
    for name, exp in sorted(expected.items()):
        img = read_image(os.path.join(folder, name), dtype=None)
        want = np.asarray(exp['array'])
        assert img.array.shape == want.shape and img.GetSize() == want.shape[::-1], name
        assert np.array_equal(np.asarray(img.array, dtype=np.float64), want.astype(np.float64)), name
        if '*' not in exp['dtype']:
            assert img.array.dtype == np.dtype(exp['dtype']), (name, img.array.dtype)
        assert np.allclose(img.GetSpacing(), exp['spacing'], rtol=1e-6, atol=0), (name, img.GetSpacing())
        assert np.allclose(img.GetOrigin(), exp['origin'], rtol=1e-6, atol=1e-6), (name, img.GetOrigin())
        assert np.allclose(img.GetDirection(), exp['direction'], rtol=0, atol=1e-6), (name, img.GetDirection())
        as_f32 = read_image(os.path.join(folder, name))                      # sitk.ReadImage(path, sitk.sitkFloat32)
        assert as_f32.array.dtype == np.float32 and np.allclose(as_f32.array, want, rtol=1e-6)
        for ext in ('.mha', '.nii.gz'):
            out = str(tmp_path / (name.split('.')[0] + '_rt' + ext))
            write_image(img, out)
            back = read_image(out, dtype=None)
            assert np.array_equal(back.array, img.array) and back.array.dtype == img.array.dtype, (name, ext)
            assert np.allclose(back.GetSpacing(), exp['spacing'], rtol=1e-6) and np.allclose(back.GetOrigin(), exp['origin'], rtol=1e-6, atol=1e-6)
            assert np.allclose(back.GetDirection(), exp['direction'], atol=1e-6), (name, ext)


def test_gradient_bucket_cuts_of_the_vnet():
    """core/ddp.bucket_cuts on the real V-Net layout: buckets tile the flat gradient buffer, complete in backward order (end of
    the buffer first), and the last-completing one -- the only all-reduce nothing can overlap -- is the small tail (stem and
    first encoder stages, < 1.1 MB) split off the deep encoder stages"""
    from segmentation3d.core.ddp import bucket_cuts
    from segmentation3d.network import vnet
    net = vnet.SegmentationNet(1, 2)
    entries, off = [], 0
    for name, p in net.named_parameters():
        entries.append((p, off, p.numel()))
        off += p.numel()
    cuts = bucket_cuts(entries, off, num_buckets=4)
    assert len(cuts) == 5
    assert cuts[0][1] == off and cuts[-1][0] == 0
    assert all(a[0] == b[1] for a, b in zip(cuts[:-1], cuts[1:]))                 # contiguous, descending
    assert sum(len(c[2]) for c in cuts) == len(entries)
    names = {id(p): n for n, p in net.named_parameters()}
    tail = [names[id(p)] for p in cuts[-1][2]]
    assert 'in_block.conv.weight' in tail and all(n.split('.')[0] in ('in_block', 'down_32', 'down_64') for n in tail)
    assert 0 < cuts[-1][1] <= (1 << 18) and cuts[-2][1] - cuts[-2][0] > (1 << 20)  # <= 1 MB exposed, > 4 MB moved under backward
    assert len(bucket_cuts(entries, off, num_buckets=4, tail_elems=0)) == 4
    small = entries[:6]
    assert len(bucket_cuts(small, sum(e[2] for e in small), num_buckets=2)) == 2   # small buffers are not split
