"""whole-volume inference jobs back to back with the caching allocator's statistics after each (driver allocations /
frees per job, reserved bytes): found the ~100 hipMalloc / hipFree calls per job behind the slow inference readings.
usage: python tools/infer_mem.py [patches per replay]"""
import os, sys, time, json
import torch
REPO = '/root/repo' if os.path.isdir('/root/repo/tools') else os.environ.get('GRAFT_REPO_ROOT', '.')
sys.path.insert(0, os.path.join(REPO, 'medical-segmentation3d-toolkit_amd')); sys.path.insert(0, REPO)
from segmentation3d.core.seg_infer import sliding_window_inference
from segmentation3d.utils.image_tools import image_partition_by_fixed_size
from segmentation3d.network import vnet
dev = torch.device('cuda:0')
torch.manual_seed(0)
net = vnet.SegmentationNet(1, 2); vnet.parameters_kaiming_init(net); net = net.to(dev).eval()
batch = int(sys.argv[1]) if len(sys.argv) > 1 else 16
X, Y, Z = 512, 512, 400
vol = torch.randn((Z, Y, X), generator=torch.Generator().manual_seed(7)).to(dev)
starts, ends = image_partition_by_fixed_size(((X, Y, Z), (1.0, 1.0, 1.0)), [0, 0, 0], [X, Y, Z], [96] * 3, [48] * 3, 16)
keys = ['num_alloc_retries', 'num_device_alloc', 'num_device_free', 'reserved_bytes.all.current', 'allocated_bytes.all.peak']
for job in range(4):
    torch.cuda.synchronize(); t0 = time.time()
    probs, mask, _ = sliding_window_inference(net, vol, starts, (96,) * 3, 2, {'type': 1, 'clip_sigma': 3}, batch_size=batch, use_graph=True, two_streams=True)
    torch.cuda.synchronize(); dt = time.time() - t0
    st = torch.cuda.memory_stats()
    print('job', job, 'batch', batch, 'seconds %.3f' % dt, {k: (st[k] if 'bytes' not in k else round(st[k] / 1e9, 2)) for k in keys}, flush=True)
    del probs, mask
