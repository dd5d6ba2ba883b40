"""does an initialised RCCL communicator change the whole-volume inference job (hipGraph replays with two half-batch branches)?
usage: python tools/infer_with_comm.py [dist] [train]   (dist: one-rank nccl process group first; train: a few train steps
first, as bench.py does before its inference leg)"""
import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, 'medical-segmentation3d-toolkit_amd')); sys.path.insert(0, REPO)
import torch
import torch.distributed as dist
os.environ.setdefault('MASTER_ADDR', '127.0.0.1'); os.environ.setdefault('MASTER_PORT', '29535')
os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
torch.cuda.set_device(0)
dev = torch.device('cuda:0')
if 'dist' in sys.argv:
    dist.init_process_group('nccl', rank=0, world_size=1, device_id=dev)
import bench
if 'noslots' in sys.argv:
    from segmentation3d.network import _vnet_base
    _vnet_base.PREFILL_SKIP_SLOTS = False
from segmentation3d.core.seg_train import TrainStep
step = TrainStep('vnet', 1, 2, 'Dice', [0.5, 0.5], device=dev, seed=0, distributed='dist' in sys.argv)
if 'train' in sys.argv:
    x, t = bench.synthetic_batch(4, 1, 2, 96, dev, 1000)
    for _ in range(5):
        step(x, t)
    torch.cuda.synchronize()
r = bench.time_inference(step.net, (512, 512, 400), 96, 48, 2, 16, dev)
print('infer', r['seconds'], r['seconds_all_runs'], 'first', r['first_job_seconds'], flush=True)
if 'dist' in sys.argv:
    dist.destroy_process_group()
