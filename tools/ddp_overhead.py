"""what does the data-parallel machinery cost per step apart from the transfer itself?  One rank on the real RCCL backend
(an all-reduce over one rank moves nothing): TrainStep with the gradient reducer (hooks, side-stream joins, five collective
launches per step) against the same step without it, eager launches, on one GPU.
usage: python tools/ddp_overhead.py [steps] [local|reducer|nodist]   (one arm per process: a second TrainStep in the same process
inherits a fragmented caching allocator and can read 2-3x slower on boxes where hipMalloc is slow)"""
import os, sys, time
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, 'medical-segmentation3d-toolkit_amd')); sys.path.insert(0, REPO)
import torch
import torch.distributed as dist

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 30
NODIST = 'nodist' in sys.argv   # control: no process group at all (only the 'local' arms run)
os.environ.setdefault('MASTER_ADDR', '127.0.0.1'); os.environ.setdefault('MASTER_PORT', '29533')
os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
torch.cuda.set_device(0)
dev = torch.device('cuda:0')
if not NODIST:
    if 'lazy' in sys.argv:      # no device_id: the communicator is created at the first collective
        dist.init_process_group('nccl', rank=0, world_size=1)
    elif 'gloo' in sys.argv:
        dist.init_process_group('gloo', rank=0, world_size=1)
    else:
        dist.init_process_group('nccl', rank=0, world_size=1, device_id=dev)
from segmentation3d.core.seg_train import TrainStep
import bench
x, t = bench.synthetic_batch(4, 1, 2, 96, dev, 1000)
import gc
ARM = 'reducer' if 'reducer' in sys.argv else 'local'
for name, distributed in ((ARM, ARM == 'reducer'),):
    step = TrainStep('vnet', 1, 2, 'Dice', [0.5, 0.5], device=dev, seed=0, distributed=distributed,
                     use_graph='graph' in sys.argv)   # graph: replay of the captured step (host-independent; local arm only)
    for _ in range(5):
        step(x, t)
    gc.collect(); gc.freeze()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        loss = step(x, t)
    torch.cuda.synchronize()
    print('{:8s} {:.3f} ms/step  loss {:.6f}'.format(name, 1e3 * (time.perf_counter() - t0) / steps, float(loss)), flush=True)
    step.opt.release_grad_sinks()
    if step.reducer is not None:
        step.reducer.remove_hooks()
    del step
    gc.unfreeze(); gc.collect(); torch.cuda.empty_cache()
if not NODIST:
    dist.destroy_process_group()
