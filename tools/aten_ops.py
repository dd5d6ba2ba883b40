"""list the stock torch (aten) device kernels a train step still launches: python tools/aten_ops.py [--dtype bf16]"""
import os, sys
import torch
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, 'medical-segmentation3d-toolkit_amd')); sys.path.insert(0, REPO)
import bench
from torch.profiler import profile, ProfilerActivity

def main():
    dtype = 'bf16' if 'bf16' in sys.argv else 'fp32'
    dev = torch.device('cuda:0')
    from segmentation3d.core.seg_train import TrainStep
    from segmentation3d import _ops
    _ops.set_activation_dtype(dtype)
    step = TrainStep('vnet', 1, 2, 'Dice', [0.5, 0.5], device=dev, seed=0, use_graph=False)
    batch = bench.synthetic_batch(4, 1, 2, 96, dev, 1000)
    for _ in range(3):
        step(*batch)
    torch.cuda.synchronize()
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True, with_stack=False) as prof:
        step(*batch)
        torch.cuda.synchronize()
    rows = [e for e in prof.key_averages(group_by_input_shape=True) if e.key.startswith('aten::') and e.device_time_total > 0]
    rows.sort(key=lambda e: -e.device_time_total)
    for e in rows[:25]:
        print('{:32s} n={:3d} dev_us={:9.1f} shapes={}'.format(e.key, e.count, e.device_time_total, str(e.input_shapes)[:110]))

if __name__ == '__main__':
    main()
