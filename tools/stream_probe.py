import os, sys
sys.path.insert(0, '/root/repo/medical-segmentation3d-toolkit_amd'); sys.path.insert(0, '/root/repo')
import torch, torch.distributed as dist
os.environ.setdefault('MASTER_ADDR', '127.0.0.1'); os.environ.setdefault('MASTER_PORT', '29534')
torch.cuda.set_device(0)
if 'dist' in sys.argv:
    dist.init_process_group('nccl', rank=0, world_size=1, device_id=torch.device('cuda:0'))
from segmentation3d import _ops
main = torch.cuda.current_stream()
cands = [torch.cuda.Stream() for _ in range(8)]
print('main', main, [int(_ops._runs_beside(main, c)) for c in cands])
# pairwise among candidates
for i in range(4):
    print(i, [int(_ops._runs_beside(cands[i], c)) if c is not cands[i] else '-' for c in cands])
# the order that matters to the train step: a long kernel on the SIDE stream is launched first, short kernels follow on main
print('spin on cand, probe on main:', [int(_ops._runs_beside(c, main)) for c in cands])
