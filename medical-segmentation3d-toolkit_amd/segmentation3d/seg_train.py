"""`seg_train` command line -- same flag as the reference's segmentation3d/seg_train.py:14-18 (-i config file).
Launch one process per GPU for data-parallel training:
    python -m torch.distributed.run --nproc-per-node 8 --master-addr 127.0.0.1 -m segmentation3d.seg_train -i cfg.py
"""
import argparse
import os

import torch
import torch.distributed as dist

from segmentation3d.core.seg_train import train


def main():
    parser = argparse.ArgumentParser(description='Training engine for 3D medical image segmentation (HIP engine).')
    parser.add_argument('-i', '--input', required=True, help='training config file (python, see config/train_config.py)')
    parser.add_argument('--synthetic-steps', type=int, default=0,
                        help='train on synthetic patches for this many steps (no dataset needed)')
    args = parser.parse_args()
    world = int(os.environ.get('WORLD_SIZE', '1'))
    if world > 1:
        torch.cuda.set_device(int(os.environ.get('LOCAL_RANK', '0')))
        dist.init_process_group('nccl')
    factory = None
    if args.synthetic_steps > 0:
        def factory(cfg):
            g = torch.Generator().manual_seed(cfg.general.seed + int(os.environ.get('RANK', '0')))
            size = list(cfg.dataset.crop_size)[::-1]
            for _ in range(args.synthetic_steps):
                x = torch.randn([cfg.train.batchsize, 1] + size, generator=g).clamp_(-3, 3)
                t = torch.randint(0, cfg.dataset.num_classes, [cfg.train.batchsize, 1] + size, generator=g).float()
                yield x, t
    train(args.input, data_iter_factory=factory)
    if world > 1:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
