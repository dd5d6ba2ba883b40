"""world_size-2 gloo tests (CPU) of the data-parallel path: the bucketed, backward-overlapped gradient all-reduce
of segmentation3d/core/ddp.py reproduces the full-batch gradient, and the inference patch sharding covers every
patch exactly once."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _toy():
    torch.manual_seed(0)
    return torch.nn.Sequential(torch.nn.Conv3d(1, 4, 3, padding=1), torch.nn.GroupNorm(1, 4), torch.nn.ReLU(),
                               torch.nn.Conv3d(4, 4, 3, padding=1), torch.nn.GroupNorm(1, 4), torch.nn.ReLU(),
                               torch.nn.Conv3d(4, 2, 1))


def _worker(rank, world, port, out):
    import sys
    from conftest import PKG  # noqa: F401
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from segmentation3d.core.ddp import FlatGradients, GradientReducer
    torch.set_num_threads(1)
    net = _toy()
    if rank == 1:                       # perturb rank 1: broadcast_parameters must repair it
        with torch.no_grad():
            for p in net.parameters():
                p.add_(1.0)
    flat = FlatGradients(list(net.parameters()))
    red = GradientReducer([flat.buffer], flat.layout(), num_buckets=3)
    params_flat = torch.cat([p.data.reshape(-1) for p in net.parameters()])
    red.broadcast_parameters(params_flat, src=0)
    off = 0
    with torch.no_grad():
        for p in net.parameters():
            p.copy_(params_flat[off:off + p.numel()].view_as(p))
            off += p.numel()
    g = torch.Generator().manual_seed(100)
    x_all = torch.randn(4, 1, 6, 6, 6, generator=g)
    losses = []
    for step in range(2):
        flat.zero()
        red.begin_step()
        x = x_all[2 * rank:2 * rank + 2]
        loss = net(x).pow(2).mean()
        loss.backward()
        red.finish_step()
        losses.append(float(loss))
    if rank == 0:
        torch.save({'grad_sum': flat.buffer.clone(), 'buckets': red.bucket_sizes(), 'world': red.world_size}, out)
    dist.barrier()
    dist.destroy_process_group()


def test_bucketed_allreduce_equals_full_batch_gradient(tmp_path):
    world, port, out = 2, _free_port(), str(tmp_path / 'r0.pt')
    mp.spawn(_worker, args=(world, port, out), nprocs=world, join=True)
    got = torch.load(out, weights_only=True)
    net = _toy()
    g = torch.Generator().manual_seed(100)
    x_all = torch.randn(4, 1, 6, 6, 6, generator=g)
    # mean over the per-rank batch losses == loss of the global batch (equal per-rank batch sizes)
    loss = 0.5 * (net(x_all[:2]).pow(2).mean() + net(x_all[2:]).pow(2).mean())
    loss.backward()
    ref = torch.cat([p.grad.reshape(-1) for p in net.parameters()])
    assert got['world'] == 2 and 2 <= len(got['buckets']) <= 3 and sum(got['buckets']) == ref.numel()
    torch.testing.assert_close(got['grad_sum'] / 2.0, ref, rtol=1e-5, atol=1e-7)


def test_inference_patch_sharding_is_a_partition():
    from segmentation3d.core.seg_infer import shard_batches
    batches = [list(range(i, min(i + 8, 803))) for i in range(0, 803, 8)]
    seen = []
    for rank in range(4):
        for b in shard_batches(batches, rank, 4):
            seen.extend(b)
    assert sorted(seen) == list(range(803))
    assert shard_batches(batches, 0, 1) == batches
