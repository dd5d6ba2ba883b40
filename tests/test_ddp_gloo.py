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


def _infer_case():
    """small sliding-window job: volume [Z, Y, X] = 64 x 32 x 48, 16^3 boxes at stride 8 (overlap counts up to 8)"""
    import numpy as np
    from oracle import detgen
    from segmentation3d.utils.image_tools import image_partition_by_fixed_size
    Z, Y, X, C = 64, 32, 48, 3
    starts, ends = image_partition_by_fixed_size(((X, Y, Z), (1.0, 1.0, 1.0)), [0, 0, 0], [X, Y, Z], [16] * 3, [8] * 3, 16)

    def patch(k):
        return np.stack([detgen.uniform(91, 'shard/p{}c{}'.format(k, c), (16, 16, 16)).astype(np.float32) for c in range(C)])
    return (Z, Y, X, C), starts, ends, patch


def _infer_worker(rank, world, port, out):
    import numpy as np
    from conftest import PKG  # noqa: F401
    from oracle import numpy_ref
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    torch.set_num_threads(1)
    from segmentation3d.core.seg_infer import SlabShardPlan, merge_slabs, gather_slabs
    from segmentation3d.core.seg_train import epoch_of_batch
    from segmentation3d.dataloader.sampler import EpochConcateDistributedSampler
    (Z, Y, X, C), starts, ends, patch = _infer_case()
    plan = SlabShardPlan(starts, (16, 16, 16), (Z, Y, X), world)
    acc = np.zeros((C, Z, Y, X), np.float32)
    cnt = np.zeros((Z, Y, X), np.float32)
    for k in plan.patches[rank]:
        numpy_ref.accumulate_patch(acc, cnt, starts[k], ends[k], patch(k))
    z0, z1 = plan.owned(rank)
    assert cnt[:plan.touch_lo[rank]].sum() == 0 and cnt[plan.touch_hi[rank]:].sum() == 0     # a rank touches only its range
    acc_t, cnt_t = torch.from_numpy(acc), torch.from_numpy(cnt)
    merge_slabs(acc_t, cnt_t, plan, rank)
    probs, mask = numpy_ref.finalize(acc_t.numpy()[:, z0:z1].copy(), cnt_t.numpy()[z0:z1])
    full_p = torch.zeros((C, Z, Y, X))
    full_m = torch.zeros((Z, Y, X), dtype=torch.int8)
    full_p[:, z0:z1] = torch.from_numpy(probs)
    full_m[z0:z1] = torch.from_numpy(mask)
    gather_slabs(full_p, full_m, plan)
    # epoch accounting under data parallelism: one full pass = len(dataset) / world samples per rank
    sampler = EpochConcateDistributedSampler(list(range(12)), 3, 0, num_replicas=world, rank=rank, shuffle=False)
    per_rank_pass = len(list(sampler)) // 3
    steps_per_pass = per_rank_pass // 2                              # batchsize 2
    torch.save({'probs': full_p, 'mask': full_m, 'moved': sum((b - a) for q, r, a, b in plan.transfers()),
                'epoch_after_pass': epoch_of_batch(steps_per_pass, 2, 12, world), 'bounds': plan.bounds},
               out.format(rank))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize('world', [2, 3])
def test_sharded_inference_slab_merge_equals_single_rank(tmp_path, world):
    """SlabShardPlan + merge_slabs + gather_slabs on `world` gloo ranks reproduce the single-process accumulation:
    overlap counts exactly, averaged probabilities to rounding, the arg-max mask, on EVERY rank; and only halo planes move"""
    import numpy as np
    from oracle import numpy_ref
    port, out = _free_port(), str(tmp_path / 'r{}.pt')
    mp.spawn(_infer_worker, args=(world, port, out), nprocs=world, join=True)
    (Z, Y, X, C), starts, ends, patch = _infer_case()
    acc = np.zeros((C, Z, Y, X), np.float32)
    cnt = np.zeros((Z, Y, X), np.float32)
    for k in range(len(starts)):
        numpy_ref.accumulate_patch(acc, cnt, starts[k], ends[k], patch(k))
    assert cnt.max() == 8.0
    rp, rm = numpy_ref.finalize(acc, cnt)
    for rank in range(world):
        got = torch.load(out.format(rank), weights_only=True)
        assert float(np.abs(got['probs'].numpy() - rp).max()) < 1e-6
        assert float(np.mean(got['mask'].numpy() != rm)) < 1e-4
        assert got['bounds'][0] == 0 and got['bounds'][-1] == Z and got['bounds'] == sorted(got['bounds'])
        assert 0 < got['moved'] <= (world - 1) * 16 < Z             # halo planes only (box - stride per cut at most box)
        assert got['epoch_after_pass'] == 1                          # (the reference's formula would say 0 at world 2)


def test_slab_plan_covers_every_patch_once():
    from segmentation3d.core.seg_infer import SlabShardPlan
    from segmentation3d.utils.image_tools import image_partition_by_fixed_size
    starts, _ = image_partition_by_fixed_size(((512, 512, 400), (1.0, 1.0, 1.0)), [0, 0, 0], [512, 512, 400], [96] * 3,
                                              [48] * 3, 16)
    for world in (1, 2, 4, 8, 13):
        plan = SlabShardPlan(starts, (96, 96, 96), (400, 512, 512), world)
        seen = sorted(k for r in range(world) for k in plan.patches[r])
        assert seen == list(range(800))
        assert max(len(p) for p in plan.patches) - min(len(p) for p in plan.patches) <= 1
        moved = sum(b - a for _, _, a, b in plan.transfers())
        assert moved <= (world - 1) * 96 * 2                           # vs world * 400 planes for a full all-reduce
        for q, r, a, b in plan.transfers():
            assert q < r and plan.owned(r)[0] <= a < b <= plan.owned(r)[1]


def test_inference_patch_sharding_is_a_partition():
    from segmentation3d.core.seg_infer import shard_batches
    batches = [list(range(i, min(i + 8, 803))) for i in range(0, 803, 8)]
    seen = []
    for rank in range(4):
        for b in shard_batches(batches, rank, 4):
            seen.extend(b)
    assert sorted(seen) == list(range(803))
    assert shard_batches(batches, 0, 1) == batches
