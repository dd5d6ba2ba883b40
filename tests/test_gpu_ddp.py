"""GPU rehearsal of the data-parallel train step: two ranks share the one test GPU (gloo backend; RCCL refuses two
ranks on one device), each with its own sample; the all-reduced flat gradient equals the single-process gradient of
the two-sample batch, and both ranks end the step with identical parameters."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _data():
    from oracle import detgen
    x = torch.from_numpy(detgen.normal(81, 'ddp/x', (2, 1, 32, 32, 32)))
    t = torch.from_numpy(detgen.labels(82, 'ddp/t', (2, 1, 32, 32, 32), 2))
    return x, t


def _worker(rank, world, port, out, mode):
    from conftest import PKG  # noqa: F401  (sys.path)
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    torch.cuda.set_device(0)
    from segmentation3d.core.seg_train import TrainStep
    from segmentation3d import _ops
    _ops.set_activation_dtype(mode)
    step = TrainStep('vnet', 1, 2, 'Dice', [0.5, 0.5], device=torch.device('cuda:0'), seed=rank)  # different init per rank
    x, t = _data()
    dev = step.device
    loss = step(x[rank:rank + 1].to(dev), t[rank:rank + 1].to(dev))
    torch.cuda.synchronize()
    flat = step.opt._flat[0]
    torch.save({'grads': flat['grads'].cpu(), 'params': flat['params'].cpu(), 'loss': float(loss),
                'buckets': step.reducer.bucket_sizes(), 'overlapped': step.reducer.launched_in_backward},
               out.format(rank))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize('mode', ['fp32', 'bf16'])
def test_two_rank_train_step_matches_global_batch(hip_device, tmp_path, mode):
    """(bf16 mode: the gradients that are all-reduced are fp32 either way; the reference run below uses the same mode)"""
    world, port, out = 2, _free_port(), str(tmp_path / 'rank{}.pt')
    mp.spawn(_worker, args=(world, port, out, mode), nprocs=world, join=True)
    r0 = torch.load(out.format(0), weights_only=True)
    r1 = torch.load(out.format(1), weights_only=True)
    assert torch.equal(r0['params'], r1['params'])          # broadcast at start + identical reduced gradients
    assert torch.equal(r0['grads'], r1['grads'])
    assert len(r0['buckets']) == 4
    # every bucket's all-reduce was enqueued from a gradient-ready hook DURING backward (the kernels write the
    # gradients straight into the flat buffer; autograd still fires the leaf hooks)
    assert r0['overlapped'] == 4 and r1['overlapped'] == 4
    from segmentation3d.core.seg_train import TrainStep
    from segmentation3d import _ops
    ref = TrainStep('vnet', 1, 2, 'Dice', [0.5, 0.5], device=hip_device, seed=0, distributed=False)
    x, t = _data()
    ref.opt.zero_grad()
    with _ops.activation_dtype(mode):
        loss = ref.loss_func(ref.net(x.to(hip_device)), t.to(hip_device))
        loss.backward()
    g_ref = ref.opt._flat[0]['grads'].cpu()
    g_ddp = r0['grads'] / 2.0                                # buffer holds the SUM over ranks
    rel = float((g_ddp - g_ref).abs().max() / g_ref.abs().max())
    if mode == 'fp32':
        assert abs(0.5 * (r0['loss'] + r1['loss']) - float(loss)) < 1e-5
        assert rel < 2e-2, rel
    else:
        # batch 1 per rank and batch 2 in one process pick different tile plans, i.e. another fp32 summation order; in
        # bf16 mode such a 1e-7 difference can flip the rounding of an activation (4e-3 of its value), and this random-init
        # net amplifies that (tests/test_gpu_bf16.py): compare the loss to 1e-3 and the gradient by direction
        cos = float((g_ddp.double() * g_ref.double()).sum() / (g_ddp.double().norm() * g_ref.double().norm()))
        assert abs(0.5 * (r0['loss'] + r1['loss']) - float(loss)) < 1e-3
        assert cos > 0.98, (cos, rel)
