"""Index samplers with the reference's class names and semantics (dataloader/sampler.py:6-78): a training run draws
ONE long index stream made of `epoch` consecutive passes over the data set, each pass in its own random order.

All three classes are thin policies over `_concat_passes`, which differ only in how the order of pass `i` is produced:
  EpochConcateSampler             python's global `random` stream (as seeded by the caller)
  EpochConcateSamplerResume       `random.seed(i)` before pass i, so a run resumed at epoch r replays passes r, r+1, ...
  EpochConcateDistributedSampler  torch's DistributedSampler permutation of epoch i, cut to this rank's shard
"""
import random

from torch.utils.data.distributed import DistributedSampler
from torch.utils.data.sampler import Sampler


def _concat_passes(first_pass, num_passes, order_of_pass):
    stream = []
    for i in range(first_pass, first_pass + num_passes):
        stream.extend(order_of_pass(i))
    return stream


class EpochConcateSampler(Sampler):
    def __init__(self, data_source, epoch):
        self.data_length, self.epoch = len(data_source), epoch

    def _order(self, _i):
        order = list(range(self.data_length))
        random.shuffle(order)
        return order

    def __iter__(self):
        return iter(_concat_passes(0, self.epoch, self._order))

    def __len__(self):
        return self.epoch * self.data_length


class EpochConcateSamplerResume(EpochConcateSampler):
    def __init__(self, data_source, epoch, resume_epoch):
        super(EpochConcateSamplerResume, self).__init__(data_source, epoch)
        self.resume_epoch = resume_epoch

    def _order(self, i):
        random.seed(i)                      # pass i is reproducible on its own
        return super(EpochConcateSamplerResume, self)._order(i)

    def __iter__(self):
        return iter(_concat_passes(self.resume_epoch, self.epoch, self._order))


class EpochConcateDistributedSampler(DistributedSampler):
    def __init__(self, data_source, epoch, resume_epoch=0, **kwargs):
        super(EpochConcateDistributedSampler, self).__init__(data_source, **kwargs)
        self.data_length, self.epoch, self.resume_epoch = len(data_source), epoch, resume_epoch

    def _shard_of_pass(self, i):
        self.set_epoch(i)
        return list(DistributedSampler.__iter__(self))

    def __iter__(self):
        return iter(_concat_passes(self.resume_epoch, self.epoch, self._shard_of_pass))

    def __len__(self):
        return self.epoch * DistributedSampler.__len__(self)
