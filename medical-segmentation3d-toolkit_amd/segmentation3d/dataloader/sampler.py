"""Index samplers with the reference's names (dataloader/sampler.py:6-78): all epochs' shuffled index lists
concatenated into one stream, optionally resumable (epoch number = shuffle seed) or sharded over ranks."""
import random

from torch.utils.data.distributed import DistributedSampler
from torch.utils.data.sampler import Sampler


class EpochConcateSampler(Sampler):
    """`epoch` independently shuffled passes over the data set, back to back (python `random` stream)"""

    def __init__(self, data_source, epoch):
        self.data_length = len(data_source)
        self.epoch = epoch

    def __iter__(self):
        stream = []
        for _ in range(self.epoch):
            order = list(range(self.data_length))
            random.shuffle(order)
            stream.extend(order)
        return iter(stream)

    def __len__(self):
        return self.data_length * self.epoch


class EpochConcateSamplerResume(Sampler):
    """as above, but pass i is shuffled with random.seed(i) so that training can resume at `resume_epoch`"""

    def __init__(self, data_source, epoch, resume_epoch):
        self.data_length = len(data_source)
        self.epoch = epoch
        self.resume_epoch = resume_epoch

    def __iter__(self):
        stream = []
        for i in range(self.resume_epoch, self.resume_epoch + self.epoch):
            order = list(range(self.data_length))
            random.seed(i)
            random.shuffle(order)
            stream.extend(order)
        return iter(stream)

    def __len__(self):
        return self.data_length * self.epoch


class EpochConcateDistributedSampler(DistributedSampler):
    """every rank's shard of every epoch's permutation, concatenated (one process per GPU)"""

    def __init__(self, data_source, epoch, resume_epoch=0, **kwargs):
        super(EpochConcateDistributedSampler, self).__init__(data_source, **kwargs)
        self.data_length = len(data_source)
        self.epoch = epoch
        self.resume_epoch = resume_epoch

    def __iter__(self):
        stream = []
        for i in range(self.resume_epoch, self.resume_epoch + self.epoch):
            self.set_epoch(i)
            stream.extend(super(EpochConcateDistributedSampler, self).__iter__())
        return iter(stream)

    def __len__(self):
        return super(EpochConcateDistributedSampler, self).__len__() * self.epoch
