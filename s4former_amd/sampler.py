"""DistributedSemiBalanceSampler: the index stream of the reference's semi-supervised data loader
(mmseg/datasets/samplers/semi_sampler.py:9-138), host logic only.

Every "batch" of `samples_per_gpu` indices holds the datasets in the fixed proportion `sample_ratio` (the SETR configs:
4 labelled + 4 unlabelled per GPU, configs/setr/*:31-33); each dataset is walked through a random permutation seeded by the
epoch and re-shuffled when exhausted; the batches of all replicas are then shuffled as units and rank r takes batches
[r * max_iter_size, (r + 1) * max_iter_size).  The generator calls (torch.randperm on ONE torch.Generator seeded with the
epoch) happen in the reference's order, so the stream is identical index for index."""
import numpy as np
import torch


class DistributedSemiBalanceSampler(torch.utils.data.Sampler):
    def __init__(self, dataset, by_prob=False, max_iter_size=None, sample_ratio=None, samples_per_gpu=1, num_replicas=None,
                 rank=None, **kwargs):
        assert samples_per_gpu > 1, 'samples_per_gpu should be greater than 1.'
        if num_replicas is None or rank is None:
            import torch.distributed as dist
            on = dist.is_available() and dist.is_initialized()
            num_replicas = (dist.get_world_size() if on else 1) if num_replicas is None else num_replicas
            rank = (dist.get_rank() if on else 0) if rank is None else rank
        self.dataset = dataset
        self.samples_per_gpu, self.num_replicas, self.rank = samples_per_gpu, num_replicas, rank
        self.epoch = 0
        self.by_prob = by_prob
        self.max_iter_size = max_iter_size
        self.num_samples = 0
        # sizes of [labelled, labelled + unlabelled] (ConcatDataset.cumulative_sizes); a plain list is accepted too
        self.cumulative_sizes = list(getattr(dataset, 'cumulative_sizes', dataset))
        if not isinstance(sample_ratio, list):
            sample_ratio = [sample_ratio] * len(self.cumulative_sizes)
        self.sample_ratio = [int(sr / min(sample_ratio)) for sr in sample_ratio]
        self.size_of_dataset = []
        for i in range(len(self.cumulative_sizes)):
            size = np.ceil(self.cumulative_sizes[i] / self.sample_ratio[i])
            self.size_of_dataset.append(int(np.ceil(size / self.samples_per_gpu / self.num_replicas)) * self.samples_per_gpu)
        for j in range(len(self.cumulative_sizes)):
            self.num_samples += self.size_of_dataset[-1] * self.sample_ratio[j]
        self.total_size = self.num_samples * self.num_replicas

    def _quota(self):
        """samples of every dataset in one batch of `samples_per_gpu` (the last dataset takes the remainder)"""
        total = sum(self.sample_ratio)
        q = [int(sr / total * self.samples_per_gpu) for sr in self.sample_ratio]
        q[-1] = self.samples_per_gpu - sum(q[:-1])
        return q

    def _epoch_stream(self):
        """all replicas' batches of one epoch, flattened; the order of the generator calls is the reference's"""
        gen = torch.Generator()
        gen.manual_seed(self.epoch)

        def shuffled(ids):
            return ids[torch.randperm(len(ids), generator=gen).numpy()]

        edges = [0] + self.cumulative_sizes
        pools, queues = [], []
        for d in range(len(self.cumulative_sizes)):
            pools.append(np.arange(edges[d], edges[d + 1]))
            # the reference re-draws the permutation of EVERY dataset seen so far in each pass of this loop and keeps the last
            # pass (with two datasets, dataset 0 is permuted twice): the generator has to advance the same way
            queues = [shuffled(pool) for pool in pools]
        quota = self._quota()
        batches = []
        for _ in range(self.max_iter_size * self.num_replicas):
            parts = []
            for d, need in enumerate(quota):
                if len(queues[d]) < need:                      # exhausted: append a fresh permutation of the whole dataset
                    queues[d] = np.concatenate((queues[d], shuffled(pools[d])))
                parts.append(queues[d][:need])
                queues[d] = queues[d][need:]
            batches.append(np.concatenate(parts))
        order = torch.randperm(len(batches), generator=gen).tolist()     # the batches travel as units
        return np.concatenate([batches[b] for b in order])

    def __iter__(self):
        stream = self._epoch_stream()
        mine = stream[len(self) * self.rank:len(self) * (self.rank + 1)]
        assert len(mine) == len(self)
        return iter(mine.tolist())

    def __len__(self):
        return self.max_iter_size * self.samples_per_gpu

    def set_epoch(self, epoch):
        self.epoch = epoch
