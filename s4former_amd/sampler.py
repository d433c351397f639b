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

    def __iter__(self):
        g = torch.Generator()
        g.manual_seed(self.epoch)
        bounds = [0] + self.cumulative_sizes
        per_dataset = []
        for i in range(len(self.cumulative_sizes)):
            per_dataset.append(np.array(range(bounds[i], bounds[i + 1])))
            # (the reference re-draws the permutation of EVERY dataset collected so far in each pass of this loop and keeps the
            # last pass: dataset 0 is permuted twice with two datasets, and the generator advances accordingly)
            shuffled = [s[list(torch.randperm(int(s.shape[0]), generator=g).numpy())] for s in per_dataset]
        total = []
        batch_idx = 0
        while batch_idx < self.max_iter_size * self.num_replicas:
            ratio = [x / sum(self.sample_ratio) for x in self.sample_ratio]
            ratio = [int(r * self.samples_per_gpu) for r in ratio]
            ratio[-1] = self.samples_per_gpu - sum(ratio[:-1])
            selected = []
            for i in range(len(shuffled)):
                if len(shuffled[i]) < ratio[i]:
                    shuffled[i] = np.concatenate(
                        (shuffled[i], per_dataset[i][list(torch.randperm(int(per_dataset[i].shape[0]), generator=g).numpy())]))
                selected.append(shuffled[i][:ratio[i]])
                shuffled[i] = shuffled[i][ratio[i]:]
            total.append(np.concatenate(selected))
            batch_idx += 1
        indices = np.concatenate(total)
        spg = self.samples_per_gpu
        indices = [indices[j] for i in list(torch.randperm(len(indices) // spg, generator=g)) for j in range(i * spg, (i + 1) * spg)]
        offset = len(self) * self.rank
        indices = indices[offset:offset + len(self)]
        assert len(indices) == len(self)
        return iter(indices)

    def __len__(self):
        return self.max_iter_size * self.samples_per_gpu

    def set_epoch(self, epoch):
        self.epoch = epoch
