"""DistributedSemiBalanceSampler: index streams identical to the reference's (goldens made by its own semi_sampler.py), every
batch in the configured labelled / unlabelled proportion, ranks disjoint in batch units."""
import os

import numpy as np
import pytest

from s4former_amd.sampler import DistributedSemiBalanceSampler
from tests import common as C

GOLD = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'sampler.npz'))


@pytest.mark.parametrize('name', sorted(C.SAMPLER_CASES))
def test_index_stream_matches_the_reference(name):
    kw = dict(C.SAMPLER_CASES[name])
    cs, epochs = kw.pop('cumulative_sizes'), kw.pop('epochs')
    for ep in epochs:
        for rank in range(kw['num_replicas']):
            s = DistributedSemiBalanceSampler(list(cs), rank=rank, **kw)
            s.set_epoch(ep)
            got = np.array(list(iter(s)), dtype=np.int64)
            assert np.array_equal(got, GOLD[f'{name}_e{ep}_r{rank}']), (name, ep, rank)
            assert len(got) == len(s) == kw['max_iter_size'] * kw['samples_per_gpu']
            spg = kw['samples_per_gpu']
            n_sup = int(kw['sample_ratio'][0] / sum(kw['sample_ratio']) * spg)
            for b in got.reshape(-1, spg):
                assert (b[:n_sup] < cs[0]).all() and (b[n_sup:] >= cs[0]).all() and (b < cs[1]).all()
