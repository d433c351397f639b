"""Index streams of the reference's DistributedSemiBalanceSampler (container only): semi_sampler.py loaded by path with
mmcv.runner.get_dist_info stubbed.  -> tests/golden/sampler.npz"""
import importlib.util
import os
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
from tests import common as C  # noqa: E402

REF = os.environ.get('S4F_REFERENCE_DIR', '/root/reference')


def main():
    runner = types.ModuleType('mmcv.runner')
    runner.get_dist_info = lambda: (0, 1)
    mm = types.ModuleType('mmcv')
    mm.runner = runner
    sys.modules['mmcv'], sys.modules['mmcv.runner'] = mm, runner
    spec = importlib.util.spec_from_file_location('ref_sampler', os.path.join(REF, 'mmseg/datasets/samplers/semi_sampler.py'))
    M = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(M)

    class DS:
        def __init__(self, cs):
            self.cumulative_sizes = cs
    out = {}
    for name, kw in C.SAMPLER_CASES.items():
        kw = dict(kw)
        cs = kw.pop('cumulative_sizes')
        epochs = kw.pop('epochs')
        for ep in epochs:
            for rank in range(kw['num_replicas']):
                s = M.DistributedSemiBalanceSampler(DS(list(cs)), rank=rank, **kw)
                s.set_epoch(ep)
                out[f'{name}_e{ep}_r{rank}'] = np.array(list(iter(s)), dtype=np.int64)
        print(name, 'ok', len(out))
    np.savez_compressed(os.path.join(HERE, 'sampler.npz'), **out)


if __name__ == '__main__':
    main()
