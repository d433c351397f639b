"""Golden answers of the reference's metric code (container only): mmseg/core/evaluation/metrics.py loaded by path, fed with
seeded random prediction / label maps (incl. ignore 255, out-of-range predictions, reduce_zero_label).  Writes
tests/golden/eval_metrics.npz (outputs only: the inputs are regenerated from the seed by tests/common.metric_maps)."""
import importlib.util
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
from oracle import ref_harness as RH  # noqa: E402
from tests import common as C  # noqa: E402


def main():
    assert RH.available()
    RH.load_reference()                                     # installs the mmcv stand-in that metrics.py imports
    spec = importlib.util.spec_from_file_location('ref_metrics', os.path.join(RH.REF, 'mmseg/core/evaluation/metrics.py'))
    M = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(M)
    out = {}
    for name, (ncls, rz) in C.METRIC_CASES.items():
        preds, labels = C.metric_maps(name)
        ret = M.eval_metrics(preds, labels, ncls, 255, metrics=['mIoU', 'mDice', 'mFscore'], nan_to_num=None, label_map=dict(),
                             reduce_zero_label=rz)
        tot = M.total_intersect_and_union(preds, labels, ncls, 255, dict(), rz)
        for k, v in ret.items():
            out[f'{name}_{k}'] = np.asarray(v, dtype=np.float64)
        out[f'{name}_areas'] = np.stack([t.numpy() for t in tot])
        print(name, 'mIoU', float(np.nanmean(ret['IoU'])), 'aAcc', float(ret['aAcc']))
    np.savez_compressed(os.path.join(HERE, 'eval_metrics.npz'), **out)
    print('written')


if __name__ == '__main__':
    main()
