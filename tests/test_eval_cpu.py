"""CPU: the oracle's restatement of the metric code against answers produced by the reference's own metrics.py
(tests/golden/make_golden_eval.py -> tests/golden/eval_metrics.npz)."""
import os

import numpy as np
import pytest

from oracle import ops as O
from tests import common as C

GOLD = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'eval_metrics.npz'))


@pytest.mark.parametrize('name', sorted(C.METRIC_CASES))
def test_oracle_metrics_match_the_reference(name):
    ncls, rz = C.METRIC_CASES[name]
    preds, labels = C.metric_maps(name)
    ret, tot = O.mean_iou(preds, labels, ncls, 255, rz)
    assert np.array_equal(np.stack([t.numpy() for t in tot]), GOLD[f'{name}_areas'])      # pixel counts: exact
    for k in ('aAcc', 'IoU', 'Acc'):
        np.testing.assert_allclose(ret[k], GOLD[f'{name}_{k}'], rtol=1e-12, equal_nan=True)
