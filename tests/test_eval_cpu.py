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


def test_oracle_slide_inference_matches_the_reference_golden():
    """test_cfg.mode='slide' (encoder_decoder.py:1068-1116): the oracle's restatement against what the reference's own
    slide_inference / inference returned (tests/golden/make_golden_slide.py; its ema_test path, which runs as written)"""
    import json

    import torch

    from oracle import model as OM
    z = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'eval_slide.npz'))
    S = C.SLIDE_CASE
    imgs = C.slide_input()
    assert C.sha(imgs) == json.loads(str(z['meta']))['input_sha']
    cfg = C.tiny_model_cfg(unsup_weight=1.0)
    orc = OM.oracle_from_cfg(cfg)
    C.load_filled(orc, S['seed_w'], S['gain'])
    for tag, flip in (('plain', None), ('flip', 'horizontal')):
        prob, lab = orc.slide_test(imgs, S['img_shape'], S['ori_shape'], S['crop'], S['stride'], flip, ema=True)
        assert np.array_equal(lab.numpy().astype(np.uint8), z[f'{tag}_label'])
        assert float((prob.max(1).values - torch.from_numpy(z[f'{tag}_pmax'])).abs().max()) <= 1e-6


def test_oracle_whole_inference_matches_the_reference_golden():
    """test_cfg.mode='whole' (encoder_decoder.py:1118-1147, 1174-1216): the oracle's simple_test against the reference's own
    whole_inference / inference on its ema_test path (which runs as written; the non-EMA path raises - Q8)"""
    import torch

    from oracle import model as OM
    z = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'eval_slide.npz'))
    S, W = C.SLIDE_CASE, C.WHOLE_CASE
    imgs, _, _ = C.make_batch(W['seed_x'], 2, 0)
    orc = OM.oracle_from_cfg(C.tiny_model_cfg(unsup_weight=1.0))
    C.load_filled(orc, S['seed_w'], S['gain'])
    for tag, flip in (('whole_plain', None), ('whole_flip', 'horizontal')):
        prob, lab = orc.simple_test(imgs, W['img_shape'], W['ori_shape'], flip, ema=True)
        assert np.array_equal(lab.numpy().astype(np.uint8), z[f'{tag}_label'])
        assert float((prob.max(1).values - torch.from_numpy(z[f'{tag}_pmax'])).abs().max()) <= 1e-6
