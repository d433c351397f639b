"""GPU: evaluation path through the C ABI - the metric kernel against the reference's golden answers (exact
integer pixel counts), resize / softmax / arg-max kernels and the whole-image inference of the product against the oracle."""
import os

import numpy as np
import pytest
import torch

import s4former_amd as S
from oracle import model as OM
from oracle import ops as O
from s4former_amd import metrics as M
from tests import common as C

pytestmark = pytest.mark.gpu
GOLD = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'eval_metrics.npz'))


@pytest.mark.parametrize('name', sorted(C.METRIC_CASES))
def test_metrics_match_the_reference_golden(name):
    ncls, rz = C.METRIC_CASES[name]
    preds, labels = C.metric_maps(name)
    tot = M.total_intersect_and_union(preds, labels, ncls, 255, dict(), rz)
    assert np.array_equal(np.stack([t.numpy() for t in tot]), GOLD[f'{name}_areas']), 'pixel counts must be exact'
    ret = M.eval_metrics(preds, labels, ncls, 255, metrics=['mIoU', 'mDice', 'mFscore'], reduce_zero_label=rz)
    for k in ('aAcc', 'IoU', 'Acc', 'Dice', 'Fscore', 'Precision', 'Recall'):
        np.testing.assert_allclose(ret[k], GOLD[f'{name}_{k}'], rtol=1e-12, equal_nan=True, err_msg=k)
    one = M.intersect_and_union(preds[0], labels[0], ncls, 255, dict(), rz)
    ref = O.intersect_and_union(preds[0], labels[0], ncls, 255, None, rz)
    for a, b in zip(one, ref):
        assert torch.equal(a, b.double())
    m1 = M.mean_iou(preds, labels, ncls, 255, nan_to_num=-1, reduce_zero_label=rz)
    assert not np.isnan(m1['IoU']).any()


@pytest.mark.parametrize('align', [False, True])
@pytest.mark.parametrize('size', [(37, 53), (64, 64), (200, 150), (1, 1)])
def test_resize_bilinear_matches_interpolate(size, align):
    from s4former_amd import kernels as K
    x = torch.randn(2, 5, 48, 64, generator=torch.Generator().manual_seed(1))
    for window in (None, (40, 50)):
        xs = x if window is None else x[:, :, :window[0], :window[1]]
        ref = O.resize(xs, size, align)
        got = K.resize_bilinear(x.cuda(), size, align, window=window).cpu()
        assert got.shape == ref.shape
        # (where a source coordinate lands within rounding of an integer the two index computations may pick neighbouring
        # cells; the interpolant is continuous, the values then differ by ~1e-6 of the slope)
        assert float((got - ref).abs().max()) <= 1e-5 * float(ref.abs().max()), (size, align, window)
    same = K.resize_bilinear(x.cuda(), (48, 64), align).cpu()
    assert torch.equal(same, x), 'resize to the same size is the identity'


@pytest.mark.parametrize('flip', [None, 'horizontal', 'vertical'])
def test_softmax_flip_argmax(flip):
    from s4former_amd import kernels as K
    z = torch.randn(2, 21, 33, 47, generator=torch.Generator().manual_seed(2)) * 3
    z[0, :, 0, 0] = 1.5                                      # all classes tie -> index 0
    z[0, 3, 0, 1] = z[0, 9, 0, 1] = 20.0                     # tie between 3 and 9 -> 3
    prob_ref, lab_ref = O.whole_inference_post(z, (33, 47), (33, 47), flip)
    prob, lab, pmax = K.softmax_argmax(z.cuda(), True, {None: 0, 'horizontal': 1, 'vertical': 2}[flip])
    assert float((prob.cpu() - prob_ref).abs().max()) < 1e-6
    margin = prob_ref.topk(2, dim=1).values
    tie = (margin[:, 0] - margin[:, 1]) < 1e-6               # the decision is taken by exp's last bit: not pinned
    crafted = torch.zeros_like(tie)
    for (y, x) in ((0, 0), (0, 1)):
        yy, xx = (y, 46 - x) if flip == 'horizontal' else ((32 - y, x) if flip == 'vertical' else (y, x))
        crafted[0, yy, xx] = True
        assert int(lab[0, yy, xx]) == int(lab_ref[0, yy, xx]) == (0 if x == 0 else 3)
    assert torch.equal(lab.cpu().long()[~tie | crafted], lab_ref[~tie | crafted])
    assert float((pmax.cpu() - prob_ref.max(1).values).abs().max()) < 1e-6
    # raw mode: arg-max of given probabilities (aug_test's mean)
    _, lab2, _ = K.softmax_argmax(prob_ref.cuda().contiguous(), False, 0, raw=True)
    assert torch.equal(lab2.cpu().long(), prob_ref.argmax(1))


@pytest.mark.parametrize('dtype,ema', [('fp32', False), ('fp32', True), ('bf16', False)])
def test_whole_image_inference_vs_oracle(dtype, ema):
    """forward(return_loss=False) / simple_test / aug_test of the product against the oracle: padded input (img_shape smaller than
    the tensor), rescale to a different ori_shape, flipped test image; labels equal outside near-ties; mIoU of both agree"""
    S.set_compute_dtype(dtype)
    try:
        cfg = C.tiny_model_cfg(unsup_weight=1.0, ema_test=ema)
        model = S.build_segmentor(cfg)
        vals = C.load_filled(model, 1999, 5.0)
        model.cuda().eval()
        ocfg = dict(cfg)
        ocfg.pop('ema_test')
        orc = OM.oracle_from_cfg(ocfg)
        orc.load_state_dict(vals, strict=True)
        imgs, gt, _ = C.make_batch(77, 2, 0)
        results, refs = [], []
        for flip in (None, 'horizontal'):
            meta = [dict(img_shape=(60, 56, 3), ori_shape=(83, 71, 3), pad_shape=(64, 64, 3), flip=flip is not None,
                         flip_direction=flip or 'horizontal') for _ in range(2)]
            out = model(img=[imgs.cuda()], img_metas=[meta], return_loss=False)
            prob_ref, lab_ref = orc.simple_test(imgs, (60, 56), (83, 71), flip, ema=ema)
            assert len(out) == 2 and out[0].shape == (83, 71) and out[0].dtype == np.int64
            got = np.stack(out)
            top2 = prob_ref.topk(2, dim=1).values
            if dtype == 'fp32':
                near = ((top2[:, 0] - top2[:, 1]) < 1e-4).numpy()
                bad = (got != lab_ref.numpy()) & ~near
                assert not bad.any(), f'{int(bad.sum())} labels differ outside near-ties (flip {flip})'
                assert near.mean() < 0.02
            else:
                assert (got != lab_ref.numpy()).mean() < 0.1, 'bf16 perf mode: at least 90 % of the labels agree'
            prob = model.inference(imgs.cuda(), meta, True).cpu()
            assert float((prob - prob_ref).abs().max()) < (1e-4 if dtype == 'fp32' else 5e-2)
            if dtype == 'fp32' and ema:      # the reference's OWN whole_inference output (its ema_test path runs as written)
                gold = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'eval_slide.npz'))
                tag = 'whole_flip' if flip else 'whole_plain'
                bad = (got != gold[f'{tag}_label']) & (gold[f'{tag}_margin'] >= 1e-4)
                assert not bad.any(), f'{int(bad.sum())} labels differ from the reference outside near-ties'
                assert float(np.abs(prob.max(1).values.numpy() - gold[f'{tag}_pmax']).max()) < 1e-4
            results += out
            refs += list(lab_ref.numpy())
        # test-time augmentation: mean of the two probability maps
        metas = [[dict(img_shape=(60, 56, 3), ori_shape=(83, 71, 3), pad_shape=(64, 64, 3), flip=f, flip_direction='horizontal')
                  for _ in range(2)] for f in (False, True)]
        aug = model(img=[imgs.cuda(), imgs.cuda()], img_metas=metas, return_loss=False)
        p0, _ = orc.simple_test(imgs, (60, 56), (83, 71), None, ema=ema)
        p1, _ = orc.simple_test(imgs, (60, 56), (83, 71), 'horizontal', ema=ema)
        mean = (p0 + p1) / 2
        t2 = mean.topk(2, dim=1).values
        if dtype == 'fp32':
            ok = ((t2[:, 0] - t2[:, 1]) >= 1e-4).numpy()
            assert np.array_equal(np.stack(aug)[ok], mean.argmax(1).numpy()[ok])
        else:
            assert (np.stack(aug) != mean.argmax(1).numpy()).mean() < 0.1
        # metric of the product's predictions == metric of the oracle's, up to the near-tie pixels
        gts = [np.random.RandomState(5 + i).randint(0, 21, size=(83, 71)).astype(np.uint8) for i in range(4)]
        a = M.mean_iou(results, gts, 21, 255, nan_to_num=0)['aAcc']
        b = O.mean_iou(refs, gts, 21, 255)[0]['aAcc']
        assert abs(float(a) - float(b)) < (2e-3 if dtype == 'fp32' else 0.2)
    finally:
        S.set_compute_dtype('fp32')


@pytest.mark.parametrize('dtype,ema', [('fp32', False), ('fp32', True), ('bf16', False)])
def test_slide_inference_vs_oracle(dtype, ema):
    """test_cfg.mode='slide' (encoder_decoder.py:1068-1116): 64 x 64 windows at stride (32, 48) over a 96 x 112 input - 2 x 2
    windows, the last of each row / column shifted back inside the image, overlaps averaged - padded area removed, rescaled"""
    S.set_compute_dtype(dtype)
    try:
        cfg = C.tiny_model_cfg(unsup_weight=1.0, ema_test=ema)
        cfg['test_cfg'] = dict(mode='slide', crop_size=(64, 64), stride=(32, 48))
        model = S.build_segmentor(cfg)
        vals = C.load_filled(model, C.SLIDE_CASE['seed_w'], C.SLIDE_CASE['gain'])
        model.cuda().eval()
        ocfg = dict(cfg)
        ocfg.pop('ema_test')
        orc = OM.oracle_from_cfg(ocfg)
        orc.load_state_dict(vals, strict=True)
        imgs = C.slide_input()
        gold = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'eval_slide.npz'))
        for flip in (None, 'horizontal'):
            meta = [dict(img_shape=(90, 100, 3), ori_shape=(120, 131, 3), pad_shape=(96, 112, 3), flip=flip is not None,
                         flip_direction=flip or 'horizontal') for _ in range(2)]
            out = model(img=[imgs.cuda()], img_metas=[meta], return_loss=False)
            prob_ref, lab_ref = orc.slide_test(imgs, (90, 100), (120, 131), (64, 64), (32, 48), flip, ema=ema)
            assert len(out) == 2 and out[0].shape == (120, 131)
            got = np.stack(out)
            prob = model.inference(imgs.cuda(), meta, True).cpu()
            if dtype == 'fp32':
                top2 = prob_ref.topk(2, dim=1).values
                near = ((top2[:, 0] - top2[:, 1]) < 1e-4).numpy()
                assert not ((got != lab_ref.numpy()) & ~near).any()
                assert float((prob - prob_ref).abs().max()) < 1e-4
                if ema:      # the reference's own slide_inference output (its ema_test path runs as written)
                    tag = 'flip' if flip else 'plain'
                    bad = (got != gold[f'{tag}_label']) & (gold[f'{tag}_margin'] >= 1e-4)
                    assert not bad.any(), f'{int(bad.sum())} labels differ from the reference outside near-ties'
                    assert float(np.abs(prob.max(1).values.numpy() - gold[f'{tag}_pmax']).max()) < 1e-4
            else:
                assert (got != lab_ref.numpy()).mean() < 0.1
                assert float((prob - prob_ref).abs().max()) < 5e-2
    finally:
        S.set_compute_dtype('fp32')
