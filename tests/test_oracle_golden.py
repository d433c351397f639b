"""CPU tests (no GPU): the oracle against the golden vectors generated from the reference's own code, the
reference's known-answer CE tests, and — when /root/reference is present (build container) — the oracle against
the reference itself."""
import json
import math
import os

import numpy as np
import pytest
import torch

from oracle import model as OM
from oracle import ops as O
from oracle import ref_harness as RH
from tests import common as C

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')
SCENARIOS = ['sup', 'mt_literal', 'mt_pasa', 'mt_ours']


def load_gold(name):
    z = np.load(os.path.join(GOLD, f'step_{name}.npz'), allow_pickle=False)
    return z, json.loads(str(z['meta']))


def run_oracle(meta):
    cfg = C.tiny_model_cfg(**meta['flags'])
    orc = OM.oracle_from_cfg(cfg)
    orc.train()
    vals = C.fill_state([(k, tuple(v.shape)) for k, v in orc.state_dict().items()], meta['seed_w'], meta['gain'])
    orc.load_state_dict(vals, strict=True)
    opt = OM.build_optimizer(orc, meta['lr'])
    rec = []
    for it in range(2):
        imgs, gt, metas = C.make_batch(meta['seed_b'] + it, meta['n_sup'], meta['n_unsup'])
        assert C.sha(imgs) == meta['input_sha'][it], 'deterministic input generator drifted'
        OM.set_poly_lr(opt, it)
        opt.zero_grad()
        C.seed_host_rng(meta['seed_b'] + it)       # the in-model augmentations draw from numpy's / torch's global generators
        losses = orc.forward_train(imgs, [m['tag'] for m in metas], gt)
        loss, _ = orc.parse_losses(losses)
        loss.backward()
        rec.append(dict(losses={k: float(v) for k, v in losses.items()}, loss=float(loss),
                        gn={n: float(p.grad.norm()) for n, p in orc.named_parameters() if p.grad is not None},
                        g={n: p.grad.clone() for n, p in orc.named_parameters() if p.grad is not None}))
        opt.step()
    return orc, rec


@pytest.mark.parametrize('name', SCENARIOS)
def test_oracle_matches_golden(name):
    torch.set_num_threads(min(8, os.cpu_count() or 1))
    z, meta = load_gold(name)
    orc, rec = run_oracle(meta)
    for it in range(2):
        keys = [str(k) for k in z[f'it{it}_loss_keys']]
        vals = z[f'it{it}_loss_vals']
        assert sorted(k for k in keys if 'loss' in k) == sorted(k for k in rec[it]['losses'] if 'loss' in k)
        tol = 5e-6 if it == 0 else 1e-3
        for k, v in zip(keys, vals):
            if 'loss' in k:
                assert abs(rec[it]['losses'][k] - v) <= tol * abs(v), (k, rec[it]['losses'][k], v)
        assert abs(rec[it]['loss'] - float(z[f'it{it}_loss'])) <= tol * abs(float(z[f'it{it}_loss']))
        for k, v in zip(z[f'it{it}_gn_keys'], z[f'it{it}_gn_vals']):
            assert abs(rec[it]['gn'][str(k)] - v) <= tol * abs(v) + 1e-12, (str(k), rec[it]['gn'][str(k)], v)
        msgs = []
        # 64 elements of EVERY parameter's gradient (iteration 1 sits behind an SGD step with head lr up to 0.1: summation-order
        # noise of iteration 0 is amplified; single elements move more than norms)
        C.check_grad_samples(z, it, rec[it]['g'], tol if it == 0 else 5 * tol, msgs)
        assert not msgs, msgs
    if 'teacher_label_final' in z.files:
        # pseudo-labels of the last iteration: bit-exact wherever the reference's own decision is not a tie
        imgs, gt, metas = C.make_batch(meta['seed_b'] + 1, meta['n_sup'], meta['n_unsup'])
        n0 = meta['n_sup'] + meta['n_unsup']
        info = orc.teacher_info(imgs[n0:])
        lab = info['hard_seg_label'].to(torch.uint8).numpy()
        ref = z['teacher_label_final']
        bad = (lab != ref) & ~C.fragile_pixels(z, 1e-4 * float(z['teacher_logit_absmax_final']))
        assert not bad.any(), f'{int(bad.sum())} pseudo-label pixels differ outside the tie set'
    if name == 'mt_literal':
        # Q1: the plain mean-teacher config never yields an unsupervised loss
        assert sorted(rec[0]['losses']) == sorted(['decode.loss_ce'] + [f'aux_{i}.loss_ce' for i in range(4)])


def test_reference_ce_known_answers():
    """reference tests/test_models/test_losses/test_ce_loss.py:25-39,199-254 on the oracle"""
    assert abs(float(O.ce_mean_all(torch.tensor([[100., -100.]]), torch.tensor([1]), -100)) - 200.0) < 1e-4
    l = O.ce_mean_all(torch.tensor([[100., -100.]]), torch.tensor([1]), -100, class_weight=torch.tensor([0.8, 0.2]))
    assert abs(float(l) - 40.0) < 1e-4
    pred = torch.full((2, 21, 8, 8), 0.5)
    lab = torch.ones(2, 8, 8, dtype=torch.long)
    lab[:, 0, 0] = 255
    assert abs(float(O.ce_mean_all(pred, lab, 255)) - math.log(21) * 126 / 128) < 1e-5


def test_poly_lr_and_ema():
    assert abs(O.poly_lr(0.01, 0, 80001) - 0.01) < 1e-12
    assert abs(O.poly_lr(0.01, 80000, 80001) - ((0.01 - 1e-4) * (1 / 80001) ** 0.9 + 1e-4)) < 1e-15
    t, s = torch.ones(5), torch.zeros(5)
    O.ema_update(t, s, 0.999)
    assert torch.allclose(t, torch.full((5,), 0.999))


@pytest.mark.skipif(not RH.available(), reason='reference tree not present (GPU box)')
def test_oracle_matches_reference_live():
    """container-only: one supervised + PASA step of the reference's own code against the oracle"""
    z, meta = load_gold('mt_pasa')
    cfg = C.tiny_model_cfg(**meta['flags'])
    ref = RH.build_reference_segmentor(cfg)
    ref.train()
    vals = C.load_filled(ref, meta['seed_w'], meta['gain'])
    orc = OM.oracle_from_cfg(cfg)
    orc.train()
    orc.load_state_dict(vals, strict=True)
    imgs, gt, metas = C.make_batch(meta['seed_b'], meta['n_sup'], meta['n_unsup'])
    import tempfile
    cwd = os.getcwd()
    with tempfile.TemporaryDirectory() as td:
        os.chdir(td)
        try:
            rl = ref.forward_train(imgs, metas, gt_semantic_seg=gt, iter=0)
        finally:
            os.chdir(cwd)
    ol = orc.forward_train(imgs, [m['tag'] for m in metas], gt)
    assert set(k for k in rl if 'loss' in k) == set(k for k in ol if 'loss' in k)
    for k in ol:
        assert abs(float(rl[k]) - float(ol[k])) <= 5e-6 * abs(float(rl[k])), k
