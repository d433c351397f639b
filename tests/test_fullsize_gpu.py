"""Full-size properties (BASELINE cfg3/4: DeiT-B 512x512, 8 + 8 images, 21 classes; cfg5: 768x768, 4 + 4 images, 19 classes,
2305 tokens), where the oracle would take minutes per step:

* the bf16 perf path (gemm2 / gemm5 / gemm6 variants picked by the shipped table at exactly the production shapes) agrees
  with the fp32 parity path (gemm.hip, exact fp32 MFMA chain) on the same weights and batch: every loss term, the
  confident-pixel ratio, the gradient arena;
* EMA linearity on the whole arena: teacher' = m teacher + (1 - m) student, to fp32 rounding;
* the first SGD step moves the parameters against their gradients.
"""
import os
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def _one_step(dtype, gain, n_sup, n_unsup, img, ncls):
    import bench
    import s4former_amd as S
    from s4former_amd.functional import join_side_streams
    from s4former_amd.presets import MAX_ITERS, OPTIMIZER, setr_pup_model, synthetic_batch
    dev = torch.device('cuda', 0)
    S.set_compute_dtype(dtype)
    torch.manual_seed(1999)
    flags = dict(unsup_weight=1.0, plain_mt_pseudo_loss=True) if n_unsup else dict(unsup_weight=0)
    model = S.build_segmentor(setr_pup_model(img=img, num_classes=ncls, **flags))
    model.init_weights()
    model.train()
    model.to(dev)
    model.log_vars_as_tensors = True
    opt = S.build_optimizer(model, dict(OPTIMIZER))
    sched = S.PolyLR(opt, MAX_ITERS)
    batch = synthetic_batch(1999, n_sup, n_unsup, img=img, num_classes=ncls, device=dev)
    model.ensure_engine(dev)
    model.student_store.mark_dirty()
    if not n_unsup:
        gain = 1.0
    elif gain is None:
        gain, _ = bench.calibrate_teacher(model, batch, n_sup, n_unsup, 0.5)
    else:
        with torch.no_grad():
            model.decode_head_ema.conv_seg.weight.mul_(gain)
        model.teacher_store.mark_dirty()
        model.ensure_engine(dev)
    imgs, gt, metas = batch
    teacher0 = model.teacher_store.flat.clone()
    student0 = model.student_store.flat.clone()
    sched.step(0)
    opt.zero_grad()
    out = model.train_step(dict(img=imgs, img_metas=metas, gt_semantic_seg=gt), opt, iter=0)
    out['loss'].backward()
    join_side_streams()
    torch.cuda.synchronize()
    res = dict(gain=gain,
               losses={k: float(v) for k, v in out['log_vars'].items()},
               mask_ratio=float(model.last_mask_ratio) if n_unsup else None,
               grad=model.student_store.grad.clone(),
               teacher0=teacher0, student0=student0,
               teacher1=model.teacher_store.flat.clone(),
               momentum=float(model.momentum_backbone))
    opt.step(grad_scale=1.0)
    torch.cuda.synchronize()
    res['student1'] = model.student_store.flat.clone()
    del model, opt
    torch.cuda.empty_cache()
    return res


@pytest.fixture(scope='module', params=[(8, 0, 512, 21), (8, 8, 512, 21), (4, 4, 768, 19)], ids=['cfg2_sup8', 'cfg3_512', 'cfg5_768'])
def runs(request):
    import s4former_amd as S
    try:
        f32 = _one_step('fp32', None, *request.param)
        bf16 = _one_step('bf16', f32['gain'], *request.param)
    finally:
        S.set_compute_dtype('fp32')
    return f32, bf16


def test_bf16_step_agrees_with_fp32_step(runs):
    f32, bf16 = runs
    assert set(f32['losses']) == set(bf16['losses'])
    for k, v in f32['losses'].items():
        assert abs(bf16['losses'][k] - v) <= 2e-2 * abs(v) + 1e-3, (k, v, bf16['losses'][k])
    if f32['mask_ratio'] is not None:
        assert 0.3 < f32['mask_ratio'] < 0.7                       # the pseudo-label path is not degenerate
        assert abs(bf16['mask_ratio'] - f32['mask_ratio']) < 0.02
    g0, g1 = f32['grad'].double(), bf16['grad'].double()
    assert torch.isfinite(g1).all()
    cos = float((g0 * g1).sum() / (g0.norm() * g1.norm()))
    ratio = float(g1.norm() / g0.norm())
    print(f'full-size gradient arena: cosine {cos:.5f}, norm ratio {ratio:.4f}')
    from tests import common as C
    C.record('deit_b/bf16_vs_fp32_step', **{f'grad_arena_cosine_{len(f32["losses"])}_losses': cos, f'grad_arena_norm_ratio_{len(f32["losses"])}_losses': ratio})
    assert cos > 0.999 and 0.99 < ratio < 1.01       # measured (cfg2, 8 labelled images): 0.99939, 0.9998; with S4F_RESID=bf16: 0.99872


@pytest.mark.parametrize('which', [0, 1])
def test_ema_is_linear_on_the_whole_arena(runs, which):
    r = runs[which]
    m = r['momentum']
    n = r['teacher0'].numel()            # the teacher arena (backbone + decode head) is the head of the student's layout
    want = m * r['teacher0'].double() + (1.0 - m) * r['student0'][:n].double()
    err = (r['teacher1'].double() - want).abs()
    assert float(err.max()) <= 1e-6 * float(want.abs().max()) + 1e-7


@pytest.mark.parametrize('which', [0, 1])
def test_first_sgd_step_moves_against_the_gradient(runs, which):
    r = runs[which]
    d = (r['student1'] - r['student0']).double()
    g = r['grad'].double()
    moved = (d != 0) & (g != 0)          # (g == 0 and moved: BN running statistics, updated by the forward pass)
    # (weight decay 0, first step: momentum buffer = gradient)  delta = -lr_group * grad, lr_group > 0
    # (an update below half an ulp of the parameter leaves it unchanged: only the moved ones are checked)
    assert float(moved.double().mean()) > 0.5
    assert bool((d[moved] * g[moved] < 0).all())


# ------------------------------------------------------------------------------------------------ goldens from the reference
# Full-size goldens made in the build container FROM THE REFERENCE'S OWN CODE (tests/golden/make_golden_full.py): DeiT-B /
# SETR-PUP 512x512 (2 labelled images = cfg1 / cfg2 shapes; 2 + 2 with PASA = cfg3 / cfg4 shapes), two iterations, and one
# 768x768 / 19-class / 2305-token forward (cfg5 shapes).  Both numeric modes are held against them at the PRODUCTION kernel
# variants (shipped tuning table): fp32 parity mode at north_star's 1e-4, bf16 perf mode at its stated bf16 bounds.
import json  # noqa: E402

import numpy as np  # noqa: E402

GOLD = os.path.join(ROOT, 'tests', 'golden')
TOL = {   # (loss it0, loss it1, grad-norm it0, it1, grad-element it0, it1, final |.|_1, logit bound for the tie set)
    # fp32 (round 5): <= 3 x the values measured at DeiT-B size (profiles/r04_parity_report.json): losses 5e-6 / 9e-6 (the first
    # iteration keeps north_star's own 1e-4), gradient norms <= 3.9e-4 over every fixture and both iterations
    'fp32': (1e-4, 3e-5, 1.2e-3, 1.2e-3, 1e-4, 5e-2, 2e-4, 2e-5),
    # bf16 (round 3): <= 3 x the values measured at DeiT-B size with the bf16 residual stream (profiles/r03_parity_report.json):
    # losses 6.1e-3 / 8.5e-4, gradient norms 1.8e-2 / 1.9e-2, gradient elements 90th percentile 8.8e-2 (bound 0.25; the worst
    # tensor, measured 0.17, may reach 3 x that; the median, measured 5.1e-2, is held to 0.1), pseudo-labels 1.26 % differ (3.5 %)
    'bf16': (1.8e-2, 5e-3, 5.5e-2, 5.5e-2, 2.5e-1, 2.5e-1, 2e-2, 2e-2),
}


def _golden_run(name, dtype):
    import s4former_amd as S
    from tests import common as C
    z = np.load(os.path.join(GOLD, f'{name}.npz'), allow_pickle=False)
    meta = json.loads(str(z['meta']))
    S.set_compute_dtype(dtype)
    model = S.build_segmentor(C.deit_b_cfg(img=meta['img'], num_classes=meta['num_classes'], **meta['flags']))
    model.train()
    C.load_filled(model, meta['seed_w'], meta['gain'])
    model.cuda()
    opt = S.build_optimizer(model, dict(type='SGD', lr=meta['lr'], momentum=0.9, weight_decay=0.0,
                                        paramwise_cfg=dict(custom_keys={'head': dict(lr_mult=10.)})))
    sched = S.PolyLR(opt, 80001)
    rec = []
    bkw = dict(img=meta['img'], num_classes=meta['num_classes'], block=32, border=8)
    for it in range(max(1, meta['iters'])):
        imgs, gt, metas = C.make_batch(meta['seed_b'] + it, meta['n_sup'], meta['n_unsup'], **bkw)
        assert C.sha(imgs) == meta['input_sha'][it], 'deterministic input generator drifted'
        sched.step(it)
        opt.zero_grad()
        C.seed_host_rng(meta['seed_b'] + it)          # what the generator did before the reference's iteration ("ours" augmentations)
        out = model.train_step(dict(img=imgs.cuda(), img_metas=metas, gt_semantic_seg=gt.cuda()), opt, iter=it)
        r = dict(log=out['log_vars'])
        if meta['iters']:
            out['loss'].backward()
            torch.cuda.synchronize()
            named = [(n, p) for n, p in model.named_parameters() if p.requires_grad and p.grad is not None]
            r['gn'] = {n: float(p.grad.norm()) for n, p in named}
            r['g'] = {n: C.grad_sample(p.grad, meta['ns']).clone() for n, p in named}
            opt.step()
        rec.append(r)
    torch.cuda.synchronize()
    info = None
    if meta['n_unsup']:
        imgs, gt, metas = C.make_batch(meta['seed_b'] + max(1, meta['iters']) - 1, meta['n_sup'], meta['n_unsup'], **bkw)
        n0 = meta['n_sup'] + meta['n_unsup']
        with torch.no_grad():
            model.set_eval(True)
            t = model.extract_teacher_info_ema(imgs[n0:].cuda(), metas[n0:])
            model.set_train(True)
        info = dict(label=t['hard_seg_label'].cpu().numpy(), ratio=float(t['conf_count']) / t['hard_seg_label'].numel())
    sd = {k: float(v.double().abs().sum()) for k, v in model.state_dict().items() if v.dtype.is_floating_point}
    del model, opt
    torch.cuda.empty_cache()
    return z, meta, rec, info, sd


@pytest.mark.parametrize('dtype', ['fp32', 'bf16'])
@pytest.mark.parametrize('name', ['full_sup', 'full_pasa', 'full_768', 'full_ours', 'full_sup8', 'full_semi4', 'full_semi8_fwd',
                                  'full_768_semi4_fwd'])
def test_fullsize_step_vs_reference_golden(name, dtype, monkeypatch):
    import s4former_amd as S
    from tests import common as C
    if not os.path.exists(os.path.join(GOLD, f'{name}.npz')):
        pytest.skip(f'{name}.npz not generated')
    if dtype == 'fp32' and name not in ('full_sup', 'full_sup8'):
        # PASA flags the less-confident half of the patches with torch.topk; at 1024 / 2304 patches the boundary value is tied
        # for these batches, and WHICH tied patch is returned is implementation-defined (CPU nth_element vs GPU radix select).
        # To compare with the CPU reference the tied choice is taken from the CPU implementation (a debugging switch of the
        # product: the default path selects on the device, as the bf16 runs of this test do).
        monkeypatch.setenv('S4F_TOPK_TIES', 'cpu')
    try:
        z, meta, rec, info, sd = _golden_run(name, dtype)
    finally:
        S.set_compute_dtype('fp32')
    tl0, tl1, tg0, tg1, te0, te1, tw, tz = TOL[dtype]
    msgs = []
    from s4former_amd import runtime
    sec = f'deit_b/{name}/{dtype}' + ('' if dtype == 'fp32' else ('/resid_fp32' if runtime._resid_fp32 else '/resid_bf16'))
    for it in range(max(1, meta['iters'])):
        keys = [str(k) for k in z[f'it{it}_loss_keys']]
        assert sorted(k for k in keys if 'loss' in k) == sorted(k for k in rec[it]['log'] if 'loss' in k and k != 'loss')
        lerrs = []
        for k, v in zip(keys, z[f'it{it}_loss_vals']):
            if 'loss' in k:
                e = abs(rec[it]['log'][k] - v) / abs(v)
                lerrs.append(e)
                if e > (tl0, tl1)[it]:
                    msgs.append(f'it{it} {k}: {rec[it]["log"][k]:.6f} vs {v:.6f} (rel {e:.2e})')
        e = abs(rec[it]['log']['loss'] - float(z[f'it{it}_loss'])) / abs(float(z[f'it{it}_loss']))
        C.record(sec, **{f'it{it}_loss_rel_worst': max(lerrs), f'it{it}_loss_rel_median': float(np.median(lerrs)), f'it{it}_total_loss_rel': e})
        if e > (tl0, tl1)[it]:
            msgs.append(f'it{it} total loss rel {e:.2e}')
        if not meta['iters']:
            continue
        gk = [str(k) for k in z[f'it{it}_gn_keys']]
        assert sorted(gk) == sorted(rec[it]['gn']), set(gk) ^ set(rec[it]['gn'])
        worst = max(((abs(rec[it]['gn'][k] - v) / (abs(v) + 1e-12), k) for k, v in zip(gk, z[f'it{it}_gn_vals'])))
        C.record(sec, **{f'it{it}_grad_norm_rel_worst': worst[0], f'it{it}_grad_norm_worst_tensor': worst[1]})
        if worst[0] > (tg0, tg1)[it]:
            msgs.append(f'it{it} grad norm {worst[1]}: rel {worst[0]:.2e}')
        noise = None
        if 'it0_gs64' not in z.files:     # full-batch fixtures: the reference's noise floor from the small-batch fixture of the model
            noise = C.reference_noise(np.load(os.path.join(GOLD, 'full_sup.npz' if not meta['n_unsup'] else 'full_pasa.npz'), allow_pickle=False))
        w = C.check_grad_samples(z, it, rec[it]['g'], (te0, te1)[it], msgs, rec=sec, noise=noise, mtol=0.1 if dtype == 'bf16' else None)
        print(f'{name} {dtype} it{it}: worst grad norm {worst[0]:.2e} ({worst[1]}), grad elements: worst {w[0]:.2e} ({w[1]}), median '
              f'{w[2]:.2e}; reference fp32 vs its fp64: {w[3]:.2e}')
    for k, ref in zip(z['final_sha_keys'], z['final_abs_sum']):
        got = sd[str(k)]
        if abs(got - ref) > tw * abs(ref) + 1e-9:
            msgs.append(f'final |{k}|_1: {got:.6f} vs {ref:.6f}')
    if info is not None:
        ref = z['teacher_label_final']
        mism = info['label'] != ref
        if dtype == 'fp32':
            tol = tz * float(z['teacher_logit_absmax_final'])
            fragile = C.fragile_pixels(z, tol)
            bad = mism & ~fragile
            print(f'{name} fp32 labels: {int(mism.sum())} of {ref.size} differ, {int(fragile.sum())} ties within {tol:.2e}, '
                  f'{int(bad.sum())} outside the tie set')
            C.record(sec, labels_total=int(ref.size), labels_differ=int(mism.sum()), labels_in_tie_set=int(fragile.sum()),
                     labels_differ_outside_tie_set=int(bad.sum()))
            if bad.any():
                msgs.append(f'{int(bad.sum())} pseudo-label pixels differ outside the tie set')
            if float(fragile.mean()) > 0.01:
                msgs.append('tie set is not a small minority')
        else:
            # what the difference is made of: both sides confident and another class (an argmax flip that enters the loss), or a
            # pixel on different sides of the confidence threshold (255 on one side; the synthetic teacher gain is calibrated
            # so that HALF the pixels pass the threshold, i.e. the confidences crowd around it)
            both = (info['label'] != 255) & (ref != 255)
            n_arg, n_mem = int((mism & both).sum()), int((mism & ~both).sum())
            print(f'{name} bf16 labels: {float(mism.mean()):.3%} differ: {n_arg} argmax flips among {int(both.sum())} pixels confident '
                  f'on both sides, {n_mem} on different sides of the threshold')
            C.record(sec, labels_total=int(ref.size), labels_differ=int(mism.sum()), labels_differ_fraction=float(mism.mean()),
                     labels_confident_both=int(both.sum()), labels_argmax_flips_among_confident=n_arg,
                     labels_threshold_side_differs=n_mem)
            if n_arg > 1e-4 * max(1, int(both.sum())):        # measured: see profiles/r05_parity_report.json
                msgs.append(f'{n_arg} of {int(both.sum())} confident pseudo-labels have another class than the reference')
            if float(mism.mean()) > 0.035:
                msgs.append(f'{float(mism.mean()):.2%} of the bf16 pseudo-labels differ from the reference')
        C.record(sec, mask_ratio=info['ratio'], mask_ratio_reference=float(z['teacher_mask_ratio_final']))
        if abs(info['ratio'] - float(z['teacher_mask_ratio_final'])) > (2e-3 if dtype == 'fp32' else 3e-2):
            msgs.append(f'mask ratio {info["ratio"]:.4f} vs {float(z["teacher_mask_ratio_final"]):.4f}')
    assert not msgs, '\n'.join(msgs[:20])


@pytest.mark.parametrize('name', ['full_semi8_fwd', 'full_pasa'])
def test_precise_teacher_gives_the_reference_pseudo_labels_in_the_bf16_mode(name):
    """Round 6 (S4F_TEACHER_PRECISE / runtime.set_teacher_precise): the student in the bf16 perf mode, the teacher pass on the fp32
    parity kernels.  The reference forms softmax / max / `> 0.95` in fp32 (encoder_decoder.py:888-901); with this switch the
    pseudo-label masks of the perf mode differ from the reference's only INSIDE its tie set (the bar of the fp32 mode), while the
    named losses stay inside the bf16 bounds.  `full_semi8_fwd` is cfg3's own batch (8 + 8, PASA), `full_pasa` runs two
    iterations with the EMA in between (the teacher then follows a student stepped on bf16 gradients)."""
    import s4former_amd as S
    from s4former_amd import runtime
    from tests import common as C
    if not os.path.exists(os.path.join(GOLD, f'{name}.npz')):
        pytest.skip(f'{name}.npz not generated')
    runtime.set_teacher_precise(True)
    try:
        z, meta, rec, info, sd = _golden_run(name, 'bf16')
    finally:
        runtime.set_teacher_precise(False)
        S.set_compute_dtype('fp32')
    tl = TOL['bf16']
    keys = [str(k) for k in z['it0_loss_keys']]
    worst = max(abs(rec[0]['log'][k] - v) / abs(v) for k, v in zip(keys, z['it0_loss_vals']) if 'loss' in k)
    assert worst <= tl[0], f'named losses: worst relative error {worst:.2e}'
    ref = z['teacher_label_final']
    mism = info['label'] != ref
    # (two iterations: the final teacher has taken one EMA step towards a student whose update came from bf16 gradients -
    # teacher' - teacher_ref = (1 - m) lr (g_bf16 - g_ref), ~1e-7 of a weight: still the fp32 mode's tie set)
    tol = TOL['fp32'][7] * float(z['teacher_logit_absmax_final'])
    fragile = C.fragile_pixels(z, tol)
    bad = mism & ~fragile
    both = (info['label'] != 255) & (ref != 255)
    print(f'{name} bf16 student + fp32 teacher: {int(mism.sum())} of {ref.size} labels differ, {int(fragile.sum())} ties within {tol:.2e}, '
          f'{int(bad.sum())} outside the tie set, {int((mism & both).sum())} argmax flips; losses worst {worst:.2e}')
    C.record(f'deit_b/{name}/bf16/teacher_fp32', labels_total=int(ref.size), labels_differ=int(mism.sum()), labels_in_tie_set=int(fragile.sum()),
             labels_differ_outside_tie_set=int(bad.sum()), it0_loss_rel_worst=worst, mask_ratio=info['ratio'],
             mask_ratio_reference=float(z['teacher_mask_ratio_final']))
    assert not bad.any(), f'{int(bad.sum())} pseudo-label pixels differ outside the tie set'
    assert float(fragile.mean()) <= 0.01, 'tie set is not a small minority'
    assert abs(info['ratio'] - float(z['teacher_mask_ratio_final'])) <= 2e-3


def test_fullsize_pasa_device_topk_differs_from_the_cpu_choice_only_inside_the_tie_set(monkeypatch):
    """The fp32 PASA comparison above borrows the reference CPU path's choice among TIED patches (S4F_TOPK_TIES=cpu).  Here the
    same DeiT-B fp32 step runs with the DEFAULT device selection - what trains and what is benchmarked - and every selection
    it makes is checked against the definition (vit.py:523-530: the k = 512 least confident of 1024 patches get the bias):
    all patches below the k-th smallest confidence are selected, none above it, exactly k in total, and every patch on which
    the device and torch.topk on the CPU disagree carries exactly the boundary value.  Losses that do not pass through the
    masked rows must still meet the golden at 1e-4; the masked ones are recorded."""
    import s4former_amd as S
    from s4former_amd.vit import VisionTransformer
    from tests import common as C
    name = 'full_pasa'
    if not os.path.exists(os.path.join(GOLD, f'{name}.npz')):
        pytest.skip(f'{name}.npz not generated')
    monkeypatch.delenv('S4F_TOPK_TIES', raising=False)
    seen = []
    inner = VisionTransformer._rank1_mask

    def spy(attn_mask, attn_mask_weight, adaptive_attn_mask):
        u, flag, w = inner(attn_mask, attn_mask_weight, adaptive_attn_mask)
        if flag is not None:
            seen.append((u.detach().cpu().clone(), flag.detach().cpu().clone()))
        return u, flag, w
    monkeypatch.setattr(VisionTransformer, '_rank1_mask', staticmethod(spy))
    try:
        z, meta, rec, info, sd = _golden_run(name, 'fp32')
    finally:
        S.set_compute_dtype('fp32')
    assert seen, 'the PASA mask was never built'
    n_diff = n_tied = n_rows = 0
    for u, flag in seen:
        conf = u[:, 1:]                                   # the cls column carries no confidence and is never selected
        k = int(0.5 * conf.size(-1))
        sel = flag[:, 1:] == 0
        assert bool((flag[:, 0] == 1).all())
        kth = conf.sort(dim=-1)[0][:, k - 1:k]            # k-th smallest per image
        below, at = conf < kth, conf == kth
        assert bool((sel.sum(-1) == k).all()), sel.sum(-1)
        assert bool((sel | ~below).all()), 'a patch below the boundary value was not selected'
        assert bool((~sel | below | at).all()), 'a patch above the boundary value was selected'
        cpu_idx = torch.topk(conf, k, dim=-1, largest=False)[1]
        cpu_sel = torch.zeros_like(sel)
        cpu_sel[torch.arange(conf.size(0)).unsqueeze(1), cpu_idx] = True
        diff = sel ^ cpu_sel
        assert bool((~diff | at).all()), 'device and CPU top-k disagree outside the tie set'
        n_diff += int(diff.sum()); n_tied += int(at.sum()); n_rows += conf.size(0)
    keys = [str(k) for k in z['it0_loss_keys']]
    worst_plain = worst_masked = 0.0
    for k, v in zip(keys, z['it0_loss_vals']):
        if 'loss' not in k:
            continue
        e = abs(rec[0]['log'][k] - v) / abs(v)
        if 'unsup' in k:
            worst_masked = max(worst_masked, e)
        else:
            worst_plain = max(worst_plain, e)
    print(f'PASA selections checked: {len(seen)} calls, {n_rows} images, {n_tied} patches on the boundary value, '
          f'{n_diff} chosen differently by the device; losses: supervised rel {worst_plain:.2e}, masked rel {worst_masked:.2e}')
    C.record('deit_b/full_pasa/fp32_device_topk', pasa_calls=len(seen), images=n_rows, patches_on_boundary_value=n_tied,
             patches_chosen_differently=n_diff, differing_outside_tie_set=0, it0_supervised_loss_rel_worst=worst_plain,
             it0_masked_loss_rel_worst=worst_masked)
    assert worst_plain <= 1e-4, worst_plain
    assert worst_masked <= 1e-3, worst_masked          # measured 1.3e-5: 14 of 28 boundary patches chosen differently
