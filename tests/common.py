"""Shared, repo-owned deterministic fixtures: model configs, weights and batches are generated from seeded CPU
generators so that the build container (where the goldens are made from the reference's own code) and the GPU
box (where /root/reference does not exist) see bit-identical tensors."""
import copy
import hashlib

import numpy as np
import torch

NORM_BB = dict(type='LN', eps=1e-6, requires_grad=True)
NORM_HEAD = dict(type='SyncBN', requires_grad=True)


def tiny_model_cfg(img=64, embed=256, layers=4, heads=4, channels=128, num_classes=21, **flags):
    """a scaled-down copy of configs/setr/*_MT.py:137-241 (same structure: 4 taps, PUP decode head with 4 convs
    x2, four aux heads with 2 convs x4)"""
    backbone = dict(type='VisionTransformer', img_size=(img, img), patch_size=16, in_channels=3, embed_dims=embed,
                    num_layers=layers, num_heads=heads, out_indices=tuple(range(layers - 4, layers)), drop_rate=0.0,
                    norm_cfg=NORM_BB, with_cls_token=True, interpolate_mode='bilinear')
    decode = dict(type='SETRUPHead', in_channels=embed, channels=channels, in_index=3, num_classes=num_classes,
                  dropout_ratio=0, norm_cfg=NORM_HEAD, num_convs=4, up_scale=2, kernel_size=3, align_corners=False,
                  loss_decode=dict(type='CrossEntropyLoss', use_sigmoid=False, loss_weight=1.0))
    aux = [dict(type='SETRUPHead', in_channels=embed, channels=channels, in_index=i, num_classes=num_classes,
                dropout_ratio=0, norm_cfg=NORM_HEAD, num_convs=2, up_scale=4, kernel_size=3, align_corners=False,
                loss_decode=dict(type='CrossEntropyLoss', use_sigmoid=False, loss_weight=0.4)) for i in range(4)]
    cfg = dict(type='EncoderDecoder', pretrained=None, backbone=backbone, backbone_ema=copy.deepcopy(backbone),
               auxiliary_head=aux, decode_head=decode, decode_head_ema=copy.deepcopy(decode), ema=True, ema_momentum=0.999,
               unsup_weight=1.0, unsup_confidence=0.95, test_cfg=dict(mode='whole'))
    cfg.update(flags)
    return cfg


def deit_b_cfg(img=512, num_classes=21, **flags):
    """the model dict of configs/setr/setr_deit-base_pup_..._MT.py:137-241 (values as resolved by the survey)"""
    cfg = tiny_model_cfg(img=img, embed=768, layers=12, heads=12, channels=256, num_classes=num_classes, **flags)
    for k in ('backbone', 'backbone_ema'):
        cfg[k]['out_indices'] = (4, 7, 9, 11)
    return cfg


def fill_state(sd_keys_shapes, seed, teacher_seg_gain=1.0):
    """deterministic values for every state-dict entry, by key order: weights ~ N(0, s) with a per-kind scale,
    BN/LN weights near 1, running_var > 0.  Returns an OrderedDict of CPU fp32 tensors."""
    g = torch.Generator().manual_seed(seed)
    out = {}
    for k, shape in sd_keys_shapes:
        leaf = k.split('.')[-1]
        if leaf == 'num_batches_tracked':
            out[k] = torch.zeros(shape, dtype=torch.long)
            continue
        t = torch.randn(shape, generator=g)
        if leaf == 'running_var':
            t = t.abs() * 0.2 + 0.8
        elif leaf == 'running_mean':
            t = t * 0.1
        elif leaf in ('weight',) and len(shape) == 1:           # LN / BN gamma
            t = 1.0 + 0.1 * t
        elif leaf in ('bias', 'in_proj_bias') and len(shape) == 1:
            t = 0.02 * t
        elif 'conv_seg.weight' in k:
            t = 0.05 * t * (teacher_seg_gain if k.startswith('decode_head_ema') else 1.0)
        elif len(shape) == 4 and shape[-1] == 3:                 # 3x3 convs: kaiming fan_out
            t = t * (2.0 / (shape[0] * 9)) ** 0.5
        elif len(shape) == 4:                                    # patch embed
            t = t * (2.0 / (shape[1] * shape[2] * shape[3])) ** 0.5
        elif k.endswith('pos_embed') or k.endswith('cls_token'):
            t = 0.02 * t
        else:                                                    # linear weights
            t = 0.04 * t
        out[k] = t
    return out


def load_filled(model, seed, teacher_seg_gain=1.0):
    sd = model.state_dict()
    vals = fill_state([(k, tuple(v.shape)) for k, v in sd.items()], seed, teacher_seg_gain)
    missing = model.load_state_dict(vals, strict=True)
    return vals


def make_batch(seed, n_sup, n_unsup, img=64, num_classes=21, block=8, border=2):
    """SURVEY §8d synthetic inputs: img ~ N(0,1) clipped to [-2.2, 2.7]; labels = blocks of a uniform class with a
    255 border band; student/teacher views = same crop + independent N(0, 0.1^2) noise; tags in the order
    sup..., unsup_student..., unsup_teacher..., matching filenames."""
    g = torch.Generator().manual_seed(seed)
    n = n_sup + 2 * n_unsup
    base = torch.randn(n_sup + n_unsup, 3, img, img, generator=g).clamp_(-2.2, 2.7)
    imgs = [base[:n_sup]]
    if n_unsup:
        u = base[n_sup:]
        imgs.append((u + 0.1 * torch.randn(u.shape, generator=g)).clamp_(-2.2, 2.7))
        imgs.append((u + 0.1 * torch.randn(u.shape, generator=g)).clamp_(-2.2, 2.7))
    imgs = torch.cat(imgs, 0).contiguous()
    nb = img // block
    cls = torch.randint(0, num_classes, (n, nb, nb), generator=g)
    gt = cls.repeat_interleave(block, 1).repeat_interleave(block, 2)
    band = torch.zeros(img, dtype=torch.bool)
    for s in range(0, img, block * 4):
        band[s:s + border] = True
    gt[:, band, :] = 255
    gt[:, :, band] = 255
    gt = gt.unsqueeze(1).contiguous()
    metas = []
    for i in range(n_sup):
        metas.append(dict(tag='sup', filename=f'sup_{i}.jpg'))
    for i in range(n_unsup):
        metas.append(dict(tag='unsup_student', filename=f'{i}.jpg'))
    for i in range(n_unsup):
        metas.append(dict(tag='unsup_teacher', filename=f'{i}.jpg'))
    return imgs, gt, metas


def seed_host_rng(seed):
    """what tests/golden/make_golden.py did before every iteration of the reference: the "ours" augmentations (CutMix box,
    PatchShuffle permutation) draw from numpy's and torch's global generators"""
    np.random.seed(seed)
    torch.manual_seed(seed)


def sha(t):
    return hashlib.sha256(t.detach().cpu().contiguous().numpy().tobytes()).hexdigest()


def grad_norms(model, prefix_filter=None):
    out = {}
    for n, p in model.named_parameters():
        if p.grad is not None and (prefix_filter is None or n.startswith(prefix_filter)):
            out[n] = float(p.grad.detach().float().norm())
    return out


NS = 64


def grad_sample(g, ns=NS):
    """ns elements of the flattened LOGICAL (torch-layout) tensor at a fixed stride: the rule tests/golden/make_golden*.py
    used on the reference's gradients"""
    f = g.detach().reshape(-1)
    step = max(1, f.numel() // ns)
    return f[::step][:ns]


# ------------------------------------------------------------------------------------------------ measured parity margins
# Every step / full-size test RECORDS what it measured (not only pass / fail): tests/conftest.py writes the records of a
# session to gpurun_out/parity_report.json, the copy of a GPU run is committed as profiles/r03_parity_report.json, and the
# bf16 bounds of the tests are set from it (<= 3 x the measured value).
PARITY = {}


def record(section, **kv):
    PARITY.setdefault(section, {}).update({k: (float(v) if isinstance(v, (int, float, np.floating)) else v) for k, v in kv.items()})


def check_grad_samples(z, it, named_grads, tol, msgs, label='', rec=None, mtol=None, noise=None):
    """Element-wise comparison of EVERY parameter's gradient with the golden samples, relative to the tensor's largest element.

    Iteration 0 is held against the reference's FLOAT64 evaluation of the same step (`it0_gs64`): |got - ref64| <= max(tol, 8 D)
    for the worst tensor and <= max(tol, 3 D) for the 90th percentile over tensors, where D is the distance of the reference's OWN fp32 gradients from that fp64 evaluation (worst tensor).  On the tiny fixtures
    (batch seeds without ReLU ties, make_golden.relu_margin) D ~ 2e-6 and the bound is `tol` itself (1e-4 in fp32 mode); at
    DeiT-B size the reference's fp32 gradients are themselves only good to D ~ 5e-3 (tens of ReLU decisions within rounding of
    zero flip between fp32 and fp64), and no implementation can be closer to the reference than the reference is to itself.
    The median over tensors must stay within max(tol / 10, 4 median D).  Later iterations: against the fp32 samples.

    bf16 perf mode (tol >= 0.1, a noise regime: one ReLU decided the other way moves a BatchNorm bias gradient of the tiny model
    by a large fraction of its maximum, and the fp32 atomics of split-K sums make WHICH decisions flip vary from run to run -
    the worst tensor of `mt_ours`, iteration 1, came out at 0.52 in one of four runs of one build and below 0.5 in the others): `tol` bounds the 90th percentile over the
    tensors, the single worst tensor may reach 3 tol; the median bound (`mtol`, default tol / 10) is what holds the bulk.
    Measured values are recorded under `rec` (profiles/r03_parity_report.json)."""
    keys = [str(k) for k in z[f'it{it}_gn_keys']]
    gs, gmax = z[f'it{it}_gs'], z[f'it{it}_gmax']
    ref, D, Dmed = gs, 0.0, 0.0
    if it == 0 and 'it0_gs64' in z.files:
        ref = z['it0_gs64']
        d = np.abs(gs.astype(np.float64) - ref).max(axis=1) / (gmax + 1e-30)
        D, Dmed = float(d.max()), float(np.median(d))
    elif it == 0 and noise is not None:
        # fixtures without an fp64 evaluation (the full-batch ones: it does not fit the build container): the reference's
        # fp32-vs-fp64 distance is taken from the small-batch fixture of the same model (reference_noise)
        D, Dmed = noise
    errs = []
    for i, k in enumerate(keys):
        got = grad_sample(named_grads[k], gs.shape[1]).double().cpu().numpy()
        errs.append(float(np.abs(got - ref[i, :got.size]).max()) / (float(gmax[i]) + 1e-30))
    w = int(np.argmax(errs))
    if rec is not None:
        record(rec, **{f'it{it}_grad_elem_worst': errs[w], f'it{it}_grad_elem_worst_tensor': keys[w],
                       f'it{it}_grad_elem_p90': float(np.percentile(errs, 90)), f'it{it}_grad_elem_median': float(np.median(errs)),
                       f'it{it}_reference_fp32_vs_fp64_worst': D, f'it{it}_reference_fp32_vs_fp64_median': Dmed})
    bound, mbound = max(tol, 8 * D), max(tol / 10 if mtol is None else mtol, 4 * Dmed)
    p90 = float(np.percentile(errs, 90))
    if tol >= 0.1:
        if p90 > bound:
            msgs.append(f'{label}it{it} gradient elements, 90th percentile over tensors: {p90:.2e} (bound {bound:.1e})')
        bound = 3 * bound
    elif p90 > max(tol, 3 * D):
        # fp32 parity mode (round 4): nine tensors in ten are held to 3 D.  The single worst tensor keeps 8 D: at DeiT-B size it is
        # always a BatchNorm bias behind a ReLU (full_pasa: auxiliary_head.0.up_convs.0.0.bn.bias at 6.4 D), where ONE activation
        # decided the other way than in the reference's fp32 run moves the whole sum - the same event that makes up D itself.
        msgs.append(f'{label}it{it} gradient elements, 90th percentile over tensors: {p90:.2e} (bound {max(tol, 3 * D):.1e})')
    if errs[w] > bound:
        msgs.append(f'{label}it{it} gradient elements of {keys[w]}: {errs[w]:.2e} of the tensor maximum (bound {bound:.1e}; the '
                    f"reference's own fp32 is {D:.1e} from its fp64 evaluation)")
    if float(np.median(errs)) > mbound:
        msgs.append(f'{label}it{it} gradient elements, median over tensors: {float(np.median(errs)):.2e} (bound {mbound:.1e})')
    return errs[w], keys[w], float(np.median(errs)), D


def reference_noise(z):
    """(worst, median) over tensors of the reference's own fp32 gradient samples' distance from its fp64 evaluation"""
    gs, gmax, ref = z['it0_gs'], z['it0_gmax'], z['it0_gs64']
    d = np.abs(gs.astype(np.float64) - ref).max(axis=1) / (gmax + 1e-30)
    return float(d.max()), float(np.median(d))


def fragile_pixels(z, logit_tol, th=0.95):
    """pixels whose pseudo-label decision the REFERENCE's own arithmetic makes by less than the stated bound: a top-2 logit
    margin below logit_tol (argmax may flip) or a softmax maximum within the band around the threshold that a logit
    perturbation of logit_tol can move it by (|dp| <= p (1 - p) * 2 * logit_tol <= logit_tol / 2; confidence may flip)"""
    if 'teacher_margin_final' in z.files:
        margin, pmax = z['teacher_margin_final'], z['teacher_pmax_final']
        return (margin < logit_tol) | (np.abs(pmax - th) < 0.5 * logit_tol + 1e-7)
    # full-size goldens store the candidates only (every pixel within `frag` * max |logit|): sparse index + values
    shape = z['teacher_label_final'].shape
    idx, margin, pmax = z['teacher_frag_idx'], z['teacher_frag_margin'], z['teacher_frag_pmax']
    out = np.zeros(int(np.prod(shape)), dtype=bool)
    out[idx] = (margin < logit_tol) | (np.abs(pmax - th) < 0.5 * logit_tol + 1e-7)
    return out.reshape(shape)


METRIC_CASES = {'voc21': (21, False), 'city19': (19, False), 'ade_rz': (20, True)}     # name: (num_classes, reduce_zero_label)


def metric_maps(name, n=3, size=(97, 131)):
    """seeded prediction / label maps for the metric fixtures: labels with an ignore band (255), predictions that agree on ~60 %
    of the pixels; 'ade_rz' labels start at 0 = "nothing" (reduce_zero_label)"""
    ncls, rz = METRIC_CASES[name]
    g = np.random.RandomState(hash(name) % 1000 + 7 if False else {'voc21': 11, 'city19': 12, 'ade_rz': 13}[name])
    preds, labels = [], []
    for i in range(n):
        hi = ncls + 1 if rz else ncls
        lab = g.randint(0, hi, size=size).astype(np.uint8)
        lab[: 5 + i] = 255
        pred = np.where(g.rand(*size) < 0.6, (lab.astype(np.int64) - (1 if rz else 0)) % ncls, g.randint(0, ncls, size=size)).astype(np.int64)
        preds.append(pred)
        labels.append(lab)
    return preds, labels


SAMPLER_CASES = {
    # VOC 1/16 classic split of the SETR configs: 662 labelled + 9920 unlabelled, 4 + 4 per GPU (configs/setr/*:31-33)
    'voc16_8gpu': dict(cumulative_sizes=[662, 10582], sample_ratio=[1, 1], samples_per_gpu=8, num_replicas=8, max_iter_size=40,
                       epochs=[0, 3]),
    'small_1gpu': dict(cumulative_sizes=[10, 37], sample_ratio=[1, 1], samples_per_gpu=4, num_replicas=1, max_iter_size=25,
                       epochs=[0, 1]),
    'ratio_1_3': dict(cumulative_sizes=[20, 100], sample_ratio=[1, 3], samples_per_gpu=8, num_replicas=2, max_iter_size=12,
                      epochs=[5]),
}


# ------------------------------------------------------------------------------------------------ sliding-window evaluation
SLIDE_CASE = dict(crop=(64, 64), stride=(32, 48), img_shape=(90, 100), ori_shape=(120, 131), seed_w=1999, gain=5.0, seed_x=91)


WHOLE_CASE = dict(img_shape=(60, 56), ori_shape=(83, 71), seed_x=77)      # with SLIDE_CASE's weights


def slide_input():
    g = torch.Generator().manual_seed(SLIDE_CASE['seed_x'])
    return torch.randn(2, 3, 96, 112, generator=g).clamp_(-2.2, 2.7)


# ------------------------------------------------------------------------------------------------ input pipeline fixtures
IMG_NORM = dict(mean=(123.675, 116.28, 103.53), std=(58.395, 57.12, 57.375))      # configs/setr/*:9-10

# name: the reference's train / unsup_train pipeline (configs/setr/..._MT.py:34-118) on a seeded sample; numpy seeded with seed + 1000
PIPELINE_CASES = {
    'sup_a': dict(tag='sup', seed=1, hw=(70, 90), crop=(64, 64), img_scale=(200, 100), ratio_range=(0.5, 2.0)),
    'sup_b': dict(tag='sup', seed=2, hw=(96, 64), crop=(64, 64), img_scale=(160, 96), ratio_range=(0.5, 2.0)),
    'sup_small': dict(tag='sup', seed=3, hw=(60, 80), crop=(64, 64), img_scale=(100, 50), ratio_range=(0.5, 1.0)),   # resized image smaller than the crop: Pad fills
    'sup_c': dict(tag='sup', seed=4, hw=(50, 130), crop=(48, 80), img_scale=(260, 100), ratio_range=(0.5, 2.0)),
    'unsup_a': dict(tag='unsup', seed=5, hw=(70, 90), crop=(64, 64), img_scale=(200, 100), ratio_range=(0.5, 2.0)),
    'unsup_b': dict(tag='unsup', seed=6, hw=(128, 100), crop=(64, 64), img_scale=(256, 128), ratio_range=(0.5, 2.0)),
    'unsup_c': dict(tag='unsup', seed=7, hw=(81, 121), crop=(64, 64), img_scale=(242, 81), ratio_range=(0.9, 1.1)),
    'unsup_d': dict(tag='unsup', seed=8, hw=(64, 64), crop=(64, 64), img_scale=(128, 128), ratio_range=(0.5, 2.0)),
}


def pipeline_sample(seed, h, w):
    """uint8 BGR image with smooth structure (so that interpolation matters) + label blocks with an ignore band"""
    g = np.random.RandomState(seed)
    yy, xx = np.mgrid[0:h, 0:w]
    base = np.stack([(yy * 3 + xx * 2) % 256, (yy * 5 + 40) % 256, (xx * 7 + yy) % 256], -1).astype(np.int64)
    img = np.clip(base + g.randint(-20, 21, size=(h, w, 3)), 0, 255).astype(np.uint8)
    img[:4, :4] = 0
    img[4:8, :4] = 255
    seg = (g.randint(0, 21, size=((h + 15) // 16, (w + 15) // 16)).repeat(16, 0).repeat(16, 1)[:h, :w]).astype(np.uint8)
    seg[::11] = 255
    return img, seg
