"""Host half of the device input pipeline: the random decisions of the reference's training pipeline, drawn from
numpy's global generator in the reference's call order, and the launch of the one-pass view kernel (csrc/pipeline.hip).

Covered (configs/setr/..._MT.py:34-118): Resize(img_scale, ratio_range) - `draw_resize` (the resized image is never built: the
kernel interpolates the pixels of the crop window from the source); RandomCrop(crop_size, cat_max_ratio) - `random_crop_bbox`;
RandomFlip(prob) - `draw_flip`; PhotoMetricDistortion - `draw_photometric`; Normalize + Pad + DefaultFormatBundle - the kernel;
MultiBranch (compose.py:69-83: a strong and a weak view of the SAME scale / crop / flip, each with its own photometric draw) -
`semi_views`.  The draw ORDER is pinned against the reference's own transforms.py (tests/golden/pipeline.npz).  Image decoding
stays in the CPU dataset layer (outside SURVEY §8)."""
import ctypes

import numpy as np
import torch

from . import _lib as L
from ._lib import S4FError

IMG_NORM = dict(mean=(123.675, 116.28, 103.53), std=(58.395, 57.12, 57.375), to_rgb=True)     # configs/setr/*:9-10


def rescale_size(old_wh, scale):
    """mmcv.image.geometric.rescale_size: (w, h), scale = factor | (edge, edge) -> (new_w, new_h)"""
    w, h = old_wh
    if isinstance(scale, (float, int)):
        factor = scale
    else:
        factor = min(max(scale) / max(h, w), min(scale) / min(h, w))
    return int(w * float(factor) + 0.5), int(h * float(factor) + 0.5)


def draw_resize(img_hw, img_scale=(2048, 512), ratio_range=(0.5, 2.0), min_size=None):
    """Resize._random_scale + the size _resize_img arrives at with keep_ratio=True (transforms.py:286-308,318-383): ONE
    np.random.random_sample() when ratio_range is given.  -> (new_h, new_w) of the resized image"""
    h, w = int(img_hw[0]), int(img_hw[1])
    if ratio_range is not None:
        base = (w, h) if img_scale is None else tuple(img_scale)
        lo, hi = ratio_range
        ratio = np.random.random_sample() * (hi - lo) + lo
        scale = int(base[0] * ratio), int(base[1] * ratio)
    else:
        scale = tuple(img_scale)
    if min_size is not None:
        new_short = min_size if min(scale) < min_size else min(scale)
        scale = (new_short * h / w, new_short) if h > w else (new_short, new_short * w / h)
    new_w, new_h = rescale_size((w, h), scale)
    return new_h, new_w


def resize_seg_nearest_host(seg, new_hw):
    """the resized label map on the HOST (cv2.INTER_NEAREST rule), for RandomCrop's cat_max_ratio test only"""
    H, W = seg.shape[:2]
    RH, RW = int(new_hw[0]), int(new_hw[1])
    if (RH, RW) == (H, W):
        return seg
    sy = np.minimum(np.floor(np.arange(RH) * (1.0 / (RH / H))).astype(np.int64), H - 1)
    sx = np.minimum(np.floor(np.arange(RW) * (1.0 / (RW / W))).astype(np.int64), W - 1)
    return seg[sy][:, sx]


def random_crop_bbox(img_hw, seg, crop_size, cat_max_ratio=1.0, ignore_index=255):
    """RandomCrop.get_crop_bbox + the cat_max_ratio retry loop (transforms.py:820-859) -> (y1, y2, x1, x2); seg: numpy [H, W]"""
    def draw():
        mh, mw = max(img_hw[0] - crop_size[0], 0), max(img_hw[1] - crop_size[1], 0)
        oh = np.random.randint(0, mh + 1)
        ow = np.random.randint(0, mw + 1)
        return oh, oh + crop_size[0], ow, ow + crop_size[1]
    box = draw()
    if cat_max_ratio < 1.0:
        for _ in range(10):
            tmp = seg[box[0]:box[1], box[2]:box[3]]
            labels, cnt = np.unique(tmp, return_counts=True)
            cnt = cnt[labels != ignore_index]
            if len(cnt) > 1 and np.max(cnt) / np.sum(cnt) < cat_max_ratio:
                break
            box = draw()
    return box


def draw_flip(prob=0.5):
    """RandomFlip.__call__ (transforms.py:450-459): np.random.rand() < prob"""
    return bool(np.random.rand() < prob)


def draw_photometric(brightness_delta=32, contrast_range=(0.5, 1.5), saturation_range=(0.5, 1.5), hue_delta=18):
    """PhotoMetricDistortion.__call__ (transforms.py:1203-1268): the nine parameters of the kernel, drawn in the reference's
    order (brightness, mode, [contrast], saturation, hue, [contrast]); every stage fires when randint(2) == 0"""
    rnd = np.random
    p = np.zeros(9, dtype=np.float32)
    if 1 - rnd.randint(2):
        p[0], p[1] = 1, rnd.uniform(-brightness_delta, brightness_delta)
    mode = rnd.randint(2)
    p[4] = 1 if mode == 1 else 0

    def contrast():
        if 1 - rnd.randint(2):
            p[2], p[3] = 1, rnd.uniform(contrast_range[0], contrast_range[1])
    if mode == 1:
        contrast()
    if 1 - rnd.randint(2):
        p[5], p[6] = 1, rnd.uniform(saturation_range[0], saturation_range[1])
    if 1 - rnd.randint(2):
        p[7], p[8] = 1, rnd.randint(-hue_delta, hue_delta)
    if mode == 0:
        contrast()
    return p


NO_PHOTOMETRIC = np.zeros(9, dtype=np.float32)


def input_view(img_u8, seg_u8, bbox, flip, photo, crop_size, out_img=None, out_seg=None, norm=IMG_NORM, pad_val=0.0,
               seg_pad_val=255, flip_direction='horizontal', resize_to=None):
    """one view of one sample on the device: img_u8 uint8 [H, W, 3] BGR, seg_u8 uint8 [H, W] | None (device tensors);
    resize_to (new_h, new_w) | None: the size `draw_resize` drew - bbox then refers to the resized image;
    bbox (y1, y2, x1, x2) as RandomCrop draws it (clipped to the image here, like numpy slicing does).
    -> (img fp32 [3, ch, cw] normalised + padded, seg uint8 [ch, cw] | None, img_shape (h, w, 3))"""
    if not img_u8.is_cuda or img_u8.dtype != torch.uint8 or img_u8.dim() != 3 or img_u8.shape[2] != 3 or not img_u8.is_contiguous():
        raise S4FError('input_view: a contiguous device uint8 [H, W, 3] image is expected')
    H, W = int(img_u8.shape[0]), int(img_u8.shape[1])
    if seg_u8 is not None and (seg_u8.dtype != torch.uint8 or tuple(seg_u8.shape) != (H, W) or not seg_u8.is_contiguous()):
        raise S4FError('input_view: seg must be uint8 [H, W]')
    RH, RW = (H, W) if resize_to is None else (int(resize_to[0]), int(resize_to[1]))
    y1, y2, x1, x2 = bbox
    y2, x2 = min(y2, RH), min(x2, RW)
    ch, cw = int(crop_size[0]), int(crop_size[1])
    if out_img is None:
        out_img = torch.empty(3, ch, cw, device=img_u8.device, dtype=torch.float32)
    if out_seg is None and seg_u8 is not None:
        out_seg = torch.empty(ch, cw, device=img_u8.device, dtype=torch.uint8)
    crop = (ctypes.c_int * 4)(int(y1), int(x1), int(y2 - y1), int(x2 - x1))
    ph = (ctypes.c_float * 9)(*[float(v) for v in photo])
    order = (0, 1, 2)                # mean / std are given in the OUTPUT channel order (RGB when to_rgb)
    mean = (ctypes.c_float * 3)(*[float(norm['mean'][i]) for i in order])
    std = (ctypes.c_float * 3)(*[float(norm['std'][i]) for i in order])
    fl = 0 if not flip else (1 if flip_direction == 'horizontal' else 2)
    L.call('s4f_input_view_resized', L.p(img_u8), L.p(seg_u8), L.p(out_img), L.p(out_seg), H, W, RH, RW, ch, cw, crop, fl, ph,
           mean, std, 1 if norm.get('to_rgb', True) else 0, float(pad_val), int(seg_pad_val), L.stream())
    return out_img, out_seg, (y2 - y1, x2 - x1, 3)


def semi_views(img_u8, seg_u8, seg_host, crop_size, cat_max_ratio=0.75, flip_prob=0.5, tag='unsup', filename='', resize=None):
    """the per-sample part of the semi-supervised pipelines: [Resize ->] RandomCrop -> RandomFlip -> (labelled sample: one view,
    tag 'sup'; unlabelled sample: MultiBranch's strong + weak views, tags 'unsup_student' / 'unsup_teacher' in that order,
    configs/setr/..._MT.py:34-118).  resize: None | dict(img_scale=(2048, 512), ratio_range=(0.5, 2.0)[, min_size]) - the
    configs' multi-scale Resize.  Returns a list of dict(img, gt_semantic_seg, img_metas)."""
    resize_to = None
    if resize is not None:
        resize_to = draw_resize(img_u8.shape[:2], **resize)
        seg_host = resize_seg_nearest_host(seg_host, resize_to) if seg_host is not None else None
    bbox = random_crop_bbox(img_u8.shape[:2] if resize_to is None else resize_to, seg_host, crop_size, cat_max_ratio)
    flip = draw_flip(flip_prob)
    tags = ['sup'] if tag == 'sup' else ['unsup_student', 'unsup_teacher']
    out = []
    for t in tags:
        photo = draw_photometric()
        img, seg, shape = input_view(img_u8, seg_u8, bbox, flip, photo, crop_size, resize_to=resize_to)
        out.append(dict(img=img, gt_semantic_seg=seg.unsqueeze(0) if seg is not None else None,
                        img_metas=dict(tag=t, filename=filename, ori_filename=filename, img_shape=shape,
                                       pad_shape=(crop_size[0], crop_size[1], 3), flip=flip, flip_direction='horizontal',
                                       img_norm_cfg=dict(IMG_NORM))))
    return out
