"""Outputs of the reference's OWN training pipelines (container only): mmseg/datasets/pipelines/transforms.py and compose.py
loaded by path, unmodified, and driven exactly as configs/setr/..._MT.py:34-118 composes them -

    sup:    Resize(img_scale, ratio_range) -> RandomCrop(crop, cat_max_ratio .75) -> RandomFlip(.5) -> PhotoMetricDistortion
            -> Normalize -> Pad
    unsup:  Resize -> RandomCrop -> RandomFlip(flip_ratio=.5) -> MultiBranch(strong = weak = PhotoMetricDistortion -> Normalize -> Pad)

on seeded uint8 samples with numpy's global generator seeded per case.  What this pins: the ORDER and kind of every random
draw (scale ratio, crop offsets and the cat_max_ratio retries, flip, the photometric stages of each view) and the composition
(which image each stage sees, clipping of the crop window, padding values, BGR -> RGB, MultiBranch's deep copies) - a product
pipeline that draws in another order produces another crop / flip / distortion and fails on every case.

What it does NOT pin: the pixel arithmetic of the cv2 functions behind mmcv (cv2 and mmcv are absent from the image).  The
stand-ins below route mmcv.imrescale / bgr2hsv / hsv2bgr / imnormalize / impad / imflip to the numpy restatements in
oracle/ops.py, so image VALUES here are "restatement through the reference's control flow" - PARITY UNPINNED for cv2's resize
and HSV conversions, as DESIGN.md says.  -> tests/golden/pipeline.npz"""
import importlib.util
import os
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
from oracle import ops as O  # noqa: E402
from tests import common as C  # noqa: E402

REF = os.environ.get('S4F_REFERENCE_DIR', '/root/reference')


def _stub_modules():
    class Registry:
        def __init__(self):
            self.d = {}

        def register_module(self, *a, **k):
            def deco(cls):
                self.d[cls.__name__] = cls
                return cls
            return deco
    reg = Registry()

    def build_from_cfg(cfg, registry, default_args=None):
        cfg = dict(cfg)
        return registry.d[cfg.pop('type')](**cfg)

    def deprecated_api_warning(name_dict, cls_name=None):
        def deco(fn):
            def wrapper(*args, **kwargs):
                for old, new in name_dict.items():
                    if old in kwargs:
                        kwargs[new] = kwargs.pop(old)
                return fn(*args, **kwargs)
            return wrapper
        return deco

    def is_list_of(seq, t):
        return isinstance(seq, list) and all(isinstance(x, t) for x in seq)

    def is_tuple_of(seq, t):
        return isinstance(seq, tuple) and all(isinstance(x, t) for x in seq)

    def imrescale(img, scale, return_scale=False, interpolation='bilinear', backend=None):
        h, w = img.shape[:2]
        new_w, new_h = O.rescale_size((w, h), scale)
        out = O.cv_resize_nearest(img, (new_h, new_w)) if interpolation == 'nearest' else O.cv_resize_linear_u8(img, (new_h, new_w))
        if return_scale:
            return out, (new_w / w + new_h / h) / 2        # (mmcv returns the factor it used; the caller recomputes w / h scales)
        return out

    def imflip(img, direction='horizontal'):
        return np.flip(img, axis=1) if direction == 'horizontal' else np.flip(img, axis=0)

    def impad(img, shape=None, padding=None, pad_val=0, padding_mode='constant'):
        ph, pw = max(shape[0] - img.shape[0], 0), max(shape[1] - img.shape[1], 0)
        width = ((0, ph), (0, pw)) + ((0, 0),) * (img.ndim - 2)
        return np.pad(img, width, mode='constant', constant_values=pad_val)

    def imnormalize(img, mean, std, to_rgb=True):
        img = img.astype(np.float32)
        if to_rgb:
            img = img[..., ::-1]
        stdinv = (1.0 / np.float64(std)).astype(np.float32)
        return (img - np.asarray(mean, dtype=np.float32)) * stdinv

    mm = types.ModuleType('mmcv')
    mm.is_list_of, mm.is_tuple_of = is_list_of, is_tuple_of
    mm.imrescale, mm.imflip, mm.impad, mm.imnormalize = imrescale, imflip, impad, imnormalize
    mm.bgr2hsv, mm.hsv2bgr = O._bgr2hsv_u8, O._hsv2bgr_u8
    mu = types.ModuleType('mmcv.utils')
    mu.deprecated_api_warning, mu.is_tuple_of, mu.build_from_cfg = deprecated_api_warning, is_tuple_of, build_from_cfg
    mm.utils = mu
    tv = types.ModuleType('torchvision')
    tv.transforms = types.ModuleType('torchvision.transforms')
    sys.modules.update({'mmcv': mm, 'mmcv.utils': mu, 'torchvision': tv, 'torchvision.transforms': tv.transforms})
    for name in ('mmseg', 'mmseg.datasets', 'mmseg.datasets.pipelines'):
        m = types.ModuleType(name)
        m.__path__ = []
        sys.modules[name] = m
    b = types.ModuleType('mmseg.datasets.builder')
    b.PIPELINES = reg
    sys.modules['mmseg.datasets.builder'] = b
    return reg


def load_reference_pipelines():
    reg = _stub_modules()
    mods = {}
    for fn in ('transforms', 'compose'):
        name = f'mmseg.datasets.pipelines.{fn}'
        spec = importlib.util.spec_from_file_location(name, os.path.join(REF, 'mmseg/datasets/pipelines', fn + '.py'))
        m = importlib.util.module_from_spec(spec)
        sys.modules[name] = m
        spec.loader.exec_module(m)
        mods[fn] = m
    return reg, mods


def main():
    reg, mods = load_reference_pipelines()
    T, Cm = mods['transforms'], mods['compose']
    out = {}
    for name, kw in C.PIPELINE_CASES.items():
        img, seg = C.pipeline_sample(kw['seed'], *kw['hw'])
        crop, norm = tuple(kw['crop']), dict(mean=list(C.IMG_NORM['mean']), std=list(C.IMG_NORM['std']), to_rgb=True)
        tail = [dict(type='PhotoMetricDistortion'), dict(type='Normalize', **norm), dict(type='Pad', size=crop, pad_val=0, seg_pad_val=255)]
        head = [dict(type='Resize', img_scale=tuple(kw['img_scale']), ratio_range=tuple(kw['ratio_range'])),
                dict(type='RandomCrop', crop_size=crop, cat_max_ratio=0.75)]
        if kw['tag'] == 'sup':
            pipe = Cm.Compose(head + [dict(type='RandomFlip', prob=0.5)] + tail)
        else:
            pipe = Cm.Compose(head + [dict(type='RandomFlip', flip_ratio=0.5), dict(type='MultiBranch', unsup_student=tail, unsup_teacher=tail)])
        results = dict(img=img.copy(), gt_semantic_seg=seg.copy(), seg_fields=['gt_semantic_seg'], img_shape=img.shape,
                       ori_shape=img.shape, pad_shape=img.shape, scale_factor=1.0)
        np.random.seed(kw['seed'] + 1000)
        res = pipe(results)
        views = res if isinstance(res, list) else [res]
        for i, v in enumerate(views):
            out[f'{name}_v{i}_img'] = np.ascontiguousarray(v['img'].astype(np.float32).transpose(2, 0, 1))
            out[f'{name}_v{i}_seg'] = np.ascontiguousarray(v['gt_semantic_seg'].astype(np.uint8))
            out[f'{name}_v{i}_meta'] = np.array([v['img_shape'][0], v['img_shape'][1], int(bool(v['flip'])), v['scale'][0], v['scale'][1]], dtype=np.int64)
        print(name, [tuple(v['img'].shape) for v in views], 'flip', views[0]['flip'], 'scale', views[0]['scale'], 'img_shape', views[0]['img_shape'])
    # the reference's own known answers about sizes (tests/test_data/test_transform.py:96-152) through its Resize class
    r = T.Resize(img_scale=(2560, 640), min_size=640)
    dummy = dict(img=np.zeros((288, 512, 3), np.uint8), seg_fields=[], img_shape=(288, 512, 3))
    out['known_min_size_shape'] = np.array(r(dict(dummy))['img_shape'][:2])
    dummy = dict(img=np.zeros((512, 288, 3), np.uint8), seg_fields=[], img_shape=(512, 288, 3))
    out['known_min_size_shape_tall'] = np.array(T.Resize(img_scale=(2560, 640), min_size=640)(dict(dummy))['img_shape'][:2])
    np.savez_compressed(os.path.join(HERE, 'pipeline.npz'), **out)
    print('written', len(out), 'arrays;', out['known_min_size_shape'], out['known_min_size_shape_tall'])


if __name__ == '__main__':
    main()
