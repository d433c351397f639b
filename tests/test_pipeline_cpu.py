"""CPU: the host decisions of the device input pipeline (s4former_amd/pipeline.py) and the oracle's composition against the
outputs of the reference's OWN pipelines (tests/golden/pipeline.npz: transforms.py + compose.py run as configs/setr/..._MT.py
composes them, tests/golden/make_golden_pipeline.py).  A draw taken in another order than the reference's, a crop window clipped
differently or a wrong pad value fails every case.  The reference's own known answers about sizes
(tests/test_data/test_transform.py:96-152) are restated at the end."""
import os

import numpy as np
import pytest

from oracle import ops as O
from s4former_amd import pipeline as P
from tests import common as C

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'pipeline.npz')


def host_decisions(kw, img, seg):
    """the draws of one sample in the reference's order: Resize ratio, crop offsets (+ retries), flip, then one photometric
    draw per emitted view"""
    np.random.seed(kw['seed'] + 1000)
    resize_to = P.draw_resize(img.shape[:2], img_scale=tuple(kw['img_scale']), ratio_range=tuple(kw['ratio_range']))
    seg_r = P.resize_seg_nearest_host(seg, resize_to)
    bbox = P.random_crop_bbox(resize_to, seg_r, tuple(kw['crop']), 0.75)
    flip = P.draw_flip(0.5)
    photos = [P.draw_photometric() for _ in range(1 if kw['tag'] == 'sup' else 2)]
    return resize_to, bbox, flip, photos


@pytest.mark.parametrize('name', sorted(C.PIPELINE_CASES))
def test_host_draws_and_oracle_reproduce_the_reference_pipeline(name):
    z = np.load(GOLD)
    kw = C.PIPELINE_CASES[name]
    img, seg = C.pipeline_sample(kw['seed'], *kw['hw'])
    resize_to, bbox, flip, photos = host_decisions(kw, img, seg)
    yb = (bbox[0], min(bbox[1], resize_to[0]), bbox[2], min(bbox[3], resize_to[1]))
    for i, photo in enumerate(photos):
        meta = z[f'{name}_v{i}_meta']
        assert (yb[1] - yb[0], yb[3] - yb[2]) == (int(meta[0]), int(meta[1])), 'img_shape after the crop'
        assert bool(meta[2]) == flip
        gi, gs = O.input_view(img, seg, yb, flip, photo, tuple(kw['crop']), resize_to=resize_to)
        assert np.array_equal(gs, z[f'{name}_v{i}_seg']), f'{name} view {i}: labels'
        d = np.abs(gi - z[f'{name}_v{i}_img']).max()
        assert d <= 1e-5, f'{name} view {i}: image differs by {d}'
    if kw['tag'] != 'sup':
        assert np.array_equal(z[f'{name}_v0_seg'], z[f'{name}_v1_seg'])            # MultiBranch: same crop + flip for both views


def test_resize_sizes_known_answers():
    """tests/test_data/test_transform.py:96-152 restated (mmcv.rescale_size + Resize._resize_img's min_size rule)"""
    z = np.load(GOLD)
    assert P.rescale_size((512, 288), (1333, 800)) == (1333, 750) and P.rescale_size((512, 288), (1333, 400)) == (711, 400)
    assert P.draw_resize((288, 512), img_scale=(2560, 640), ratio_range=None, min_size=640) == (640, 1138) == tuple(z['known_min_size_shape'])
    assert P.draw_resize((288, 512), img_scale=(512, 640), ratio_range=None, min_size=640) == (640, 1138)
    assert P.draw_resize((512, 288), img_scale=(2560, 640), ratio_range=None, min_size=640) == (1138, 640) == tuple(z['known_min_size_shape_tall'])
    for seed in range(20):           # ratio_range=(0.9, 1.1) of (1333, 800): the long edge stays within 1333 * 1.1
        np.random.seed(seed)
        h, w = P.draw_resize((288, 512), img_scale=(1333, 800), ratio_range=(0.9, 1.1))
        assert max(h, w) <= 1333 * 1.1
        np.random.seed(seed)         # img_scale=None, ratio_range=(0.5, 2.0): relative to the image's own size
        h, w = P.draw_resize((288, 512), img_scale=None, ratio_range=(0.5, 2.0))
        assert int(288 * 0.5) <= h <= 288 * 2.0 and int(512 * 0.5) <= w <= 512 * 2.0


def test_oracle_resize_rules():
    """the two cv2 rules the restatement follows: identity at the same size, the 2 x 2 area mean at an exact half, nearest =
    floor(d * scale); a constant image stays constant under the fixed-point bilinear pass"""
    img, seg = C.pipeline_sample(3, 40, 60)
    assert np.array_equal(O.cv_resize_linear_u8(img, (40, 60)), img)
    half = O.cv_resize_linear_u8(img, (20, 30)).astype(np.int64)
    a = img.astype(np.int64)
    assert np.array_equal(half, (a[0::2, 0::2] + a[0::2, 1::2] + a[1::2, 0::2] + a[1::2, 1::2] + 2) >> 2)
    assert np.array_equal(O.cv_resize_nearest(seg, (80, 120)), seg.repeat(2, 0).repeat(2, 1))
    const = np.full((17, 23, 3), 77, np.uint8)
    assert np.array_equal(O.cv_resize_linear_u8(const, (40, 31)), np.full((40, 31, 3), 77, np.uint8))
