"""GPU: the one-pass input-view kernel (crop / flip / PhotoMetricDistortion / Normalize / Pad / CHW) against the numpy oracle,
bit-exact on the uint8 stages and to fp32 rounding after Normalize; the host decisions follow the reference's RNG call order."""
import numpy as np
import pytest
import torch

from oracle import ops as O
from s4former_amd import pipeline as P

pytestmark = pytest.mark.gpu


def _sample(seed, h, w):
    g = np.random.RandomState(seed)
    img = g.randint(0, 256, size=(h, w, 3)).astype(np.uint8)
    img[:4, :4] = 0; img[4:8, :4] = 255; img[:4, 4:8] = [10, 10, 10]          # black / white / grey: s = 0 and v = 0 branches
    seg = g.randint(0, 21, size=(h, w)).astype(np.uint8)
    seg[::7] = 255
    return img, seg


@pytest.mark.parametrize('seed', range(8))
def test_input_view_matches_the_oracle(seed):
    h, w = [(70, 90), (64, 64), (40, 100), (130, 50)][seed % 4]
    img, seg = _sample(seed, h, w)
    crop = (64, 64)
    np.random.seed(seed)
    bbox = P.random_crop_bbox((h, w), seg, crop, 0.75)
    flip = P.draw_flip(0.5)
    photo = P.draw_photometric()
    if seed == 0:
        photo = np.array([1, -20.5, 1, 1.37, 1, 1, 0.66, 1, -11], dtype=np.float32)        # every stage on, contrast first
    if seed == 1:
        photo = np.array([1, 31.2, 1, 0.55, 0, 1, 1.49, 1, 17], dtype=np.float32)          # every stage on, contrast last
    yb = (bbox[0], min(bbox[1], h), bbox[2], min(bbox[3], w))
    ref_img, ref_seg = O.input_view(img, seg, yb, flip, photo, crop)
    gi, gs, shape = P.input_view(torch.from_numpy(img).cuda(), torch.from_numpy(seg).cuda(), bbox, flip, photo, crop)
    assert shape == (yb[1] - yb[0], yb[3] - yb[2], 3)
    assert np.array_equal(gs.cpu().numpy(), ref_seg)
    d = np.abs(gi.cpu().numpy() - ref_img)
    assert d.max() <= 1e-6 * np.abs(ref_img).max() + 1e-6, (seed, d.max())
    # the uint8 stages are exact: undo Normalize and compare the integers
    mean, std = np.array(P.IMG_NORM['mean'], np.float32), np.array(P.IMG_NORM['std'], np.float32)
    back = np.rint(gi.cpu().numpy() * std[:, None, None] + mean[:, None, None])
    refb = np.rint(ref_img * std[:, None, None] + mean[:, None, None])
    assert np.array_equal(back[:, :shape[0], :shape[1]], refb[:, :shape[0], :shape[1]])


def test_hsv_round_trip_tables():
    """bgr -> hsv -> bgr through both implementations on every grey level, the primaries and a random cloud"""
    cols = [[v, v, v] for v in range(256)] + [[255, 0, 0], [0, 255, 0], [0, 0, 255], [255, 255, 0], [0, 255, 255], [255, 0, 255]]
    cols += np.random.RandomState(3).randint(0, 256, size=(4096 - len(cols), 3)).tolist()
    img = np.array(cols, dtype=np.uint8).reshape(64, 64, 3)
    photo = np.array([0, 0, 0, 1, 0, 1, 1.0, 1, 0], dtype=np.float32)                     # saturation x1 and hue +0: two round trips
    ref, _ = O.input_view(img, None, (0, 64, 0, 64), False, photo, (64, 64))
    got, _, _ = P.input_view(torch.from_numpy(img).cuda(), None, (0, 64, 0, 64), False, photo, (64, 64))
    assert np.abs(got.cpu().numpy() - ref).max() <= 1e-5


def test_semi_views_emit_student_and_teacher_of_the_same_crop():
    img, seg = _sample(5, 90, 120)
    np.random.seed(9)
    views = P.semi_views(torch.from_numpy(img).cuda(), torch.from_numpy(seg).cuda(), seg, (64, 64), tag='unsup', filename='a.jpg')
    assert [v['img_metas']['tag'] for v in views] == ['unsup_student', 'unsup_teacher']
    assert torch.equal(views[0]['gt_semantic_seg'], views[1]['gt_semantic_seg'])           # same crop + flip
    assert views[0]['img'].shape == (3, 64, 64) and views[0]['img_metas']['flip'] == views[1]['img_metas']['flip']
    np.random.seed(9)
    sup = P.semi_views(torch.from_numpy(img).cuda(), torch.from_numpy(seg).cuda(), seg, (64, 64), tag='sup')
    assert len(sup) == 1 and sup[0]['img_metas']['tag'] == 'sup' and torch.equal(sup[0]['gt_semantic_seg'], views[0]['gt_semantic_seg'])
