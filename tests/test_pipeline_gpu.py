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


# ------------------------------------------------------------------------------------------------ round 3: Resize on the device, pinned pipeline
import os  # noqa: E402

from tests import common as C  # noqa: E402
from tests.test_pipeline_cpu import GOLD, host_decisions  # noqa: E402


@pytest.mark.parametrize('name', sorted(C.PIPELINE_CASES))
def test_device_pipeline_reproduces_the_reference_pipeline(name):
    """P.semi_views (Resize -> RandomCrop -> RandomFlip -> [MultiBranch] PhotoMetricDistortion -> Normalize -> Pad, one kernel
    launch per view, the resized image never built) against the outputs of the reference's own Compose / MultiBranch
    (tests/golden/pipeline.npz): labels exact, images to fp32 rounding"""
    z = np.load(GOLD)
    kw = C.PIPELINE_CASES[name]
    img, seg = C.pipeline_sample(kw['seed'], *kw['hw'])
    np.random.seed(kw['seed'] + 1000)
    views = P.semi_views(torch.from_numpy(img).cuda(), torch.from_numpy(seg).cuda(), seg, tuple(kw['crop']), cat_max_ratio=0.75,
                         flip_prob=0.5, tag=kw['tag'], resize=dict(img_scale=tuple(kw['img_scale']), ratio_range=tuple(kw['ratio_range'])))
    assert [v['img_metas']['tag'] for v in views] == (['sup'] if kw['tag'] == 'sup' else ['unsup_student', 'unsup_teacher'])
    for i, v in enumerate(views):
        meta = z[f'{name}_v{i}_meta']
        assert tuple(v['img_metas']['img_shape'][:2]) == (int(meta[0]), int(meta[1])) and v['img_metas']['flip'] == bool(meta[2])
        assert np.array_equal(v['gt_semantic_seg'][0].cpu().numpy(), z[f'{name}_v{i}_seg']), f'{name} view {i}: labels'
        d = np.abs(v['img'].cpu().numpy() - z[f'{name}_v{i}_img']).max()
        assert d <= 1e-5, f'{name} view {i}: image differs by {d}'


@pytest.mark.parametrize('hw,to', [((37, 53), (74, 106)), ((37, 53), (50, 70)), ((37, 53), (20, 31)), ((40, 60), (20, 30)), ((64, 48), (200, 150)),
                                   ((33, 47), (33, 47)), ((50, 50), (7, 9))])
def test_device_resize_matches_the_oracle_bit_for_bit(hw, to):
    """the whole resized image through the kernel (crop window = everything, no flip / photometric): uint8 values identical to
    the numpy restatement of cv2.INTER_LINEAR (image) / INTER_NEAREST (labels)"""
    img, seg = C.pipeline_sample(11, *hw)
    gi, gs, shape = P.input_view(torch.from_numpy(img).cuda(), torch.from_numpy(seg).cuda(), (0, to[0], 0, to[1]), False, P.NO_PHOTOMETRIC,
                                 to, norm=dict(mean=(0, 0, 0), std=(1, 1, 1), to_rgb=False), resize_to=to)
    ref = O.cv_resize_linear_u8(img, to)
    assert np.array_equal(gi.cpu().numpy().transpose(1, 2, 0), ref.astype(np.float32))
    assert np.array_equal(gs.cpu().numpy(), O.cv_resize_nearest(seg, to))


def test_reference_known_answers_on_the_device():
    """tests/test_data/test_transform.py restated for s4f_input_view: :152-187 flip of a flip is the identity, :190-216 the crop
    has the requested shape, :296-320 Normalize == (img[..., ::-1] - mean) / std"""
    img, seg = C.pipeline_sample(21, 80, 100)
    ident = dict(mean=(0, 0, 0), std=(1, 1, 1), to_rgb=False)
    di, ds = torch.from_numpy(img).cuda(), torch.from_numpy(seg).cuda()
    for direction in ('horizontal', 'vertical'):
        f1, s1, _ = P.input_view(di, ds, (0, 80, 0, 100), True, P.NO_PHOTOMETRIC, (80, 100), norm=ident, flip_direction=direction)
        back = f1.permute(1, 2, 0).to(torch.uint8).contiguous()
        f2, s2, _ = P.input_view(back, s1, (0, 80, 0, 100), True, P.NO_PHOTOMETRIC, (80, 100), norm=ident, flip_direction=direction)
        assert np.array_equal(f2.cpu().numpy().transpose(1, 2, 0), img.astype(np.float32)) and np.array_equal(s2.cpu().numpy(), seg)
        assert not np.array_equal(f1.cpu().numpy().transpose(1, 2, 0), img.astype(np.float32))
    h, w = img.shape[:2]
    np.random.seed(0)
    bbox = P.random_crop_bbox((h, w), seg, (h - 20, w - 20), 1.0)
    ci, cs, shape = P.input_view(di, ds, bbox, False, P.NO_PHOTOMETRIC, (h - 20, w - 20), norm=ident)
    assert shape[:2] == (h - 20, w - 20) and tuple(ci.shape[1:]) == (h - 20, w - 20) and tuple(cs.shape) == (h - 20, w - 20)
    assert np.array_equal(ci.cpu().numpy().transpose(1, 2, 0), img[bbox[0]:bbox[1], bbox[2]:bbox[3]].astype(np.float32))
    ni, _, _ = P.input_view(di, None, (0, h, 0, w), False, P.NO_PHOTOMETRIC, (h, w))
    mean, std = np.array(P.IMG_NORM['mean']), np.array(P.IMG_NORM['std'])
    assert np.allclose(ni.cpu().numpy().transpose(1, 2, 0), (img[..., ::-1] - mean) / std, rtol=1e-5, atol=1e-5)
