"""Golden of the evaluation path (sliding-window AND whole-image mode) (encoder_decoder.py:1068-1116, 1174-1216) from the reference's OWN
slide_inference / inference (container only).  The reference's code runs as written on its ema_test path
(encode_decode_ema takes the two arguments slide_inference passes; encode_decode does not - Q8), so that is the path pinned:
tiny model, 64 x 64 windows at stride (32, 48) over a 96 x 112 input (2 x 2 windows, the last of each row / column shifted
back), padded area removed, rescaled to 120 x 131.  Stored: arg-max labels, the winning probability and the top-2 margin.

Usage (build container):  python tests/golden/make_golden_slide.py"""
import json
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

from oracle import ref_harness as RH  # noqa: E402
from tests import common as C  # noqa: E402


class AttrDict(dict):
    __getattr__ = dict.__getitem__


def main():
    assert RH.available(), 'reference tree needed'
    S = C.SLIDE_CASE
    cfg = C.tiny_model_cfg(unsup_weight=1.0, ema_test=True)
    cfg['test_cfg'] = dict(mode='slide', crop_size=S['crop'], stride=S['stride'])
    ref = RH.build_reference_segmentor(cfg)
    ref.test_cfg = AttrDict(cfg['test_cfg'])
    C.load_filled(ref, S['seed_w'], S['gain'])
    ref.eval()
    imgs = C.slide_input()
    out = {}
    for flip in (False, True):
        meta = [dict(img_shape=S['img_shape'] + (3,), ori_shape=S['ori_shape'] + (3,), pad_shape=tuple(imgs.shape[2:]) + (3,), flip=flip,
                     flip_direction='horizontal') for _ in range(imgs.shape[0])]
        with torch.no_grad():
            prob = ref.inference(imgs, meta, True)
        top2 = prob.topk(2, dim=1).values
        tag = 'flip' if flip else 'plain'
        out[f'{tag}_label'] = prob.argmax(1).to(torch.uint8).numpy()
        out[f'{tag}_pmax'] = top2[:, 0].numpy().astype(np.float32)
        out[f'{tag}_margin'] = (top2[:, 0] - top2[:, 1]).numpy().astype(np.float32)
    # mode='whole' on the same path (encoder_decoder.py:1118-1147): the 64 x 64 input of tests/test_eval_gpu.py, padded area
    # removed ((60, 56) of 64 x 64), rescaled to 83 x 71
    ref.test_cfg = AttrDict(mode='whole')
    W = C.WHOLE_CASE
    wimgs, _, _ = C.make_batch(W['seed_x'], 2, 0)
    for flip in (False, True):
        meta = [dict(img_shape=W['img_shape'] + (3,), ori_shape=W['ori_shape'] + (3,), pad_shape=(64, 64, 3), flip=flip,
                     flip_direction='horizontal') for _ in range(2)]
        with torch.no_grad():
            prob = ref.inference(wimgs, meta, True)
        top2 = prob.topk(2, dim=1).values
        tag = 'whole_flip' if flip else 'whole_plain'
        out[f'{tag}_label'] = prob.argmax(1).to(torch.uint8).numpy()
        out[f'{tag}_pmax'] = top2[:, 0].numpy().astype(np.float32)
        out[f'{tag}_margin'] = (top2[:, 0] - top2[:, 1]).numpy().astype(np.float32)
    out['meta'] = json.dumps(dict(S, input_sha=C.sha(imgs), whole_input_sha=C.sha(wimgs), torch=torch.__version__))
    np.savez_compressed(os.path.join(HERE, 'eval_slide.npz'), **out)
    print('written', {k: getattr(v, 'shape', None) for k, v in out.items()})


if __name__ == '__main__':
    main()
