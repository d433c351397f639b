"""CPU tests of the host logic: registry / config loader / module construction (parameter counts and state-dict
keys of the reference), optimiser groups, poly LR, the C-ABI library loads and exports every declared symbol."""
import ctypes
import os
import re

import pytest
import torch

import s4former_amd as S
from oracle import model as OM
from oracle import ops as O
from tests import common as C

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF_CFG = '/root/reference/configs/setr'


def test_library_exports_every_declared_symbol():
    from s4former_amd import _lib
    from s4former_amd.build import build_library
    path = build_library()
    lib = ctypes.CDLL(path)
    hdr = open(os.path.join(ROOT, 'include', 's4f.h')).read()
    declared = set(re.findall(r'\b(s4f_[a-z0-9_]+)\s*\(', hdr))
    assert declared, 'no declarations found'
    for sym in declared:
        assert hasattr(lib, sym), f'{sym} declared in include/s4f.h but not exported'
    assert declared == set(_lib.EXPORTED_SYMBOLS), declared ^ set(_lib.EXPORTED_SYMBOLS)


def test_no_cpu_fallback():
    m = S.build_segmentor(C.tiny_model_cfg(unsup_weight=0))
    imgs, gt, metas = C.make_batch(1, 2, 0)
    with pytest.raises(S.S4FError):
        m.forward_train(imgs, metas, gt_semantic_seg=gt, iter=0)


def test_registry_errors():
    with pytest.raises(KeyError):
        S.build_backbone(dict(type='NoSuchBackbone'))
    with pytest.raises(TypeError):
        S.MODELS.build(['not', 'a', 'dict'])
    with pytest.raises(AssertionError):
        S.build_head(dict(type='SETRUPHead', in_channels=256, channels=128, num_classes=21, kernel_size=2,
                          norm_cfg=C.NORM_HEAD, dropout_ratio=0))


def test_tiny_model_keys_match_oracle():
    cfg = C.tiny_model_cfg(unsup_weight=1.0)
    m = S.build_segmentor(cfg)
    o = OM.oracle_from_cfg(cfg)
    mk, ok = m.state_dict(), o.state_dict()
    assert list(sorted(mk)) == list(sorted(ok))
    for k in mk:
        assert tuple(mk[k].shape) == tuple(ok[k].shape), k


@pytest.mark.skipif(not os.path.isdir(REF_CFG), reason='reference configs not present')
@pytest.mark.parametrize('fn,lr,uw', [
    ('setr_deit-base_pup_bs_8_512x512_80k_pascal_1over16_split_classic_sup.py', 0.001, 0),
    ('setr_deit-base_pup_bs_8_512x512_80k_pascal_1over16_split_classic_semi_beta_1_th_0.95_MT.py', 0.01, 1.0),
])
def test_reference_configs_load_unchanged(fn, lr, uw):
    cfg = S.Config.fromfile(os.path.join(REF_CFG, fn))
    assert cfg.model.backbone.img_size == (512, 512) and cfg.model.backbone.out_indices == (4, 7, 9, 11)
    assert cfg.model.decode_head.num_convs == 4 and cfg.model.decode_head.up_scale == 2
    assert len(cfg.model.auxiliary_head) == 4 and cfg.model.auxiliary_head[0].up_scale == 4
    assert cfg.model.ema is True and cfg.model.ema_momentum == 0.999 and cfg.model.unsup_confidence == 0.95
    assert cfg.model.unsup_weight == uw and cfg.optimizer.lr == lr
    assert cfg.lr_config.policy == 'poly' and cfg.lr_config.power == 0.9 and cfg.lr_config.min_lr == 1e-4
    m = S.build_segmentor(cfg.model, train_cfg=cfg.get('train_cfg'), test_cfg=cfg.get('test_cfg'))
    assert sum(p.numel() for p in m.parameters()) == 189430910
    assert sum(p.numel() for p in m.parameters() if p.requires_grad) == 99449961
    opt = S.build_optimizer(m, cfg.optimizer)
    assert len(opt.param_groups) == 368
    by = {g['name']: g for g in opt.param_groups}
    assert by['backbone.cls_token']['lr'] == lr and abs(by['decode_head.conv_seg.weight']['lr'] - 10 * lr) < 1e-12
    assert by['decode_head_ema.conv_seg.weight']['lr'] == lr      # teacher: listed, never updated (no grad)
    sched = S.PolyLR(opt, 80001)
    sched.step(40000)
    assert abs(by['backbone.cls_token']['lr'] - O.poly_lr(lr, 40000, 80001)) < 1e-15


def test_ours_config_flags_rejected_cleanly():
    with pytest.raises(S.S4FError):
        S.build_segmentor(C.tiny_model_cfg(use_PatchShuffle_w_Cutmix=True))
