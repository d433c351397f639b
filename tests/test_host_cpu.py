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
    ('setr_deit-base_pup_bs_8_512x512_80k_pascal_1over16_split_classic_semi_beta_1_th_0.95_MT_w_ours.py', 0.001, 1.0),
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
    if fn.endswith('_w_ours.py'):      # the paper's method: PASA + CutMix / PatchShuffle + negative class ranking
        assert m.attn_mask_seperate_head and m.adaptive_attn_mask and m.attn_mask_weight == 5
        assert m.use_PatchShuffle_w_Cutmix and m.PatchMix_N == 8 and m.negative_class_ranking
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


def test_out_of_scope_flags_rejected_cleanly():
    # (the flags of configs/setr/..._MT_w_ours.py are built; the other in-model augmentations / NCR modes are outside SURVEY §8)
    for flags in (dict(use_CutMix=True), dict(use_ClassMix=True), dict(use_PatchShuffle=True), dict(unimatch=True),
                  dict(negative_class_ranking=True, negative_class_ranking_mode='both', attn_mask_seperate_head=True),
                  dict(use_PatchShuffle_w_Cutmix=True)):            # (needs attn_mask_seperate_head)
        with pytest.raises(S.S4FError):
            S.build_segmentor(C.tiny_model_cfg(**flags))


def test_strong_augmentation_decisions_follow_the_reference_rng_order():
    """augment.draw_strong_aug (product) == oracle.ops.draw_strong_aug (pinned to the reference by the mt_ours golden), and
    the token un-shuffle maps == the restated _repatchmix_inputs, incl. their adjoint"""
    import numpy as np
    import torch
    from s4former_amd import augment as A
    for seed in range(6):
        np.random.seed(seed); torch.manual_seed(seed)
        b1, p1 = A.draw_strong_aug(3, 64, 64, 0.5, 2, 0.5, 32)
        np.random.seed(seed); torch.manual_seed(seed)
        b2, p2 = O.draw_strong_aug(3, (64, 64), 0.5, 2, 0.5, 32)
        assert b1.tolist() == [list(b) for b in b2] and p1.tolist() == [p.tolist() for p in p2]
        tok = torch.randn(3, 17, 8)
        fwd, bwd = A.token_unshuffle_maps(p1, 4, 2)
        got = tok.reshape(-1, 8)[torch.from_numpy(fwd).long()].reshape(3, 17, 8)
        assert torch.equal(got[:, 1:], O.repatchmix_tokens(tok[:, 1:], p2, 2)) and torch.equal(got[:, 0], tok[:, 0])
        assert torch.equal(got.reshape(-1, 8)[torch.from_numpy(bwd).long()].reshape(3, 17, 8), tok)


def test_shipped_tuning_table_is_well_formed():
    """s4former_amd/tuned_gfx950.json: every key is a GEMM signature the host code can produce, every value a (tile variant,
    split-K) pair the library accepts - a malformed entry would only show up as a slow or failing launch on the GPU box"""
    import json
    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 's4former_amd', 'tuned_gfx950.json')
    table = json.load(open(path))
    assert len(table) >= 60
    for k, v in table.items():
        key = eval(k, {'__builtins__': {}}, {})
        assert isinstance(key, tuple) and isinstance(v, list) and len(v) == 2, k
        hint, sk = v
        assert hint in (1, 2, 3, 4, 8, 9, 10) and isinstance(sk, int) and sk >= 1, (k, v)
        if key[0] == 'wgrad_grouped':
            assert 2 <= len(key) <= 5 and all(len(t) == 3 for t in key[1:]), k
            continue
        assert len(key) == 14, k
        a_mode, b_mode, M, N, K = key[:5]
        assert a_mode in (0, 1, 2) and b_mode in (0, 1, 2, 3, 4) and min(M, N, K) > 0, k
        atomic = key[9]
        assert atomic or sk == 1, (k, v)                     # split-K needs the atomic fp32 output
        if hint in (3, 4, 10):
            assert N % 256 == 0 or a_mode == 1, (k, v)       # 256-wide tiles
        if hint in (8, 9):
            assert N % 192 == 0, (k, v)


def test_generic_workspace_query_of_the_c_abi():
    """s4f_workspace_bytes(op, extents) - a pure host function of the library (no device needed): the attention
    backward's scratch equals the dedicated query, the BatchNorm sums are 2 C floats, a split-K output is M x N floats, an
    unknown op / a wrong number of extents is an error (return -1 -> S4FError in the binding)."""
    from s4former_amd import _lib
    from s4former_amd import kernels as K
    lib = _lib.load()
    assert K.workspace_bytes(K.WS_ATTENTION_BWD, 16, 1025, 12) == int(lib.s4f_attention_bwd_ws_bytes(16, 1025, 12)) > 200_000_000
    assert K.workspace_bytes(K.WS_BN_SUMS, 256) == 2 * 256 * 4
    assert K.workspace_bytes(K.WS_GEMM_SPLITK, 8192, 256) == 8192 * 256 * 4
    for bad in ((99, 1), (K.WS_BN_SUMS, 1, 2), (K.WS_ATTENTION_BWD, 16, 1025)):
        with pytest.raises(_lib.S4FError):
            K.workspace_bytes(*bad)
