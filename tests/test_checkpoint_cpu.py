"""Checkpoint / wire compatibility, CPU only: the README's mmcls -> mmseg key mapping, the position-embedding
resize on load (reference vit.py:381-393, 447-477), the mmcv .pth layout with the EMA prefixes and the optimizer state."""
import os

import pytest
import torch
import torch.nn.functional as F

import s4former_amd as S
from oracle import ref_harness as RH
from s4former_amd import checkpoint as CK
from tests import common as C


def _mmcls_deit(embed=256, layers=4, grid=14, seed=3):
    """a checkpoint with the key names and nesting of mmcls' deit_base_p16 (README.md:43-69), at toy width"""
    g = torch.Generator().manual_seed(seed)
    r = lambda *s: torch.randn(*s, generator=g) * 0.05
    sd = {'backbone.cls_token': r(1, 1, embed), 'backbone.pos_embed': r(1, grid * grid + 1, embed),
          'backbone.patch_embed.projection.weight': r(embed, 3, 16, 16), 'backbone.patch_embed.projection.bias': r(embed),
          'backbone.ln1.weight': r(embed), 'backbone.ln1.bias': r(embed),
          'head.layers.head.weight': r(1000, embed), 'head.layers.head.bias': r(1000)}
    for i in range(layers):
        p = f'backbone.layers.{i}.'
        sd.update({p + 'ln1.weight': r(embed), p + 'ln1.bias': r(embed), p + 'ln2.weight': r(embed), p + 'ln2.bias': r(embed),
                   p + 'attn.qkv.weight': r(3 * embed, embed), p + 'attn.qkv.bias': r(3 * embed),
                   p + 'attn.proj.weight': r(embed, embed), p + 'attn.proj.bias': r(embed),
                   p + 'ffn.layers.0.0.weight': r(4 * embed, embed), p + 'ffn.layers.0.0.bias': r(4 * embed),
                   p + 'ffn.layers.1.weight': r(embed, 4 * embed), p + 'ffn.layers.1.bias': r(embed)})
    return {'state_dict': sd, 'meta': {}}


def test_readme_key_mapping_and_pos_embed_resize(tmp_path):
    ck = _mmcls_deit()
    path = str(tmp_path / 'deit_toy.pth')
    torch.save(ck, path)
    conv = CK.convert_mmcls_deit(ck['state_dict'])
    assert 'layers.0.attn.attn.in_proj_weight' in conv and 'layers.3.attn.attn.out_proj.bias' in conv
    assert not any(k.startswith(('backbone.', 'head.')) or 'qkv' in k for k in conv) and 'ln1.weight' not in conv
    # init_cfg = Pretrained -> init_weights loads it, teacher == student (configs/setr/*:137-241 use ONE backbone dict)
    cfg = C.tiny_model_cfg(unsup_weight=1.0)
    for k in ('backbone', 'backbone_ema'):
        cfg[k]['init_cfg'] = dict(type='Pretrained', checkpoint=path)
    model = S.build_segmentor(cfg)
    model.init_weights()
    sd = model.state_dict()
    src = ck['state_dict']
    for pre in ('backbone.', 'backbone_ema.'):
        assert torch.equal(sd[pre + 'layers.2.attn.attn.in_proj_weight'], src['backbone.layers.2.attn.qkv.weight'])
        assert torch.equal(sd[pre + 'layers.0.attn.attn.out_proj.bias'], src['backbone.layers.0.attn.proj.bias'])
        assert torch.equal(sd[pre + 'patch_embed.projection.weight'], src['backbone.patch_embed.projection.weight'])
        # 14 x 14 -> 4 x 4 position grid: cls token kept, the grid resized bilinearly (align_corners=False)
        pe = src['backbone.pos_embed']
        want = F.interpolate(pe[:, 1:].reshape(1, 14, 14, -1).permute(0, 3, 1, 2), size=(4, 4), mode='bilinear', align_corners=False)
        want = torch.cat((pe[:, :1], want.flatten(2).transpose(1, 2)), 1)
        assert torch.equal(sd[pre + 'pos_embed'], want)
    if RH.available():          # build container: the reference's own resize_pos_embed
        RH.load_reference()
        import sys
        ref_vit = sys.modules['mmseg.models.backbones.vit'].VisionTransformer
        ref = ref_vit.resize_pos_embed(src['backbone.pos_embed'], (4, 4), (14, 14), 'bilinear')
        assert torch.equal(ref, sd['backbone.pos_embed'])
    # a mistyped path must raise (the reference's CheckpointLoader does), not train from random weights
    cfg['backbone']['init_cfg'] = dict(type='Pretrained', checkpoint=str(tmp_path / 'missing.pth'))
    with pytest.raises(FileNotFoundError):
        S.build_segmentor(cfg).init_weights()


def test_mmcv_checkpoint_layout_round_trip(tmp_path):
    model = S.build_segmentor(C.tiny_model_cfg(unsup_weight=1.0))
    C.load_filled(model, 11, 3.0)
    opt = S.build_optimizer(model, dict(type='SGD', lr=0.01, momentum=0.9, weight_decay=0.0,
                                        paramwise_cfg=dict(custom_keys={'head': dict(lr_mult=10.)})))
    path = str(tmp_path / 'iter_4.pth')
    S.save_checkpoint(model, path, optimizer=opt, meta=dict(iter=4, epoch=1, CLASSES=('a', 'b')))
    ck = torch.load(path, map_location='cpu')
    assert set(ck) == {'meta', 'state_dict', 'optimizer'} and ck['meta']['iter'] == 4 and ck['meta']['CLASSES'] == ('a', 'b')
    keys = list(ck['state_dict'])
    for pre in ('backbone.', 'decode_head.', 'auxiliary_head.0.', 'auxiliary_head.3.', 'backbone_ema.', 'decode_head_ema.'):
        assert any(k.startswith(pre) for k in keys), pre
    assert 'decode_head.up_convs.0.0.bn.running_var' in keys and 'decode_head_ema.up_convs.3.0.bn.num_batches_tracked' in keys
    assert len(ck['optimizer']['param_groups']) == len(list(model.named_parameters()))
    # a DDP-wrapped writer's 'module.' prefix is stripped on load (mmcv revise_keys)
    ck['state_dict'] = {'module.' + k: v for k, v in ck['state_dict'].items()}
    torch.save(ck, path)
    model2 = S.build_segmentor(C.tiny_model_cfg(unsup_weight=1.0))
    opt2 = S.build_optimizer(model2, dict(type='SGD', lr=0.5, momentum=0.9, weight_decay=0.0))
    meta = S.resume(model2, opt2, path)
    assert meta['iter'] == 4
    for (k, a), (_, b) in zip(model.state_dict().items(), model2.state_dict().items()):
        assert torch.equal(a, b), k
    assert opt2.param_groups[0]['lr'] == 0.01
    with pytest.raises(FileNotFoundError):
        S.load_checkpoint(model2, str(tmp_path / 'nope.pth'))
