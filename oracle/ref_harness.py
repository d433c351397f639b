"""ORACLE support (test infrastructure, container-only): run the REFERENCE'S OWN hot-path Python files.

The reference (a fork of mmsegmentation 0.26) cannot be imported as a package here: mmcv and cv2 are not installed,
`collections.Mapping` is gone in Python 3.10, and `mmseg.utils` misses a re-export.  This module
loads, BY PATH and UNMODIFIED, the reference files that make up the hot path, after placing a thin stand-in
for the handful of mmcv symbols they import into sys.modules.  The stand-ins are torch.nn one-liners with mmcv's
documented behaviour (Registry.build, ConvModule = conv(no bias) -> norm -> ReLU with kaiming init,
MultiheadAttention = identity + nn.MultiheadAttention(...)[0] with batch_first transposes, FFN = identity +
Linear-GELU-Linear, build_norm_layer LN -> 'ln<postfix>', BN/SyncBN -> 'bn' mapped to BatchNorm2d as
revert_sync_batchnorm does for single-process runs).

Nothing from /root/reference is copied: when the directory is absent (the GPU box) `available()` is False and
every user of this module skips.  It is used to (i) validate oracle/model.py and (ii) generate the golden
vectors under tests/golden/ (tests/golden/make_golden.py).
"""
import collections
import collections.abc
import copy
import importlib.util
import os
import sys
import types
import warnings

import torch
import torch.nn as nn

REF = os.environ.get('S4F_REFERENCE_DIR', '/root/reference')
_loaded = None


def available():
    return os.path.isdir(os.path.join(REF, 'mmseg', 'models', 'segmentors'))


def _mod(name, **attrs):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    m.__path__ = []
    sys.modules[name] = m
    parent, _, child = name.rpartition('.')
    if parent and parent in sys.modules:
        setattr(sys.modules[parent], child, m)
    return m


class _Registry:
    def __init__(self, name, parent=None, **kw):
        self.name = name
        self.d = {} if parent is None else parent.d

    def register_module(self, name=None, force=False, module=None):
        if module is not None:
            self.d[name or module.__name__] = module
            return module

        def deco(cls):
            self.d[name or cls.__name__] = cls
            return cls
        return deco

    def build(self, cfg, default_args=None):
        cfg = dict(cfg)
        for k, v in (default_args or {}).items():
            cfg.setdefault(k, v)
        return self.d[cfg.pop('type')](**cfg)


class _BaseModule(nn.Module):
    def __init__(self, init_cfg=None):
        super().__init__()
        self.init_cfg = copy.deepcopy(init_cfg)

    def init_weights(self):
        pass


class _ModuleList(_BaseModule, nn.ModuleList):
    def __init__(self, modules=None, init_cfg=None):
        _BaseModule.__init__(self, init_cfg)
        nn.ModuleList.__init__(self, modules)


def _build_norm_layer(cfg, num_features, postfix=''):
    cfg = dict(cfg)
    t = cfg.pop('type')
    cfg.pop('requires_grad', None)
    cfg.setdefault('eps', 1e-5)
    if t == 'LN':
        return 'ln' + str(postfix), nn.LayerNorm(num_features, **cfg)
    if t in ('BN', 'SyncBN'):
        return 'bn' + str(postfix), nn.BatchNorm2d(num_features, **cfg)
    raise KeyError(t)


class _ConvModule(nn.Module):
    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, norm_cfg=None, act_cfg=dict(type='ReLU'), **kw):
        super().__init__()
        self.with_norm = norm_cfg is not None
        self.conv = nn.Conv2d(in_channels, out_channels, kernel_size, stride, padding, bias=not self.with_norm)
        if self.with_norm:
            self.norm_name, n = _build_norm_layer(norm_cfg, out_channels)
            self.add_module(self.norm_name, n)
        self.activate = nn.ReLU(inplace=True) if act_cfg else None
        nn.init.kaiming_normal_(self.conv.weight, a=0, mode='fan_out', nonlinearity='relu')

    def forward(self, x):
        x = self.conv(x)
        if self.with_norm:
            x = getattr(self, self.norm_name)(x)
        return self.activate(x) if self.activate is not None else x


class _MultiheadAttention(_BaseModule):
    """mmcv brick + the out-of-tree `.self_attn` store the reference reads (Q7)"""

    def __init__(self, embed_dims, num_heads, attn_drop=0., proj_drop=0., dropout_layer=None, init_cfg=None, batch_first=False, **kw):
        super().__init__(init_cfg)
        self.batch_first = batch_first
        self.attn = nn.MultiheadAttention(embed_dims, num_heads, attn_drop, **kw)
        self.proj_drop = nn.Dropout(proj_drop)

    def forward(self, query, key=None, value=None, identity=None, attn_mask=None, **kw):
        key = query if key is None else key
        value = key if value is None else value
        identity = query if identity is None else identity
        if self.batch_first:
            query, key, value = (t.transpose(0, 1) for t in (query, key, value))
        out, self.self_attn = self.attn(query=query, key=key, value=value, attn_mask=attn_mask)
        if self.batch_first:
            out = out.transpose(0, 1)
        return identity + self.proj_drop(out)


class _FFN(_BaseModule):
    def __init__(self, embed_dims, feedforward_channels, num_fcs=2, act_cfg=None, ffn_drop=0., dropout_layer=None,
                 add_identity=True, init_cfg=None, **kw):
        super().__init__(init_cfg)
        assert num_fcs == 2 and dropout_layer is None
        self.layers = nn.Sequential(nn.Sequential(nn.Linear(embed_dims, feedforward_channels), nn.GELU(), nn.Dropout(ffn_drop)),
                                    nn.Linear(feedforward_channels, embed_dims), nn.Dropout(ffn_drop))

    def forward(self, x, identity=None):
        return (x if identity is None else identity) + self.layers(x)


def _passthru(*a, **kw):
    return lambda f: f


def _load(name, relpath):
    spec = importlib.util.spec_from_file_location(name, os.path.join(REF, relpath))
    m = importlib.util.module_from_spec(spec)
    sys.modules[name] = m
    parent, _, child = name.rpartition('.')
    if parent in sys.modules:
        setattr(sys.modules[parent], child, m)
    spec.loader.exec_module(m)
    return m


def load_reference():
    """returns the registry holding the reference's EncoderDecoder / VisionTransformer / SETRUPHead / CrossEntropyLoss"""
    global _loaded
    if _loaded is not None:
        return _loaded
    if not available():
        raise RuntimeError('reference tree not present')
    warnings.filterwarnings('ignore')
    collections.Mapping = collections.abc.Mapping
    collections.Sequence = collections.abc.Sequence
    MODELS, ATT = _Registry('models'), _Registry('attention')
    _mod('mmcv', load=lambda *a, **k: None)
    _mod('mmcv.cnn', MODELS=MODELS, build_norm_layer=_build_norm_layer, build_conv_layer=lambda cfg, *a, **kw: nn.Conv2d(*a, **kw),
         ConvModule=_ConvModule)
    _mod('mmcv.cnn.bricks')
    _mod('mmcv.cnn.bricks.registry', ATTENTION=ATT)
    _mod('mmcv.cnn.bricks.transformer', FFN=_FFN, MultiheadAttention=_MultiheadAttention)
    _mod('mmcv.cnn.utils')
    _mod('mmcv.cnn.utils.weight_init',
         constant_init=lambda m, val, bias=0: (nn.init.constant_(m.weight, val), nn.init.constant_(m.bias, bias)),
         kaiming_init=lambda m, mode='fan_out', bias=0., **k: (nn.init.kaiming_normal_(m.weight, mode=mode, nonlinearity='relu'),
                                                               nn.init.constant_(m.bias, bias)),
         trunc_normal_=nn.init.trunc_normal_)
    _mod('mmcv.runner', BaseModule=_BaseModule, ModuleList=_ModuleList, CheckpointLoader=None, load_state_dict=None,
         auto_fp16=_passthru, force_fp32=_passthru)
    _mod('mmcv.runner.base_module', BaseModule=_BaseModule)
    _mod('mmcv.utils', Registry=_Registry, to_2tuple=lambda x: x if isinstance(x, tuple) else (x, x))
    for n in ('mmseg', 'mmseg.models', 'mmseg.models.utils', 'mmseg.models.losses', 'mmseg.models.backbones',
              'mmseg.models.decode_heads', 'mmseg.models.segmentors'):
        _mod(n)
    _load('mmseg.ops', 'mmseg/ops/wrappers.py')
    _mod('mmseg.core', add_prefix=_load('mmseg.core_misc', 'mmseg/core/utils/misc.py').add_prefix, build_pixel_sampler=None)
    gu = _load('mmseg.generate_unsup_data', 'mmseg/utils/generate_unsup_data.py')
    _mod('mmseg.utils', get_root_logger=lambda *a, **k: None,
         **{k: getattr(gu, k) for k in dir(gu) if k.startswith(('generate_', 'cut_mix'))})
    _load('mmseg.models.builder', 'mmseg/models/builder.py')
    emb = _load('mmseg.models.utils.embed', 'mmseg/models/utils/embed.py')
    _load('mmseg.models.utils.structual_utils', 'mmseg/models/utils/structual_utils.py')
    sys.modules['mmseg.models.utils'].PatchEmbed = emb.PatchEmbed
    _load('mmseg.models.losses.utils', 'mmseg/models/losses/utils.py')
    acc = _load('mmseg.models.losses.accuracy', 'mmseg/models/losses/accuracy.py')
    sys.modules['mmseg.models.losses'].accuracy = acc.accuracy
    _load('mmseg.models.losses.cross_entropy_loss', 'mmseg/models/losses/cross_entropy_loss.py')
    _load('mmseg.models.backbones.vit', 'mmseg/models/backbones/vit.py')
    _load('mmseg.models.decode_heads.decode_head', 'mmseg/models/decode_heads/decode_head.py')
    _load('mmseg.models.decode_heads.setr_up_head', 'mmseg/models/decode_heads/setr_up_head.py')
    _load('mmseg.models.segmentors.base', 'mmseg/models/segmentors/base.py')
    _load('mmseg.models.segmentors.encoder_decoder', 'mmseg/models/segmentors/encoder_decoder.py')
    _loaded = MODELS
    return MODELS


def build_reference_segmentor(model_cfg):
    """model_cfg: mmseg-style dict (type='EncoderDecoder', ...). SyncBN is mapped to BN (single process)."""
    MODELS = load_reference()
    cfg = copy.deepcopy(dict(model_cfg))
    cfg.pop('plain_mt_pseudo_loss', None)
    cfg.setdefault('train_cfg', dict())
    cfg.setdefault('test_cfg', dict(mode='whole'))
    for k in ('backbone', 'backbone_ema'):
        if cfg.get(k):
            cfg[k] = dict(cfg[k])
            cfg[k].pop('init_cfg', None)
    return MODELS.build(cfg)
