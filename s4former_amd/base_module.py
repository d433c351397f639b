"""BaseModule / ModuleList with the `init_cfg` protocol of mmcv.runner.BaseModule (only what the hot-path classes
use: Constant / Normal / Kaiming / TruncNormal / Pretrained entries, `layer=` and `override=` selection)."""
import copy
import warnings

import torch
import torch.nn as nn


def trunc_normal_(tensor, mean=0., std=1., a=-2., b=2.):
    return nn.init.trunc_normal_(tensor, mean, std, a, b)


def constant_init(module, val, bias=0):
    if getattr(module, 'weight', None) is not None:
        nn.init.constant_(module.weight, val)
    if getattr(module, 'bias', None) is not None:
        nn.init.constant_(module.bias, bias)


def normal_init(module, mean=0, std=1, bias=0):
    if getattr(module, 'weight', None) is not None:
        nn.init.normal_(module.weight, mean, std)
    if getattr(module, 'bias', None) is not None:
        nn.init.constant_(module.bias, bias)


def kaiming_init(module, a=0, mode='fan_out', nonlinearity='relu', bias=0, distribution='normal'):
    if getattr(module, 'weight', None) is not None:
        if distribution == 'uniform':
            nn.init.kaiming_uniform_(module.weight, a=a, mode=mode, nonlinearity=nonlinearity)
        else:
            nn.init.kaiming_normal_(module.weight, a=a, mode=mode, nonlinearity=nonlinearity)
    if getattr(module, 'bias', None) is not None:
        nn.init.constant_(module.bias, bias)


def _init_by_type(m, t, cfg):
    if t == 'Constant':
        constant_init(m, cfg.get('val', 0), cfg.get('bias', 0))
    elif t == 'Normal':
        normal_init(m, cfg.get('mean', 0), cfg.get('std', 1), cfg.get('bias', 0))
    elif t == 'TruncNormal':
        if getattr(m, 'weight', None) is not None:
            trunc_normal_(m.weight, cfg.get('mean', 0), cfg.get('std', 1), cfg.get('a', -2), cfg.get('b', 2))
        if getattr(m, 'bias', None) is not None:
            nn.init.constant_(m.bias, cfg.get('bias', 0))
    elif t == 'Kaiming':
        kaiming_init(m, cfg.get('a', 0), cfg.get('mode', 'fan_out'), cfg.get('nonlinearity', 'relu'),
                     cfg.get('bias', 0), cfg.get('distribution', 'normal'))
    else:
        raise KeyError(f'unsupported init type {t}')


def initialize(module, init_cfg):
    """mmcv.cnn.initialize for the entry kinds the hot-path classes use."""
    cfgs = init_cfg if isinstance(init_cfg, list) else [init_cfg]
    for c in cfgs:
        c = dict(c)
        t = c.pop('type')
        if t == 'Pretrained':
            continue
        layers = c.pop('layer', None)
        override = c.pop('override', None)
        if layers is None and override is None:
            raise ValueError('`layer` and `override` cannot both be None in init_cfg')
        if layers is not None:
            names = [layers] if isinstance(layers, str) else list(layers)
            for m in module.modules():
                if any(b.__name__ in names for b in type(m).__mro__):
                    _init_by_type(m, t, c)
        if override is not None:
            for ov in (override if isinstance(override, list) else [override]):
                ov = dict(ov)
                name = ov.pop('name')
                if not hasattr(module, name):
                    raise RuntimeError(f'module did not have attribute {name}, but init_cfg is {ov}')
                ot = ov.pop('type', t)
                merged = dict(c)
                merged.update(ov)
                for m in getattr(module, name).modules():
                    if getattr(m, 'weight', None) is not None or getattr(m, 'bias', None) is not None:
                        _init_by_type(m, ot, merged)


class BaseModule(nn.Module):
    def __init__(self, init_cfg=None):
        super().__init__()
        self._is_init = False
        self.init_cfg = copy.deepcopy(init_cfg)

    @property
    def is_init(self):
        return self._is_init

    def init_weights(self):
        if self._is_init:
            warnings.warn(f'init_weights of {self.__class__.__name__} has been called more than once.')
            return
        if self.init_cfg:
            initialize(self, self.init_cfg)
        for m in self.children():
            if hasattr(m, 'init_weights'):
                m.init_weights()
        self._is_init = True


class ModuleList(BaseModule, nn.ModuleList):
    def __init__(self, modules=None, init_cfg=None):
        BaseModule.__init__(self, init_cfg)
        nn.ModuleList.__init__(self, modules)
