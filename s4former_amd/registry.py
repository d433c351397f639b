"""Registry / plugin API of the hot path (mirrors the mmcv Registry use in reference mmseg/models/builder.py:8-49):
configs name classes by `type=` and build_* instantiate them with the dict's remaining keys as kwargs.
Error convention as in mmcv: KeyError for an unknown type, TypeError for a non-dict cfg."""
import copy


class Registry:
    def __init__(self, name):
        self._name = name
        self._module_dict = {}

    @property
    def name(self):
        return self._name

    @property
    def module_dict(self):
        return self._module_dict

    def __contains__(self, key):
        return key in self._module_dict

    def get(self, key):
        return self._module_dict.get(key)

    def register_module(self, name=None, force=False, module=None):
        if module is not None:
            self._register(module, name, force)
            return module

        def _deco(cls):
            self._register(cls, name, force)
            return cls
        return _deco

    def _register(self, cls, name, force):
        names = [name or cls.__name__] if not isinstance(name, (list, tuple)) else list(name)
        for n in names:
            if not force and n in self._module_dict:
                raise KeyError(f'{n} is already registered in {self._name}')
            self._module_dict[n] = cls

    def build(self, cfg, default_args=None):
        if not isinstance(cfg, dict):
            raise TypeError(f'cfg must be a dict, but got {type(cfg)}')
        if 'type' not in cfg and not (default_args and 'type' in default_args):
            raise KeyError(f'`cfg` or `default_args` must contain the key "type", but got {cfg}')
        args = copy.deepcopy(dict(cfg))
        if default_args:
            for k, v in default_args.items():
                args.setdefault(k, v)
        obj_type = args.pop('type')
        if isinstance(obj_type, str):
            cls = self.get(obj_type)
            if cls is None:
                raise KeyError(f'{obj_type} is not in the {self._name} registry')
        elif isinstance(obj_type, type):
            cls = obj_type
        else:
            raise TypeError(f'type must be a str or valid type, but got {type(obj_type)}')
        return cls(**args)


MODELS = Registry('models')
BACKBONES = MODELS
NECKS = MODELS
HEADS = MODELS
LOSSES = MODELS
SEGMENTORS = MODELS


def build_backbone(cfg):
    return BACKBONES.build(cfg)


def build_neck(cfg):
    return NECKS.build(cfg)


def build_head(cfg):
    return HEADS.build(cfg)


def build_loss(cfg):
    return LOSSES.build(cfg)


def build_segmentor(cfg, train_cfg=None, test_cfg=None):
    """reference mmseg/models/builder.py:38-49"""
    cfg = dict(cfg)
    assert cfg.get('train_cfg') is None or train_cfg is None, 'train_cfg specified in both outer field and model field'
    assert cfg.get('test_cfg') is None or test_cfg is None, 'test_cfg specified in both outer field and model field'
    return SEGMENTORS.build(cfg, default_args=dict(train_cfg=train_cfg, test_cfg=test_cfg))
