"""Minimal loader for the reference's mmcv-style config files: plain Python files of dict literals,
`_base_` inheritance relative to the file, recursive dict merge (child over base, lists replace, `_delete_`),
attribute access.  configs/setr/*.py of the reference load unchanged."""
import copy
import os


class ConfigDict(dict):
    def __getattr__(self, name):
        try:
            return self[name]
        except KeyError:
            raise AttributeError(name)

    def __setattr__(self, name, value):
        self[name] = value


def _to_cd(obj):
    if isinstance(obj, dict):
        return ConfigDict({k: _to_cd(v) for k, v in obj.items()})
    if isinstance(obj, list):
        return [_to_cd(v) for v in obj]
    if isinstance(obj, tuple):
        return tuple(_to_cd(v) for v in obj)
    return obj


def _merge(base, child):
    out = copy.deepcopy(base)
    for k, v in child.items():
        if isinstance(v, dict) and k in out and isinstance(out[k], dict) and not v.get('_delete_', False):
            out[k] = _merge(out[k], v)
        else:
            v = copy.deepcopy(v)
            if isinstance(v, dict):
                v.pop('_delete_', None)
            out[k] = v
    return out


def _load_file(path):
    path = os.path.abspath(path)
    ns = {'__file__': path}
    with open(path) as f:
        exec(compile(f.read(), path, 'exec'), ns)
    cfg = {k: v for k, v in ns.items() if not k.startswith('__') and not callable(v) and not isinstance(v, type(os))}
    bases = cfg.pop('_base_', [])
    if isinstance(bases, str):
        bases = [bases]
    merged = {}
    for b in bases:
        bcfg = _load_file(os.path.join(os.path.dirname(path), b))
        dup = set(merged) & set(bcfg)
        if dup:
            raise KeyError(f'Duplicate key is not allowed among bases: {sorted(dup)}')
        merged.update(bcfg)
    return _merge(merged, cfg)


class Config:
    def __init__(self, cfg_dict=None, filename=None):
        object.__setattr__(self, '_cfg_dict', _to_cd(cfg_dict or {}))
        object.__setattr__(self, 'filename', filename)

    @staticmethod
    def fromfile(filename):
        return Config(_load_file(filename), filename)

    def merge_from_dict(self, options):
        """--cfg-options k.a.b=v semantics"""
        nested = {}
        for full, v in options.items():
            d = nested
            keys = full.split('.')
            for k in keys[:-1]:
                d = d.setdefault(k, {})
            d[keys[-1]] = v
        object.__setattr__(self, '_cfg_dict', _to_cd(_merge(self._cfg_dict, nested)))

    def __getattr__(self, name):
        return getattr(self._cfg_dict, name)

    def __getitem__(self, name):
        return self._cfg_dict[name]

    def __contains__(self, name):
        return name in self._cfg_dict

    def get(self, name, default=None):
        return self._cfg_dict.get(name, default)

    def to_dict(self):
        return copy.deepcopy(dict(self._cfg_dict))
