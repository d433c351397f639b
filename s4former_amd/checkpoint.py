"""Checkpoint / wire compatibility with the reference's files.

* `convert_mmcls_deit` - the key mapping of the reference's README (README.md:43-69) from an mmcls DeiT / ViT checkpoint
  (`backbone.` prefix, `attn.qkv` / `attn.proj`) to the names of mmseg's VisionTransformer
  (`attn.attn.in_proj_*` / `attn.attn.out_proj.*`); classifier head, final norm and distillation token have no counterpart in
  the SETR backbone and are dropped (the reference loads with strict=False and ignores them).
* `load_backbone_pretrained` - VisionTransformer.init_weights' Pretrained branch (mmseg/models/backbones/vit.py:369-393):
  state-dict unwrapping, bilinear position-embedding resize 14^2 -> 32^2 / 48^2, non-strict load.
* `save_checkpoint` / `load_checkpoint` / `resume` - the mmcv .pth layout the reference's runner writes and reads
  (mmcv.runner.checkpoint.save_checkpoint / load_checkpoint, IterBasedRunner.resume): a dict with `meta`
  (mmcv_version, time, iter, epoch, + caller's entries such as CLASSES / PALETTE / config text), `state_dict` (CPU tensors,
  keys `backbone.` / `decode_head.` / `auxiliary_head.{i}.` / `backbone_ema.` / `decode_head_ema.`, a leading `module.` of a
  DDP wrapper stripped on both sides) and `optimizer` (torch's optimizer.state_dict() with the SGD momentum buffers).
A file written here loads in the reference and vice versa: same pickled structure, same keys, plain torch tensors."""
import math
import os
import re
import time
from collections import OrderedDict

import torch

from ._lib import S4FError

_README_MAP = (('attn.qkv.weight', 'attn.attn.in_proj_weight'), ('attn.qkv.bias', 'attn.attn.in_proj_bias'),
               ('attn.proj.weight', 'attn.attn.out_proj.weight'), ('attn.proj.bias', 'attn.attn.out_proj.bias'))


def unwrap_state_dict(ckpt):
    """a checkpoint is either the state dict itself or a dict holding it under 'state_dict' (vit.py:376-379)"""
    if not isinstance(ckpt, dict):
        raise S4FError(f'checkpoint is a {type(ckpt).__name__}, expected a dict')
    return ckpt['state_dict'] if 'state_dict' in ckpt else ckpt


def convert_mmcls_deit(state_dict, drop_unused=True):
    """README.md:43-69: strip the `backbone.` prefix, rename the attention projections.  drop_unused also removes what the
    SETR backbone has no parameter for (`head.*`, the final `ln1.*` of mmcls' ViT, `dist_token`): the reference's non-strict
    load reports them as unexpected keys and ignores them."""
    out = OrderedDict()
    for k, v in state_dict.items():
        nk = k
        if nk.startswith('backbone.'):
            nk = nk.replace('backbone.', '')
        for old, new in _README_MAP:
            if old in nk:
                nk = nk.replace(old, new)
        if drop_unused and (nk.startswith('head.') or nk in ('ln1.weight', 'ln1.bias', 'dist_token')):
            continue
        out[nk] = v
    return out


def looks_like_mmcls(state_dict):
    return any('attn.qkv.' in k or k.startswith('backbone.') for k in state_dict)


def load_backbone_pretrained(backbone, path_or_state, convert='auto', strict=False):
    """vit.py:369-393.  `path_or_state`: a file or an already loaded (state) dict.  convert: True / False / 'auto' (apply the
    README mapping when the keys are mmcls').  Returns (missing_keys, unexpected_keys) like load_state_dict."""
    if isinstance(path_or_state, (str, os.PathLike)):
        if not os.path.exists(path_or_state):
            raise FileNotFoundError(f'checkpoint {str(path_or_state)!r} does not exist')
        ckpt = torch.load(path_or_state, map_location='cpu')
    else:
        ckpt = path_or_state
    sd = OrderedDict(unwrap_state_dict(ckpt))
    if convert is True or (convert == 'auto' and looks_like_mmcls(sd)):
        sd = convert_mmcls_deit(sd)
    if 'pos_embed' in sd and tuple(backbone.pos_embed.shape) != tuple(sd['pos_embed'].shape):
        n_src = sd['pos_embed'].shape[1] - 1
        pos_size = int(math.sqrt(n_src))
        if pos_size * pos_size != n_src:
            raise S4FError(f'pos_embed with {n_src + 1} tokens is not cls + a square grid (a distilled DeiT carries a '
                           'distillation token: use the non-distilled deit_base_p16 the reference names)')
        h, w = backbone.img_size
        sd['pos_embed'] = backbone.resize_pos_embed(sd['pos_embed'], (h // backbone.patch_size, w // backbone.patch_size),
                                                    (pos_size, pos_size), backbone.interpolate_mode, backbone.no_pos_embed)
    res = backbone.load_state_dict(sd, strict=strict)
    store = getattr(backbone, '_store', None)
    if store is not None:
        store.mark_dirty()
    return list(res.missing_keys), list(res.unexpected_keys)


def _strip_module(sd):
    return OrderedDict((re.sub(r'^module\.', '', k), v) for k, v in sd.items())


def save_checkpoint(model, filename, optimizer=None, meta=None):
    """mmcv.runner.checkpoint.save_checkpoint: {'meta', 'state_dict' (CPU), 'optimizer'}"""
    meta = dict(meta or {})
    meta.setdefault('mmcv_version', 's4former_amd')
    meta.setdefault('time', time.asctime())
    model = getattr(model, 'module', model)
    if torch.cuda.is_available():
        from .functional import join_side_streams
        join_side_streams()                       # weight updates may still be in flight on the optimiser's stream
        torch.cuda.synchronize()
    sd = OrderedDict((k, v.detach().cpu().contiguous().clone()) for k, v in _strip_module(model.state_dict()).items())
    ckpt = dict(meta=meta, state_dict=sd)
    if optimizer is not None:
        osd = optimizer.state_dict()
        for st in osd.get('state', {}).values():
            for k, v in list(st.items()):
                if torch.is_tensor(v):
                    st[k] = v.detach().cpu()
        ckpt['optimizer'] = osd
    d = os.path.dirname(os.path.abspath(filename))
    os.makedirs(d, exist_ok=True)
    tmp = f'{filename}.tmp.{os.getpid()}'
    torch.save(ckpt, tmp)
    os.replace(tmp, filename)                     # never a half-written checkpoint under the final name
    return filename


def load_checkpoint(model, filename, map_location='cpu', strict=False, revise_keys=((r'^module\.', ''),)):
    """mmcv.runner.checkpoint.load_checkpoint: loads `state_dict` (or the dict itself) into the model; returns the checkpoint"""
    if not os.path.exists(filename):
        raise FileNotFoundError(f'checkpoint {filename!r} does not exist')
    ckpt = torch.load(filename, map_location=map_location)
    sd = OrderedDict(unwrap_state_dict(ckpt))
    for pat, rep in revise_keys:
        sd = OrderedDict((re.sub(pat, rep, k), v) for k, v in sd.items())
    target = getattr(model, 'module', model)
    res = target.load_state_dict(sd, strict=strict)
    for st in (getattr(target, 'student_store', None), getattr(target, 'teacher_store', None)):
        if st is not None:
            st.mark_dirty()                       # bf16 operand shadows follow the new masters at the next step
    if strict is False and (res.missing_keys or res.unexpected_keys):
        import warnings
        warnings.warn(f'load_checkpoint: missing {list(res.missing_keys)[:5]}..., unexpected {list(res.unexpected_keys)[:5]}...')
    return ckpt


def resume(model, optimizer, filename, map_location='cpu'):
    """IterBasedRunner.resume: weights + optimizer state; returns meta (with 'iter' / 'epoch').  The arenas must exist for the
    momentum buffers to land in them: call model.ensure_engine(device) first when the model is already on the GPU."""
    ckpt = load_checkpoint(model, filename, map_location=map_location, strict=True)
    if optimizer is not None and 'optimizer' in ckpt:
        optimizer.load_state_dict(ckpt['optimizer'])
    return ckpt.get('meta', {})
