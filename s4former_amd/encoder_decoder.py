"""EncoderDecoder segmentor of S4Former (mean-teacher semi-supervised SETR) with the reference's registry name,
constructor kwargs, method names, loss keys and state-dict prefixes
(reference mmseg/models/segmentors/encoder_decoder.py:25-163,386-687,875-934,1044-1066 and base.py:155-274),
driving the MI355X HIP kernels: student + teacher replicas live in flat arenas (params.ParamStore), the EMA
update is one launch, the teacher's pseudo-labels come from one fused kernel.

Scope: the supervised branch, the mean-teacher EMA, teacher pseudo-labels with confidence threshold,
the pseudo-label CE (`compute_pseudo_loss`), the PASA attention bias (rank-1, never materialised) and the "ours" additions
of configs/setr/..._MT_w_ours.py: CutMix + PatchShuffle of the student images between the masked and the plain student pass
(use_PatchShuffle_w_Cutmix), the decode head's token un-shuffle, and the negative-class-ranking loss (mode 'unsup_only').
The other in-model augmentations / UniMatch / fdrop switches are outside SURVEY §8 and raise.
"""
import os
from collections import OrderedDict
from numbers import Number

import numpy as np

import torch
import torch.distributed as dist
import torch.nn as nn

from . import kernels as K
from . import runtime
from ._lib import S4FError
from .base_module import BaseModule
from .functional import LOGIT_LD, USE_SIDE_STREAM, ZERO_POOL, join_side_streams, on_head_stream, side_stream
from .params import ParamStore
from .registry import SEGMENTORS, build_backbone, build_head, build_neck

# A/B switch (round 4): the EMA of everything behind the first EMA_SPLIT_LAYER teacher layers on the side stream, under the
# teacher's first layers (update_ema_variables).  OFF by default: measured +0.4 ms per step on the default workload (29.31 ->
# 29.74 ms, three interleaved pairs on one box, profiles/r04_ab_ema_overlap.txt) - the HBM-bound update (1.26 GB) takes the
# bandwidth and the CUs the teacher's first GEMMs want; alone at the head of the step it costs 0.23 ms.
EMA_OVERLAP = os.environ.get('S4F_EMA_OVERLAP', '0') != '0'
EMA_SPLIT_LAYER = 2
# Round 5: the double-buffered teacher.  The EMA of an arena range is computed OUT OF PLACE (s4f_ema_to) into a second teacher
# arena right behind that range's SGD - on the optimiser's stream, under the rest of the backward pass - and becomes the visible
# teacher by a pointer swap exactly where the reference updates it, at the head of the next forward_train
# (encoder_decoder.py:416-423): between the steps state_dict(), evaluation and checkpoints see the teacher the reference
# shows them; the 1.08 GB launch at the head of the chain is gone.  `=0`: the in-place launch.
EMA_DOUBLE = os.environ.get('S4F_EMA_DOUBLE', '1') != '0'


def add_prefix(inputs, prefix):
    """mmseg/core/utils/misc.py:4-17"""
    return {f'{prefix}.{name}': value for name, value in inputs.items()}


def dict_split(dict1, key):
    """models/utils/structual_utils.py:42-53: group every field by dict1[key] (tensors are re-stacked)"""
    groups = {}
    names = list(dict.fromkeys(dict1[key]))
    for name in names:
        flag = [v == name for v in dict1[key]]
        sel = {}
        for k, v in dict1.items():
            if isinstance(v, torch.Tensor):
                idx = [i for i, f in enumerate(flag) if f]
                if idx == list(range(idx[0], idx[-1] + 1)):
                    sel[k] = v[idx[0]:idx[-1] + 1]          # contiguous group: a view, no copy
                else:
                    sel[k] = v[torch.tensor(idx, device=v.device)]
            else:
                sel[k] = [vv for vv, f in zip(v, flag) if f]
        groups[name] = sel
    return groups


def weighted_loss(loss, weight, ignore_keys=()):
    """models/utils/structual_utils.py:132-154 (warmup = 0)"""
    if not isinstance(weight, Number):
        raise NotImplementedError()
    for name in list(loss.keys()):
        if 'loss' in name:
            loss[name] = loss[name] * (0.0 if any(k in name for k in ignore_keys) else weight)
    return loss


class BaseSegmentor(BaseModule):
    """reference mmseg/models/segmentors/base.py (training entry points only)."""

    def __init__(self, init_cfg=None):
        super().__init__(init_cfg)
        self.fp16_enabled = False
        self.log_vars_as_tensors = False    # True: no host sync in _parse_losses (values stay 0-dim tensors)

    @property
    def with_neck(self):
        return hasattr(self, 'neck') and self.neck is not None

    @property
    def with_auxiliary_head(self):
        return hasattr(self, 'auxiliary_head') and self.auxiliary_head is not None

    @property
    def with_decode_head(self):
        return hasattr(self, 'decode_head') and self.decode_head is not None

    def forward(self, img, img_metas, return_loss=True, **kwargs):
        """base.py:108-121"""
        if return_loss:
            return self.forward_train(img, img_metas, **kwargs)
        return self.forward_test(img, img_metas, **kwargs)

    def forward_test(self, imgs, img_metas, **kwargs):
        """base.py:70-106: imgs / img_metas are lists over test-time augmentations"""
        for var, name in [(imgs, 'imgs'), (img_metas, 'img_metas')]:
            if not isinstance(var, list):
                raise TypeError(f'{name} must be a list, but got {type(var)}')
        num_augs = len(imgs)
        if num_augs != len(img_metas):
            raise ValueError(f'num of augmentations ({len(imgs)}) != num of image meta ({len(img_metas)})')
        for img_meta in img_metas:
            ori_shapes = [_['ori_shape'] for _ in img_meta]
            assert all(shape == ori_shapes[0] for shape in ori_shapes)
            img_shapes = [_['img_shape'] for _ in img_meta]
            assert all(shape == img_shapes[0] for shape in img_shapes)
            pad_shapes = [_['pad_shape'] for _ in img_meta]
            assert all(shape == pad_shapes[0] for shape in pad_shapes)
        if num_augs == 1:
            return self.simple_test(imgs[0], img_metas[0], **kwargs)
        return self.aug_test(imgs, img_metas, **kwargs)

    def train_step(self, data_batch, optimizer, **kwargs):
        """base.py:155-206 without the per-iteration debug dumps to the CWD (Q6)."""
        data_batch = dict(data_batch)
        data_batch['iter'] = kwargs['iter']
        losses = self(**data_batch)
        loss, log_vars = self._parse_losses(losses)
        return dict(loss=loss, log_vars=log_vars, num_samples=len(data_batch['img_metas']))

    def _parse_losses(self, losses):
        """base.py:230-274.  The reference's one all-reduce + .item() per scalar becomes one batched all-reduce
        and one device->host copy."""
        log_vars = OrderedDict()
        scalars = all(isinstance(v, torch.Tensor) and v.dim() == 0 and v.is_cuda and v.dtype == torch.float32 for v in losses.values())
        for name, value in losses.items():
            if isinstance(value, torch.Tensor):
                log_vars[name] = value if scalars else value.mean()
            elif isinstance(value, list):
                log_vars[name] = sum(v.mean() for v in value)
            else:
                raise TypeError(f'{name} is not a tensor or list of tensors')
        distributed = dist.is_available() and dist.is_initialized()
        if scalars and any('loss' in k for k in log_vars):
            # the hot path: every entry is a 0-dim device scalar.  ONE stack + ONE sum (two launches forward, views only
            # backward) instead of a mean, an add and a cast launch per entry (~35 launches of ~5 us each on the critical
            # path between the heads' forward and backward passes)
            keys = list(log_vars.keys())
            lk = [k for k in keys if 'loss' in k]
            stacked = torch.stack([log_vars[k] for k in lk])
            loss = stacked.sum()
            rest = [log_vars[k].detach() for k in keys if 'loss' not in k]
            packed = torch.cat([stacked.detach(), loss.detach().reshape(1)] + [r.reshape(1) for r in rest])
            order = lk + ['loss'] + [k for k in keys if 'loss' not in k]
        else:
            loss = sum(v for k, v in log_vars.items() if 'loss' in k)
            order = None
        if distributed and not self.log_vars_as_tensors:   # (the key-count check reads a device scalar: a host sync per step)
            n = torch.tensor(len(log_vars), device=loss.device)
            dist.all_reduce(n)
            assert int(n) == len(log_vars) * dist.get_world_size(), \
                'loss log variables are different across GPUs!\n' + f'rank {dist.get_rank()} keys: ' + ','.join(log_vars)
        log_vars['loss'] = loss
        if order is None:
            order = list(log_vars.keys())
            packed = torch.stack([log_vars[k].detach().reshape(()).to(torch.float32) for k in order])
        if distributed:
            packed = packed / dist.get_world_size()
            dist.all_reduce(packed)
        if self.log_vars_as_tensors:
            vals = list(packed.unbind(0))
        else:
            vals = packed.tolist()
        by_key = dict(zip(order, vals))
        for k in list(log_vars.keys()):
            log_vars[k] = by_key[k]
        return loss, log_vars


def _lockstep_on(var):
    """S4F_AUX_LOCKSTEP / S4F_DECODE_LOCKSTEP = 1 | 0 (default; 'auto' is read as 0 since round 4).  Round 3 turned the lockstep
    on whenever the step exchanges anything (fewer SyncBN exchanges: 32 -> 12 per step; -0.47 ms through a one-rank RCCL group
    then).  Measured again on the round-4 tree with the same tool (tools/exp/rccl_ab3.sh, one box, 28.38 ms without a process
    group): both off 28.72 ms, auxiliary heads in lockstep 29.59, decode head in lockstep 30.70, both 30.70 - the heads lose
    more overlap with each other than the 20 saved exchanges cost at one rank.  What an exchange costs with peers is unmeasured
    (DESIGN section 6): the switches stay, parity-tested at world 2 in both settings."""
    return os.environ.get(var, '0') == '1'


_UNSUP_STREAM = os.environ.get('S4F_UNSUP_STREAM', 'decode')    # experiment: 'decode2' = a third head stream

_UNSUPPORTED_TRUE = ('sup_ema', 'attn_frozen', 'sup_ClassMix', 'sup_cutmix', 'unsup_soft', 'use_CutMix', 'use_CutOut',
                     'use_ClassMix', 'mix_with_labeled', 'patchwise', 'use_PatchShuffle', 'use_PatchShuffle_w_Classmix',
                     'no_pos_embed', 'avg_pos_emd', 'duplicate_pos_emd', 'attn_mask_w_fdrop',
                     'use_fdrop', 'unimatch', 'use_cutmix_adaptive')


@SEGMENTORS.register_module()
class EncoderDecoder(BaseSegmentor):
    def __init__(self, backbone, decode_head, neck=None, auxiliary_head=None, projection_head=None, backbone_ema=None,
                 decode_head_ema=None, neck_ema=None, auxiliary_head_ema=None, projection_head_ema=None,
                 backbone_pretrain=None, pretrained=None, train_cfg=None, test_cfg=None, init_cfg=None,
                 ema=False, sup_ema=False, ema_momentum=0.999, attn_frozen=False, attn_frozen_rate=0.0,
                 momentum_backbone=None, momentum_head=None, momentum_head_dropout=0.0, momentum_head_exp=0.0,
                 momentum_exp=0.0, ema_test=False, sup_ClassMix=False, sup_cutmix=False, unsup_weight=2.0,
                 unsup_confidence=0.75, unsup_soft=False, unsup_temperature=1.0, iter_unsup_start=0, strong_aug_prob=0.5,
                 cutout_area=2, use_CutMix=False, use_CutOut=False, use_ClassMix=False, mix_with_labeled=False,
                 patchwise=False, use_PatchShuffle=False, PatchMix_N=8, patchmix_ratio=0.5, patchsize=16,
                 use_PatchShuffle_w_Classmix=False, use_PatchShuffle_w_Cutmix=False, no_pos_embed=False, avg_pos_emd=False,
                 duplicate_pos_emd=False, adaptive_attn_mask=False, attn_mask_weight=50, attn_mask_seperate_head=False,
                 attn_mask_w_fdrop=False, negative_class_ranking=False, negative_class_ranking_mode='sup_only',
                 use_fdrop=False, unimatch=False, fdrop_loss_weight=0.5, use_cutmix_adaptive=False,
                 plain_mt_pseudo_loss=False):
        super().__init__(init_cfg)
        lcl = locals()
        bad = [k for k in _UNSUPPORTED_TRUE if lcl[k]]
        if bad or neck is not None or projection_head is not None or auxiliary_head_ema is not None \
                or momentum_head_dropout or momentum_head_exp or momentum_exp or unsup_temperature != 1.0:
            raise S4FError(f'options outside the hot-path scope of this build (SURVEY §8, "next" rows): {bad}')
        if pretrained is not None:
            assert backbone.get('pretrained') is None, 'both backbone and segmentor set pretrained weight'
            backbone = dict(backbone, pretrained=pretrained)
        self.backbone = build_backbone(backbone)
        self._init_decode_head(decode_head)
        self._init_auxiliary_head(auxiliary_head)
        self.train_cfg, self.test_cfg = train_cfg, test_cfg
        self.unsup_weight, self.unsup_confidence = unsup_weight, unsup_confidence
        self.iter_unsup_start = iter_unsup_start
        self.ema, self.momentum = ema, ema_momentum
        self.momentum_backbone = momentum_backbone if momentum_backbone is not None else ema_momentum
        self.momentum_head = momentum_head if momentum_head is not None else ema_momentum
        self.patchsize, self.PatchMix_N = patchsize, PatchMix_N
        self.fdrop_loss_weight = fdrop_loss_weight
        self.attn_mask_weight, self.adaptive_attn_mask = attn_mask_weight, adaptive_attn_mask
        self.attn_mask_seperate_head = attn_mask_seperate_head
        # extension (documented in DESIGN.md): the literal MT config never produces an unsupervised loss (Q1);
        # True runs compute_pseudo_loss on the plain mean-teacher branch as encoder_decoder.py:681-685 would.
        self.plain_mt_pseudo_loss = plain_mt_pseudo_loss
        # "ours" (configs/setr/..._MT_w_ours.py:236-256)
        if negative_class_ranking and negative_class_ranking_mode != 'unsup_only':
            raise S4FError(f"negative_class_ranking_mode={negative_class_ranking_mode!r}: only 'unsup_only' (the mode of the SETR "
                           'config) is built; the sup_only / both / kl variants are outside SURVEY §8')
        if (negative_class_ranking or use_PatchShuffle_w_Cutmix) and not attn_mask_seperate_head:
            raise S4FError('use_PatchShuffle_w_Cutmix / negative_class_ranking act on the plain student pass that '
                           'attn_mask_seperate_head=True adds (encoder_decoder.py:633-687)')
        if use_PatchShuffle_w_Cutmix and not isinstance(cutout_area, int):
            raise S4FError('cutout_area must be an int')
        self.use_PatchShuffle_w_Cutmix, self.patchmix_ratio = use_PatchShuffle_w_Cutmix, patchmix_ratio
        self.strong_aug_prob, self.cutout_area = strong_aug_prob, cutout_area
        self.negative_class_ranking = negative_class_ranking
        self.ema_test = ema_test
        self.with_auxiliary_head_ema = False
        if self.ema:
            if self.momentum_backbone != self.momentum_head:
                raise S4FError('one EMA launch covers backbone and head: momentum_backbone must equal momentum_head')
            self._init_ema_model(pretrained, backbone_ema, decode_head_ema)
        assert self.with_decode_head
        self.current_iter = 0
        self.last_mask_ratio = None
        self._student_store = None
        self._teacher_store = None

    # ------------------------------------------------------------------ construction
    def _init_ema_model(self, pretrained, backbone_ema, decode_head_ema):
        if pretrained is not None:
            backbone_ema = dict(backbone_ema, pretrained=pretrained)
        self.backbone_ema = build_backbone(backbone_ema)
        for p_ in self.backbone_ema.parameters():
            p_.detach_()
        self.decode_head_ema = build_head(decode_head_ema)
        for p_ in self.decode_head_ema.parameters():
            p_.detach_()

    def _init_decode_head(self, decode_head):
        self.decode_head = build_head(decode_head)
        self.align_corners = self.decode_head.align_corners
        self.num_classes = self.decode_head.num_classes

    def _init_auxiliary_head(self, auxiliary_head):
        if auxiliary_head is not None:
            if isinstance(auxiliary_head, list):
                self.auxiliary_head = nn.ModuleList([build_head(c) for c in auxiliary_head])
            else:
                self.auxiliary_head = build_head(auxiliary_head)

    # ------------------------------------------------------------------ arenas
    def _aux_list(self):
        if not self.with_auxiliary_head:
            return []
        return list(self.auxiliary_head) if isinstance(self.auxiliary_head, nn.ModuleList) else [self.auxiliary_head]

    def _build_stores(self):
        groups = [('backbone', self.backbone, 'backbone'), ('decode_head', self.decode_head, 'decode_head')]
        aux = self._aux_list()
        for i, a in enumerate(aux):
            pre = f'auxiliary_head.{i}' if isinstance(self.auxiliary_head, nn.ModuleList) else 'auxiliary_head'
            groups.append((f'auxiliary_head.{i}', a, pre))
        self._student_store = ParamStore(groups, with_grad=True)
        self.backbone._attach_store(self._student_store)
        self.decode_head._attach_store(self._student_store)
        for a in aux:
            a._attach_store(self._student_store)
        if self.ema:
            self._teacher_store = ParamStore([('backbone', self.backbone_ema, 'backbone_ema'),
                                              ('decode_head', self.decode_head_ema, 'decode_head_ema')], with_grad=False)
            self.backbone_ema._attach_store(self._teacher_store)
            self.decode_head_ema._attach_store(self._teacher_store)
            n_t = self._teacher_store.total
            s_sig = [s for s in self._student_store.layout_signature() if s[1] < n_t]
            if s_sig != self._teacher_store.layout_signature():
                raise S4FError('teacher and student arenas do not line up (backbone_ema/decode_head_ema must mirror '
                               'backbone/decode_head)')

    def ensure_engine(self, device):
        """(re)build the flat arenas after the model has been moved to `device`; refresh bf16 shadows if stale."""
        if self._student_store is None:
            self._build_stores()
        code = runtime.compute_dtype()
        self._student_store.ensure(device, code)
        self._student_store.ensure_grads()
        self._student_store.sync_shadow()
        if self._teacher_store is not None:
            self._teacher_store.ensure(device, runtime.teacher_dtype())
            self._teacher_store.sync_shadow()
        return self._student_store, self._teacher_store

    @property
    def student_store(self):
        return self._student_store

    @staticmethod
    def sync_gradients():
        """Make the caller's stream wait for every stream this package writes parameter gradients on (the weight-gradient
        side stream, the head streams).  `optimizer.step()` and the gradient reducer do this themselves; anything ELSE that
        reads `.grad` after `loss.backward()` - gradient clipping, a logging hook of a foreign runner - calls this first."""
        join_side_streams()

    @property
    def teacher_store(self):
        return self._teacher_store

    # ------------------------------------------------------------------ feature extraction / heads
    def extract_feat(self, img, no_pos_embed=False, avg_pos_emd=False, duplicate_pos_emd=False, use_fdrop=False,
                     attn_mask=None, attn_mask_weight=5, adaptive_attn_mask=False):
        return self.backbone(img, no_pos_embed=no_pos_embed, avg_pos_emd=avg_pos_emd, duplicate_pos_emd=duplicate_pos_emd,
                             use_fdrop=use_fdrop, attn_mask=attn_mask, attn_mask_weight=attn_mask_weight,
                             adaptive_attn_mask=adaptive_attn_mask)

    def extract_feat_ema(self, img):
        return self.backbone_ema(img)

    # The decode head (both of its calls: BN statistics and parameter gradients of one head stay ordered) and the
    # auxiliary heads run on two extra HIP streams; forward_train joins them before it returns the losses.
    def _decode_head_forward_train(self, x, img_metas, gt_semantic_seg):
        with on_head_stream(gt_semantic_seg.device, 'decode'):
            return add_prefix(self.decode_head.forward_train(x, img_metas, gt_semantic_seg, self.train_cfg), 'decode')

    def _auxiliary_head_forward_train(self, x, img_metas, gt_semantic_seg):
        losses = dict()
        with on_head_stream(gt_semantic_seg.device, 'aux'):
            if isinstance(self.auxiliary_head, nn.ModuleList) and self._aux_lockstep():
                # N > 1: the structurally identical auxiliary heads advance layer by layer TOGETHER, one SyncBN exchange
                # per layer for all of them (16 -> 4 per step); same arithmetic per head
                heads = list(self.auxiliary_head)
                outs = type(heads[0]).forward_train_lockstep(heads, x, img_metas, gt_semantic_seg, self.train_cfg)
                for idx, o in enumerate(outs):
                    losses.update(add_prefix(o, f'aux_{idx}'))
            elif isinstance(self.auxiliary_head, nn.ModuleList):
                for idx, aux_head in enumerate(self.auxiliary_head):
                    losses.update(add_prefix(aux_head.forward_train(x, img_metas, gt_semantic_seg, self.train_cfg), f'aux_{idx}'))
            else:
                losses.update(add_prefix(self.auxiliary_head.forward_train(x, img_metas, gt_semantic_seg, self.train_cfg), 'aux'))
        return losses

    def _aux_lockstep(self):
        """The structurally identical auxiliary heads advance layer by layer together, one SyncBN exchange per layer for all four
        (16 -> 4 per step).  S4F_AUX_LOCKSTEP=1; off by default (see _lockstep_on): one head after the other, the path every
        single-GPU parity test runs."""
        if not _lockstep_on('S4F_AUX_LOCKSTEP'):
            return False
        heads = list(self.auxiliary_head)
        return len(heads) > 1 and all(hasattr(type(h), 'forward_train_lockstep') for h in heads) and \
            len({(h.num_convs, h.channels, h.up_scale) for h in heads}) == 1

    def _decode_lockstep(self):
        """S4F_DECODE_LOCKSTEP=1: the decode head's calls of a step advance in lockstep too (8 fewer SyncBN exchanges); off by
        default (see _lockstep_on)."""
        return _lockstep_on('S4F_DECODE_LOCKSTEP') and hasattr(type(self.decode_head), 'fused_losses_lockstep')

    # ------------------------------------------------------------------ evaluation
    # Reference: encoder_decoder.py:265-333, 1118-1231 with the intended semantics of its `whole` mode (as written,
    # whole_inference calls encode_decode without the positional `adaptive_attn_mask` it requires - Q8 - and raises):
    # backbone -> decode head (eval-mode BN) -> logits at the input size -> crop the padding, rescale to ori_shape ->
    # softmax -> flip back -> argmax.  All tensor work is in HIP kernels (head logits, kernels.resize_bilinear,
    # kernels.softmax_argmax); nothing here runs under autograd.
    def _decode_head_forward_test(self, x, img_metas):
        return self.decode_head.forward_test(x, img_metas, self.test_cfg)

    def _decode_head_forward_test_ema(self, x, img_metas):
        return self.decode_head_ema.forward_test(x, img_metas, self.test_cfg)

    def encode_decode(self, img, img_metas):
        """encoder_decoder.py:265-295: logits [B, C, H, W] of the student at the size of `img`"""
        with torch.no_grad():
            self.ensure_engine(img.device)
            out = self._decode_head_forward_test(self.extract_feat(img), img_metas)
            if tuple(out.shape[2:]) != tuple(img.shape[2:]):
                out = K.resize_bilinear(out, img.shape[2:], self.align_corners)
        return out

    def encode_decode_ema(self, img, img_metas):
        """encoder_decoder.py:297-307: the same through the EMA teacher (ema_test=True)"""
        with torch.no_grad():
            self.ensure_engine(img.device)
            out = self._decode_head_forward_test_ema(self.extract_feat_ema(img), img_metas)
            if tuple(out.shape[2:]) != tuple(img.shape[2:]):
                out = K.resize_bilinear(out, img.shape[2:], self.align_corners)
        return out

    def whole_inference(self, img, img_meta, rescale):
        """encoder_decoder.py:1118-1147"""
        seg_logit = self.encode_decode_ema(img, img_meta) if self.ema_test else self.encode_decode(img, img_meta)
        if rescale:
            resize_shape = img_meta[0]['img_shape'][:2]          # remove padding area: read only this window
            size = img_meta[0]['ori_shape'][:2]
            seg_logit = K.resize_bilinear(seg_logit, size, self.align_corners, window=resize_shape)
        return seg_logit

    def slide_inference(self, img, img_meta, rescale):
        """encoder_decoder.py:1068-1116: overlapping windows of test_cfg.crop_size at test_cfg.stride (the last window of a
        row / column is shifted back inside the image), the windows' logits summed where they land and divided by the
        cover count.  Every window goes through the HIP forward; the sum / count / crop bookkeeping is the reference's
        torch arithmetic (adding the zero-padded window logits = adding them inside the window)."""
        tc = self.test_cfg
        get = tc.get if isinstance(tc, dict) else (lambda k: getattr(tc, k))
        h_stride, w_stride = get('stride')
        h_crop, w_crop = get('crop_size')
        batch_size, _, h_img, w_img = img.size()
        h_grids = max(h_img - h_crop + h_stride - 1, 0) // h_stride + 1
        w_grids = max(w_img - w_crop + w_stride - 1, 0) // w_stride + 1
        preds = img.new_zeros((batch_size, self.num_classes, h_img, w_img))
        count_mat = img.new_zeros((batch_size, 1, h_img, w_img))
        for h_idx in range(h_grids):
            for w_idx in range(w_grids):
                y1, x1 = h_idx * h_stride, w_idx * w_stride
                y2, x2 = min(y1 + h_crop, h_img), min(x1 + w_crop, w_img)
                y1, x1 = max(y2 - h_crop, 0), max(x2 - w_crop, 0)
                crop_img = img[:, :, y1:y2, x1:x2].contiguous()
                logit = self.encode_decode_ema(crop_img, img_meta) if self.ema_test else self.encode_decode(crop_img, img_meta)
                preds[:, :, y1:y2, x1:x2] += logit
                count_mat[:, :, y1:y2, x1:x2] += 1
        assert (count_mat == 0).sum() == 0
        preds = preds / count_mat
        if rescale:
            resize_shape = img_meta[0]['img_shape'][:2]          # remove padding area: read only this window
            preds = K.resize_bilinear(preds.contiguous(), img_meta[0]['ori_shape'][:2], self.align_corners, window=resize_shape)
        return preds

    def inference(self, img, img_meta, rescale, return_labels=False):
        """encoder_decoder.py:1174-1203 -> softmax probabilities [B, C, H, W], flipped back when the test image was flipped
        (return_labels=True: also the arg-max labels [B, H, W] uint8 of the same pass)"""
        mode = (self.test_cfg or {}).get('mode', 'whole') if isinstance(self.test_cfg, dict) else self.test_cfg.mode
        assert mode in ['slide', 'whole']
        ori_shape = img_meta[0]['ori_shape']
        assert all(_['ori_shape'] == ori_shape for _ in img_meta)
        seg_logit = self.slide_inference(img, img_meta, rescale) if mode == 'slide' else self.whole_inference(img, img_meta, rescale)
        flip = 0
        if img_meta[0].get('flip', False):
            flip_direction = img_meta[0]['flip_direction']
            assert flip_direction in ['horizontal', 'vertical']
            flip = 1 if flip_direction == 'horizontal' else 2
        prob, label, _ = K.softmax_argmax(seg_logit.contiguous(), want_prob=True, flip=flip)
        return (prob, label) if return_labels else prob

    def simple_test(self, img, img_meta, rescale=True):
        """encoder_decoder.py:1205-1223 -> list of [H, W] integer label maps (numpy), one per image"""
        _, label = self.inference(img, img_meta, rescale, return_labels=True)
        return list(label.cpu().numpy().astype(np.int64))

    def aug_test(self, imgs, img_metas, rescale=True):
        """encoder_decoder.py:1249-1267: mean of the augmentations' probabilities, then arg-max"""
        assert rescale
        seg_logit = self.inference(imgs[0], img_metas[0], rescale)
        for i in range(1, len(imgs)):
            seg_logit += self.inference(imgs[i], img_metas[i], rescale)
        seg_logit /= len(imgs)
        _, label, _ = K.softmax_argmax(seg_logit.contiguous(), want_prob=False, flip=0, raw=True)    # arg-max of the mean probabilities
        return list(label.cpu().numpy().astype(np.int64))

    # ------------------------------------------------------------------ EMA
    def update_ema_variables(self, model=None, ema_model=None, momentum=None, dropout=0.0, attn_frozen=False):
        """encoder_decoder.py:1044-1066 for (backbone, decode_head incl. BN running stats) in ONE launch over the
        arenas; the per-module signature of the reference is accepted and ignored."""
        s, t = self._student_store, self._teacher_store
        m = self.momentum_backbone if momentum is None else momentum
        pend, self._ema_pending = self.__dict__.get('_ema_pending'), None
        if pend is not None and momentum is None and pend == self._ema_pending_key(m):
            t.swap()                                  # the update was computed behind the previous step's SGD: make it visible
            return
        cut = self._ema_split()
        if not cut:
            K.ema(t.flat, s.flat, t.flat_t, t.total, m, t.dtype)
            return
        # Round 4 (S4F_EMA_OVERLAP=1): only the head of the arena (embeddings + the first EMA_SPLIT_LAYER encoder layers, 17 % of
        # DeiT-B) is updated in front of the teacher pass; the rest goes to the side stream and the teacher backbone waits for it
        # in front of layer EMA_SPLIT_LAYER (vit.py forward_rank1), ~0.8 ms of teacher kernels later.
        # Same arithmetic, same place in the iteration as the reference (encoder_decoder.py:416-423); forward_train joins the
        # side stream before it returns, so a reader on the caller's stream never sees a half-updated teacher.
        K.ema(t.flat[:cut], s.flat[:cut], None if t.flat_t is None else t.flat_t[:cut], cut, m, t.dtype)
        cur, side = torch.cuda.current_stream(), side_stream(t.flat.device)
        side.wait_stream(cur)
        with torch.cuda.stream(side):
            K.ema(t.flat[cut:], s.flat[cut:], None if t.flat_t is None else t.flat_t[cut:], t.total - cut, m, t.dtype)
            ev = torch.cuda.Event()
            ev.record(side)
        self.backbone_ema._pre_layer_wait = (EMA_SPLIT_LAYER, ev)

    # ---- the double-buffered teacher (EMA_DOUBLE): hooks called by S4FSGD on the stream of the update
    def _ema_pending_key(self, m):
        """what the pending update was computed from: any foreign write to the student or the teacher in between (load_state_dict,
        a torch optimiser, an in-place edit - of a parameter OR of a BatchNorm running statistic; raw-pointer writers call
        mark_dirty()) moves a version counter and the update is dropped for the in-place launch"""
        s, t = self._student_store, self._teacher_store
        return (float(m), s.generation, t.generation, s._state_version_sum(), t._state_version_sum())

    def _ema_double_on(self):
        t = self._teacher_store
        return (EMA_DOUBLE and self.ema and t is not None and t.flat is not None and t.flat.is_cuda and not EMA_OVERLAP and
                os.environ.get('S4F_TEACHER_GRAPH', '0') != '1')

    @torch.no_grad()
    def _ema_behind_update(self, lo, hi):
        """student arena range [lo, hi) has just been stepped (current stream): the teacher's next value of it"""
        if not self._ema_double_on():
            return
        s, t = self._student_store, self._teacher_store
        hi = min(hi, t.total)
        if lo >= hi:
            return
        t.enable_double()
        if self.__dict__.get('_ema_cov_epoch') != getattr(s, 'step_epoch', 0):
            self._ema_cov, self._ema_cov_epoch = [], getattr(s, 'step_epoch', 0)
        f2, ft2 = t.other()
        K.ema_to(t.flat[lo:hi], s.flat[lo:hi], f2[lo:hi], None if ft2 is None else ft2[lo:hi], hi - lo, self.momentum_backbone, t.dtype)
        self._ema_cov.append((lo, hi))

    @torch.no_grad()
    def _ema_finish(self):
        """end of optimizer.step() (its stream has joined the eager streams): the rest of the teacher arena - BatchNorm running
        statistics, ranges stepped here - and the record that makes the next forward_train swap instead of launch"""
        if not self._ema_double_on():
            return
        s, t = self._student_store, self._teacher_store
        t.enable_double()
        cov = sorted(self._ema_cov) if self.__dict__.get('_ema_cov_epoch') == getattr(s, 'step_epoch', 0) else []
        self._ema_cov, self._ema_cov_epoch = [], None
        f2, ft2 = t.other()
        pos = 0
        for a, b in cov + [(t.total, t.total)]:
            if a > pos:
                K.ema_to(t.flat[pos:a], s.flat[pos:a], f2[pos:a], None if ft2 is None else ft2[pos:a], a - pos, self.momentum_backbone, t.dtype)
            pos = max(pos, b)
        self._ema_pending = self._ema_pending_key(self.momentum_backbone)

    def _ema_split(self):
        """arena offset where the side-stream part of the EMA starts (0 = one launch on the caller's stream)"""
        t = self._teacher_store
        key = (getattr(t, 'generation', 0), id(t.flat))
        c = self.__dict__.get('_ema_cut')
        if c is not None and c[0] == key:
            return c[1]
        cut = 0
        if EMA_OVERLAP and USE_SIDE_STREAM and os.environ.get('S4F_TEACHER_GRAPH', '0') != '1' and t.flat.is_cuda:
            import re
            first = None
            for e in t.entries:                              # arena order
                mm = re.search(r'^backbone_ema\.layers\.(\d+)\.', e.name)
                late = e.group != 'backbone' or (mm is not None and int(mm.group(1)) >= EMA_SPLIT_LAYER)
                if late and first is None:
                    first = e.off
                if not late and first is not None:           # something the first layers need lies behind the cut: no split
                    first = 0
                    break
            cut = first or 0
            if len(self.backbone_ema.layers) <= EMA_SPLIT_LAYER:
                cut = 0
        self._ema_cut = (key, cut)
        return cut

    def set_eval(self, ema=False):
        mods = [self.backbone_ema, self.decode_head_ema] if ema else [self.backbone, self.decode_head] + self._aux_list()
        for m in mods:
            m.eval()

    def set_train(self, ema=False):
        mods = [self.backbone_ema, self.decode_head_ema] if ema else [self.backbone, self.decode_head] + self._aux_list()
        for m in mods:
            m.train()

    # ------------------------------------------------------------------ training
    def forward_train(self, img, img_metas, **kwargs):
        """encoder_decoder.py:386-514"""
        losses = self._forward_train(img, img_metas, **kwargs)
        join_side_streams()                       # head streams -> caller's stream; the loss scalars are read there
        cur = torch.cuda.current_stream()
        for v in list(losses.values()) + [getattr(self, 'last_mask_ratio', None)]:
            if isinstance(v, torch.Tensor) and v.is_cuda:
                v.record_stream(cur)
        return losses

    def _forward_train(self, img, img_metas, **kwargs):
        if not img.is_cuda:
            raise S4FError('the S4Former step runs on the MI355X HIP kernels only: move model and batch to the GPU')
        self.ensure_engine(img.device)
        ZERO_POOL.begin(img.device)
        self._student_store.step_epoch = getattr(self._student_store, 'step_epoch', 0) + 1
        current_iter = kwargs.pop('iter')
        self.current_iter = current_iter
        kwargs.update({'img': img, 'img_metas': img_metas, 'tag': [meta['tag'] for meta in img_metas]})
        data_groups = dict_split(kwargs, 'tag')
        for _, v in data_groups.items():
            v.pop('tag')

        self.losses = dict()
        if self.ema:
            with torch.no_grad():
                self.update_ema_variables()

        sup_imgs = sup_gts = None
        do_unsup = ('unsup_student' in data_groups) and self.unsup_weight != 0
        fused = 'sup' in data_groups and do_unsup and self.ema and (self.attn_mask_seperate_head or self.plain_mt_pseudo_loss)
        if fused:
            # One backbone pass for the supervised and the unlabeled-student images (every backbone op is per image;
            # the heads, whose BatchNorm statistics are per call (Q10), still run per group).  Same arithmetic as the
            # reference's separate extract_feat calls, twice the GEMM rows per launch.
            self._fused_step(data_groups)
            return self.losses
        if 'sup' in data_groups:
            sup_imgs = data_groups['sup']['img']
            sup_gts = data_groups['sup']['gt_semantic_seg']
            labeled_features = self.extract_feat(sup_imgs)
            loss_decode_sup = self._decode_head_forward_train(labeled_features, data_groups['sup']['img_metas'], sup_gts)
            if self.with_auxiliary_head:
                self.losses.update(self._auxiliary_head_forward_train(labeled_features, data_groups['sup']['img_metas'], sup_gts))
            self.losses.update(loss_decode_sup)

        if do_unsup:
            unsup_raw = self.foward_unsup_train(data_groups['unsup_teacher'], data_groups['unsup_student'], sup_imgs, sup_gts)
            with on_head_stream(img.device, 'decode'):
                unsup_loss = weighted_loss(unsup_raw, weight=self.unsup_weight)
            if self.iter_unsup_start != 0:
                if self.current_iter > self.iter_unsup_start:
                    self.losses.update(unsup_loss)
            else:
                self.losses.update(unsup_loss)
        return self.losses

    def _teacher_pass(self, teacher_data, student_data):
        tnames = [meta['filename'] for meta in teacher_data['img_metas']]
        snames = [meta['filename'] for meta in student_data['img_metas']]
        tidx = [tnames.index(name) for name in snames]
        timg = teacher_data['img']
        if tidx != list(range(len(tidx))):
            timg = timg[torch.tensor(tidx, device=timg.device)]
        if not self.ema:
            raise S4FError('the teacher of this build is the EMA model (ema=True in all three SETR configs)')
        metas = [teacher_data['img_metas'][i] for i in tidx]
        with torch.no_grad():
            # The reference switches the EMA modules to eval() around this call (encoder_decoder.py:520-524, 586).  The only
            # mode-dependent op of the teacher is the head's BatchNorm, so the same effect is had by an override flag
            # on that head instead of two walks over ~400 modules per step; module.training is what it was before.
            self.decode_head_ema._eval_override = True
            try:
                teacher_info = self._teacher_graphed(timg, metas)
                if teacher_info is None:
                    teacher_info = self.extract_teacher_info_ema(timg, metas)
            finally:
                self.decode_head_ema._eval_override = False
        return teacher_info

    # The teacher pass is a fixed sequence of ~110 launches on fixed-shape inputs with no autograd and no host decision in
    # it: after two eager steps (GEMM variants resolved, kernel attributes set) it is captured ONCE in a hipGraph and
    # replayed - one launch per step instead of ~110 ctypes calls, and no launch gaps between its kernels.  The EMA
    # weights it reads live at fixed arena addresses, so the graph follows every update; the input is copied into the
    # graph's static buffer, the outputs live in the graph's memory pool until the next replay (they are consumed inside
    # the step).  OPT-IN (S4F_TEACHER_GRAPH=1): measured on the default workload the replayed graph costs 1.0 ms per step MORE
    # than the eager launches (33.2 vs 32.2 ms, same box, same run) although it saves ~2 ms of host time - the graph's kernel
    # nodes do not run back to back the way stream-ordered launches do on this runtime - so it only pays on a host-bound
    # box.  A failed capture falls back to the eager path with one warning.
    def _teacher_graphed(self, timg, metas):
        if os.environ.get('S4F_TEACHER_GRAPH', '0') != '1' or not timg.is_cuda:
            return None
        st = self.__dict__.setdefault('_tgraph', dict(seen={}, graphs={}, off=False))
        if st['off']:
            return None
        key = (tuple(timg.shape), runtime.compute_dtype(), getattr(self._teacher_store, 'generation', 0),
               float(self.unsup_confidence), timg.device.index)
        if key not in st['graphs']:
            st['seen'][key] = st['seen'].get(key, 0) + 1
            if st['seen'][key] <= 2:
                return None                                   # eager warm-up steps
            try:
                static_in = torch.empty_like(timg)
                static_in.copy_(timg)
                torch.cuda.synchronize()
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g):
                    info = self.extract_teacher_info_ema(static_in, metas)
                st['graphs'] = {key: (g, static_in, info)}    # one shape at a time: a new shape drops the old graph's pool
            except Exception as e:                            # noqa: BLE001 - any capture failure means: stay eager
                import warnings
                warnings.warn(f'teacher hipGraph capture failed ({type(e).__name__}: {e}); the teacher pass stays eager')
                st['off'] = True
                torch.cuda.synchronize()
                return None
        g, static_in, info = st['graphs'][key]
        static_in.copy_(timg)
        g.replay()
        return dict(info, img_metas=metas)

    def _fused_step(self, data_groups):
        sup, stu = data_groups['sup'], data_groups['unsup_student']
        teacher_info = self._teacher_pass(data_groups['unsup_teacher'], stu)
        sup_imgs, simg = sup['img'], stu['img']
        ns, nu = sup_imgs.shape[0], simg.shape[0]
        conf = teacher_info['conf_mask']
        direct = (not self.adaptive_attn_mask and conf.dtype == torch.uint8 and conf.is_contiguous() and self.backbone.with_cls_token
                  and conf.shape[1] % self.patchsize == 0 and conf.shape[2] % self.patchsize == 0)
        if not direct:
            u = self._conf_to_patch_u(conf)
            bu, flag, w = self.backbone._rank1_mask(u, self.attn_mask_weight, self.adaptive_attn_mask)
        plain_img, aug = simg, None
        if self.attn_mask_seperate_head and self.use_PatchShuffle_w_Cutmix:
            plain_img, aug = self._strong_augment(simg, teacher_info)       # images of the plain pass, labels, token maps
        groups = [sup_imgs, simg] + ([plain_img] if self.attn_mask_seperate_head else [])
        nb = sum(g.shape[0] for g in groups)
        # the batch usually is [sup..., unsup_student..., unsup_teacher...]: sup + student is then one contiguous view
        if not self.attn_mask_seperate_head and sup_imgs.data_ptr() + sup_imgs.numel() * 4 == simg.data_ptr() and \
                sup_imgs._base is not None and sup_imgs._base is simg._base:
            base = sup_imgs._base
            off = (sup_imgs.data_ptr() - base.data_ptr()) // (4 * sup_imgs[0].numel())
            imgs = base[off:off + nb]
        else:
            imgs = torch.cat(groups, 0)
        row_flag = None
        if direct:
            # one launch writes the whole bias matrix (zero rows for the images without a mask): s4f_pasa_patch_u
            ps = self.patchsize
            bias_u = torch.empty(nb, (conf.shape[1] // ps) * (conf.shape[2] // ps) + 1, device=imgs.device)
            K.pasa_patch_u(conf, bias_u, ps, ns)
            w = float(self.attn_mask_weight)
        else:
            N = bu.shape[1]
            bias_u = torch.zeros(nb, N, device=imgs.device)
            bias_u[ns:ns + nu] = bu
            if flag is not None:
                row_flag = torch.ones(nb, N, device=imgs.device)
                row_flag[ns:ns + nu] = flag
        bounds = tuple([(0, ns), (ns, ns + nu)] + ([(ns + nu, ns + 2 * nu)] if self.attn_mask_seperate_head else []))
        outs = self.backbone.forward_rank1(imgs, (bias_u, row_flag, w), tap_groups=bounds)
        parts = self.backbone.split_taps_multi(outs, bounds)
        f_sup, f_mask = parts[0], parts[1]
        f_plain = parts[2] if self.attn_mask_seperate_head else None
        if self._decode_lockstep() and aug is None and not self.negative_class_ranking:
            # N > 1: the decode head's calls (labelled, masked pseudo-labelled, plain pseudo-labelled) advance layer by layer
            # TOGETHER: one SyncBN exchange per layer for all of them; BN running statistics are updated in call order
            dh = self.decode_head
            pseudo = teacher_info['hard_seg_label']
            calls = [(dh, f_sup, dh._loss_labels(sup['img_metas'], sup['gt_semantic_seg']), dh.loss_decode.loss_weight),
                     (dh, f_mask, pseudo, 1.0)]
            if self.attn_mask_seperate_head:
                calls.append((dh, f_plain, pseudo, 1.0))
            with on_head_stream(simg.device, 'decode'):
                dl = type(dh).fused_losses_lockstep(calls)
                loss_decode_sup = add_prefix({dh.loss_decode.loss_name: dl[0]}, 'decode')
                loss_unsup = {}
                if self.attn_mask_seperate_head:
                    loss_unsup['loss_seg_unsup_attn_mask'] = dl[1] * 0.5
                loss_unsup['loss_seg_unsup'] = dl[-1] * self.fdrop_loss_weight
                if self.unsup_confidence != 0:
                    self.last_mask_ratio = teacher_info['conf_count'].to(torch.float32) / pseudo.numel()
                unsup_loss = weighted_loss(loss_unsup, weight=self.unsup_weight)
            if self.with_auxiliary_head:
                self.losses.update(self._auxiliary_head_forward_train(f_sup, sup['img_metas'], sup['gt_semantic_seg']))
            self.losses.update(loss_decode_sup)
        else:
            self._fused_heads_sequential(sup, stu, simg, f_plain, f_sup, f_mask, ns, nu, teacher_info, aug)
            return
        if self.iter_unsup_start != 0:
            if self.current_iter > self.iter_unsup_start:
                self.losses.update(unsup_loss)
        else:
            self.losses.update(unsup_loss)

    def _fused_heads_sequential(self, sup, stu, simg, f_plain, f_sup, f_mask, ns, nu, teacher_info, aug=None):
        # supervised heads
        loss_decode_sup = self._decode_head_forward_train(f_sup, sup['img_metas'], sup['gt_semantic_seg'])
        if self.with_auxiliary_head:
            self.losses.update(self._auxiliary_head_forward_train(f_sup, sup['img_metas'], sup['gt_semantic_seg']))
        self.losses.update(loss_decode_sup)
        # unsupervised heads (same order of head calls as the reference: sup, masked-unsup, plain-unsup); the loss
        # scalings stay on the decode head's stream with the losses they scale
        # (a stream of its own for this second call of the decode head, S4F_UNSUP_STREAM=decode2, was measured: +0.4 ms with
        # equal stream priorities - and 42 ms instead of 34 under a high-priority chain, see bench.py; it shares the
        # labelled call's stream)
        with on_head_stream(simg.device, _UNSUP_STREAM):
            loss_unsup = {}
            student_info = dict(img=simg, img_metas=stu['img_metas'], backbone_feature=f_mask)
            if self.attn_mask_seperate_head:
                loss_unsup['loss_seg_unsup_attn_mask'] = self.compute_pseudo_loss(student_info, teacher_info)['loss_seg_unsup'] * 0.5
                student_info['backbone_feature'] = f_plain
            losses = self.compute_pseudo_loss(student_info, teacher_info, aug=aug, ncr=self.negative_class_ranking)
            if self.negative_class_ranking:
                loss_unsup['loss_ncr_unsup'] = losses['loss_ncr_unsup'] * 0.5
            loss_unsup['loss_seg_unsup'] = losses['loss_seg_unsup'] * self.fdrop_loss_weight
            unsup_loss = weighted_loss(loss_unsup, weight=self.unsup_weight)
        if self.iter_unsup_start != 0:
            if self.current_iter > self.iter_unsup_start:
                self.losses.update(unsup_loss)
        else:
            self.losses.update(unsup_loss)

    def _conf_to_patch_u(self, conf_mask):
        """encoder_decoder.py:547-555: per-patch mean of (1 - conf) -> [B, gh, gw] (Q12: square crops)"""
        ps = self.patchsize
        Bn, H, W = conf_mask.shape
        c = conf_mask.view(Bn, H // ps, ps, W // ps, ps).to(torch.float32)
        return (1.0 - c).sum(dim=(2, 4)) / (ps * ps)

    def foward_unsup_train(self, teacher_data, student_data, sup_imgs, sup_gts):
        """encoder_decoder.py:516-687 (mean-teacher branch; the in-model strong augmentations are 'next' rows)"""
        loss_unsup = {}
        teacher_info = self._teacher_pass(teacher_data, student_data)
        # (hard_seg_label already carries 255 where conf_mask == 0: encoder_decoder.py:541-542 is fused in K15)
        student_info = dict(img=student_data['img'], img_metas=student_data['img_metas'])

        if self.attn_mask_seperate_head:
            attn_mask = self._conf_to_patch_u(teacher_info['conf_mask'])
            feat = self.extract_feat(student_info['img'], attn_mask=attn_mask, attn_mask_weight=self.attn_mask_weight,
                                     adaptive_attn_mask=self.adaptive_attn_mask)
            student_info['backbone_feature'] = feat
            with on_head_stream(feat[-1].device, 'decode'):      # the scaling stays on the stream of the loss it scales
                loss_unsup['loss_seg_unsup_attn_mask'] = self.compute_pseudo_loss(student_info, teacher_info)['loss_seg_unsup'] * 0.5
            aug = None
            if self.use_PatchShuffle_w_Cutmix:
                student_info['img'], aug = self._strong_augment(student_info['img'], teacher_info)
            feat = self.extract_feat(student_info['img'])
            student_info['backbone_feature'] = feat
            with on_head_stream(student_info['img'].device, 'decode'):
                losses = self.compute_pseudo_loss(student_info, teacher_info, aug=aug, ncr=self.negative_class_ranking)
                if self.negative_class_ranking:
                    loss_unsup['loss_ncr_unsup'] = losses['loss_ncr_unsup'] * 0.5
                loss_unsup['loss_seg_unsup'] = losses['loss_seg_unsup'] * self.fdrop_loss_weight
            return loss_unsup
        elif self.plain_mt_pseudo_loss:
            attn_mask = self._conf_to_patch_u(teacher_info['conf_mask'])
            feat = self.extract_feat(student_info['img'], attn_mask=attn_mask, attn_mask_weight=self.attn_mask_weight,
                                     adaptive_attn_mask=self.adaptive_attn_mask)
            student_info['backbone_feature'] = feat
        else:
            # literal MT config (Q1): the reference runs this student forward and throws the result away; it has no
            # side effect (backbone only, no BN), so it is skipped here.
            return loss_unsup

        with on_head_stream(student_info['img'].device, 'decode'):
            losses = self.compute_pseudo_loss(student_info, teacher_info)
            loss_unsup['loss_seg_unsup'] = losses['loss_seg_unsup'] * self.fdrop_loss_weight
        return loss_unsup

    def extract_teacher_info_ema(self, img, img_metas, unsup_confidence=None):
        """encoder_decoder.py:875-904: teacher forward (eval-mode BN), softmax / max / threshold fused into one
        kernel: hard_seg_label uint8 (255 where not confident), conf_mask uint8 {0,1}."""
        th = self.unsup_confidence if unsup_confidence is None else unsup_confidence
        feat = self.extract_feat_ema(img)
        logits_lo, (Bn, h, w) = self.decode_head_ema.logits_lowres(feat)
        s = self.decode_head_ema.up_scale
        label = torch.empty(Bn, h * s, w * s, device=img.device, dtype=torch.uint8)
        conf = torch.empty(Bn, h * s, w * s, device=img.device, dtype=torch.uint8)
        cnt = torch.zeros(1, device=img.device, dtype=torch.int64)
        K.up_pseudo_label(logits_lo, label, conf, cnt, th, Bn, h, w, self.num_classes, LOGIT_LD, s)
        return dict(backbone_feature=feat, seg_logits_lowres=logits_lo, hard_seg_label=label, conf_mask=conf,
                    conf_count=cnt, img_metas=img_metas)

    def _strong_augment(self, simg, teacher_info):
        """use_PatchShuffle_w_Cutmix (encoder_decoder.py:633-638): CutMix between the student images (and between their
        pseudo-labels; conf_mask and the teacher logits stay as they are), then PatchShuffle of the images in blocks of
        patchsize * PatchMix_N pixels.  The decisions come from the host generators in the reference's order
        (augment.draw_strong_aug); the pixels move in one gather kernel.  Returns the augmented images and
        dict(labels, maps): cut-mixed labels and the token un-shuffle maps the decode head applies (decode_head.py:186-212)."""
        from . import augment as A
        Bn, _, H, W = simg.shape
        block = self.patchsize * self.PatchMix_N
        if H != W or H % block:
            raise S4FError(f'PatchShuffle needs square images of whole {block}-pixel blocks, got {H}x{W}')
        boxes, perms = A.draw_strong_aug(Bn, H, W, self.strong_aug_prob, self.cutout_area, self.patchmix_ratio, block)
        dev = simg.device
        box_d = torch.from_numpy(boxes).to(dev, non_blocking=True)
        perm_d = torch.from_numpy(perms).to(dev, non_blocking=True)
        simg = simg.contiguous()
        out = torch.empty_like(simg)
        K.mix_images(simg, out, box_d.reshape(-1), perm_d.reshape(-1), block)
        labels = teacher_info['hard_seg_label']
        if boxes.any():
            mixed = torch.empty_like(labels)
            K.cutmix_labels(labels, mixed, box_d.reshape(-1))
            labels = mixed
        maps = None
        if (perms != np.arange(perms.shape[1], dtype=perms.dtype)[None]).any():
            fwd, bwd = A.token_unshuffle_maps(perms, H // self.patchsize, self.PatchMix_N)
            maps = (torch.from_numpy(fwd).to(dev, non_blocking=True), torch.from_numpy(bwd).to(dev, non_blocking=True))
        self.last_aug = dict(boxes=boxes, perms=perms)
        return out, dict(labels=labels, maps=maps)

    def compute_pseudo_loss(self, student_info, teacher_info, aug=None, ncr=False):
        """encoder_decoder.py:906-954 (hard labels): mean over ALL pixels of CE(student logits, pseudo labels with
        ignore 255); mask_ratio = sum(conf) / numel (kept on the device in self.last_mask_ratio).  aug: the cut-mixed labels
        and the token un-shuffle of a strongly augmented student batch; ncr: add 'loss_ncr_unsup' (mode 'unsup_only')
        against the teacher's logits (the reference also evaluates it on the masked pass and drops the value: skipped)."""
        with on_head_stream(teacher_info['hard_seg_label'].device, 'decode'):
            labels = teacher_info['hard_seg_label'] if aug is None else aug['labels']
            res = self.decode_head.fused_loss(student_info['backbone_feature'], labels, 1.0,
                                              ncr_teacher_lo=teacher_info['seg_logits_lowres'] if ncr else None,
                                              token_maps=None if aug is None else aug['maps'])
            out = {'loss_seg_unsup': res[0], 'loss_ncr_unsup': res[1]} if ncr else {'loss_seg_unsup': res}
            if self.unsup_confidence != 0:
                numel = teacher_info['hard_seg_label'].numel()
                self.last_mask_ratio = teacher_info['conf_count'].to(torch.float32) / numel
                out['mask_ratio'] = self.last_mask_ratio
        return out
