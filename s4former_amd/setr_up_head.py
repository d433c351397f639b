"""SETRUPHead / BaseDecodeHead with the reference's constructor kwargs, call protocol and state-dict keys
(reference mmseg/models/decode_heads/setr_up_head.py:28-111, decode_head.py:54-355), running on the C-ABI HIP
kernels (functional.head_forward / HeadLossFn).

forward_train (the training hot path) goes through the fused node: LN -> [conv3x3 -> (Sync)BN -> ReLU -> up]*n ->
conv_seg -> fused (last upsample + CE).  forward()/forward_test()/forward_get_logits() return the full-size
[B, C, H, W] fp32 logits the mmseg API promises (inference / teacher use; no autograd through that path).
"""
import torch
import torch.nn as nn

from . import kernels as K
from . import runtime
from ._lib import S4FError
from .base_module import BaseModule, constant_init, kaiming_init
from .functional import LOGIT_LD, HeadLossFn, head_forward
from .losses import CrossEntropyLoss
from .params import ParamStore
from .registry import HEADS, build_loss


class ConvModule(nn.Module):
    """mmcv ConvModule container: conv (no bias when a norm follows) -> bn -> ReLU; names `conv`, `bn`, `activate`."""

    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, norm_cfg=None, act_cfg=dict(type='ReLU')):
        super().__init__()
        assert norm_cfg is not None and norm_cfg['type'] in ('BN', 'SyncBN'), 'the PUP head uses (Sync)BN'
        assert act_cfg is not None and act_cfg['type'] == 'ReLU'
        self.conv = nn.Conv2d(in_channels, out_channels, kernel_size, stride, padding, bias=False)
        self.bn = nn.BatchNorm2d(out_channels, eps=norm_cfg.get('eps', 1e-5), momentum=norm_cfg.get('momentum', 0.1))
        for p_ in self.bn.parameters():
            p_.requires_grad = norm_cfg.get('requires_grad', True)
        self.activate = nn.ReLU(inplace=True)
        self.norm_name = 'bn'
        self.sync = norm_cfg['type'] == 'SyncBN'
        self.init_weights()

    def init_weights(self):
        kaiming_init(self.conv, a=0, nonlinearity='relu')
        constant_init(self.bn, 1, bias=0)


class Upsample(nn.Module):
    """mmseg.ops.Upsample container (ops/wrappers.py:31-51); the resampling itself is fused into the kernels."""

    def __init__(self, size=None, scale_factor=None, mode='nearest', align_corners=None):
        super().__init__()
        self.size, self.mode, self.align_corners = size, mode, align_corners
        self.scale_factor = float(scale_factor) if scale_factor else None


class BaseDecodeHead(BaseModule):
    def __init__(self, in_channels, channels, *, num_classes, dropout_ratio=0.1, conv_cfg=None, norm_cfg=None,
                 act_cfg=dict(type='ReLU'), in_index=-1, input_transform=None,
                 loss_decode=dict(type='CrossEntropyLoss', use_sigmoid=False, loss_weight=1.0), ignore_index=255,
                 sampler=None, align_corners=False, class_re_weight=False,
                 init_cfg=dict(type='Normal', std=0.01, override=dict(name='conv_seg')), get_mean_feat=False,
                 decoder_params=None):
        super().__init__(init_cfg)
        self._init_inputs(in_channels, in_index, input_transform)
        self.channels, self.num_classes, self.dropout_ratio = channels, num_classes, dropout_ratio
        self.conv_cfg, self.norm_cfg, self.act_cfg, self.in_index = conv_cfg, norm_cfg, act_cfg, in_index
        self.ignore_index, self.align_corners = ignore_index, align_corners
        if isinstance(loss_decode, dict):
            self.loss_decode = build_loss(loss_decode)
        elif isinstance(loss_decode, (list, tuple)):
            self.loss_decode = nn.ModuleList([build_loss(l) for l in loss_decode])
        else:
            raise TypeError(f'loss_decode must be a dict or sequence of dict, but got {type(loss_decode)}')
        if sampler is not None:
            raise S4FError('pixel samplers (OHEM) are not on the SETR hot path')
        self.sampler = None
        self.conv_seg = nn.Conv2d(channels, num_classes, kernel_size=1)
        if dropout_ratio > 0:
            raise S4FError('dropout_ratio must be 0 (as in every SETR config): the fused head has no dropout')
        self.dropout = None
        self.fp16_enabled = False

    def extra_repr(self):
        return f'input_transform={self.input_transform}, ignore_index={self.ignore_index}, align_corners={self.align_corners}'

    def _init_inputs(self, in_channels, in_index, input_transform):
        if input_transform is not None:
            raise S4FError('input_transform is not used by the SETR heads')
        self.input_transform = input_transform
        assert isinstance(in_channels, int)
        assert isinstance(in_index, int)
        self.in_channels, self.in_index = in_channels, in_index

    def _transform_inputs(self, inputs):
        return inputs[self.in_index]


@HEADS.register_module()
class SETRUPHead(BaseDecodeHead):
    def __init__(self, norm_layer=dict(type='LN', eps=1e-6, requires_grad=True), num_convs=1, up_scale=4, kernel_size=3,
                 use_addition_up_scale=False,
                 init_cfg=[dict(type='Constant', val=1.0, bias=0, layer='LayerNorm'),
                           dict(type='Normal', std=0.01, override=dict(name='conv_seg'))], **kwargs):
        assert kernel_size in [1, 3], 'kernel_size must be 1 or 3.'
        super().__init__(init_cfg=init_cfg, **kwargs)
        assert isinstance(self.in_channels, int)
        if kernel_size != 3 or use_addition_up_scale:
            raise S4FError('the implicit-GEMM head kernels are built for kernel_size=3 without the extra up-scale')
        if int(up_scale) != up_scale or up_scale < 1:
            raise S4FError('up_scale must be a positive integer')
        if self.align_corners:
            raise S4FError('align_corners=True is not used by the SETR configs')
        if self.num_classes > LOGIT_LD:
            raise S4FError(f'at most {LOGIT_LD} classes supported by the fused logits kernels')
        assert norm_layer['type'] == 'LN'
        self.norm = nn.LayerNorm(self.in_channels, eps=norm_layer.get('eps', 1e-5))
        self.up_scale, self.num_convs = int(up_scale), num_convs
        self.up_convs = nn.ModuleList()
        in_channels, out_channels = self.in_channels, self.channels
        for _ in range(num_convs):
            self.up_convs.append(nn.Sequential(
                ConvModule(in_channels=in_channels, out_channels=out_channels, kernel_size=kernel_size, stride=1,
                           padding=int(kernel_size - 1) // 2, norm_cfg=self.norm_cfg, act_cfg=self.act_cfg),
                Upsample(scale_factor=up_scale, mode='bilinear', align_corners=self.align_corners)))
            in_channels = out_channels
        self._nbt = [[0] for _ in range(num_convs)]
        self._store = None
        self._store_owner = False
        self._register_state_dict_hook(SETRUPHead._sd_hook)
        self._register_load_state_dict_pre_hook(self._load_hook)

    # num_batches_tracked is counted on the host (no per-call device op) and materialised on state_dict()
    @staticmethod
    def _sd_hook(module, state_dict, prefix, local_metadata):
        for k, c in enumerate(module._nbt):
            key = f'{prefix}up_convs.{k}.0.bn.num_batches_tracked'
            if key in state_dict:
                state_dict[key] = state_dict[key] + c[0]
        return state_dict

    def _load_hook(self, state_dict, prefix, *args):
        """a loaded num_batches_tracked already contains what the host counters held when it was saved"""
        for k, c in enumerate(self._nbt):
            if f'{prefix}up_convs.{k}.0.bn.num_batches_tracked' in state_dict:
                c[0] = 0

    # ------------------------------------------------------------------ store plumbing
    def _attach_store(self, store):
        self._store = store

    def _ensure_store(self, device):
        if self._store is None:
            self._store_owner = True
            self._store = ParamStore([('head', self, '')], with_grad=True)
        if self._store_owner:
            self._store.ensure(device, runtime.compute_dtype())
            self._store.ensure_grads()
            self._store.sync_shadow()
        return self._store

    def _hp(self, grid):
        """parameter / buffer / config bundle of the fused head node; cached per (grid, mode, arena generation): the
        Parameter objects are stable, the BN buffers are re-pointed whenever the ParamStore rebuilds its arena"""
        training = self.training and not getattr(self, '_eval_override', False)
        key = (grid, training, getattr(self._store, 'generation', None))
        cache = self.__dict__.setdefault('_hp_cache', {})
        hp = cache.get(key)
        if hp is None:
            if len(cache) > 8:
                cache.clear()
            hp = cache[key] = self._build_hp(grid, training)
        return hp

    def _build_hp(self, grid, training):
        convs = []
        for k, seq in enumerate(self.up_convs):
            cm = seq[0]
            convs.append(dict(w=cm.conv.weight, bn_w=cm.bn.weight, bn_b=cm.bn.bias, rm=cm.bn.running_mean,
                              rv=cm.bn.running_var, nbt=self._nbt[k]))
        cm0 = self.up_convs[0][0]
        return dict(grid=grid, norm_w=self.norm.weight, norm_b=self.norm.bias, ln_eps=self.norm.eps, convs=convs,
                    seg_w=self.conv_seg.weight, seg_b=self.conv_seg.bias, num_classes=self.num_classes,
                    up_scale=self.up_scale, bn_eps=cm0.bn.eps, bn_momentum=cm0.bn.momentum, sync_bn=cm0.sync,
                    ignore_index=self.ignore_index, training=training)

    def _params(self):
        ps = self.__dict__.get('_params_cache')
        if ps is not None and ps[0] is self.norm.weight:
            return ps
        ps = self.__dict__['_params_cache'] = self._build_params()
        return ps

    def _build_params(self):
        ps = [self.norm.weight, self.norm.bias]
        for seq in self.up_convs:
            cm = seq[0]
            ps += [cm.conv.weight, cm.bn.weight, cm.bn.bias]
        return ps + [self.conv_seg.weight, self.conv_seg.bias]

    @staticmethod
    def _tokens_of(x):
        """token-major source of a backbone output: the [B, T+1, E] tensor riding on the NCHW view, or a
        re-packed copy for a foreign NCHW tensor (off the hot path)."""
        tok = getattr(x, '_s4f_tokens', None)
        if tok is not None:
            return tok, x._s4f_grid
        n, c, h, w = x.shape
        t = x.reshape(n, c, h * w).transpose(2, 1)
        tok = torch.cat((torch.zeros(n, 1, c, device=x.device, dtype=torch.float32), t.to(torch.float32)), dim=1).contiguous()
        return tok, (h, w)

    # ------------------------------------------------------------------ mmseg call protocol
    def forward(self, x, PatchMix_N=0, PatchMixIndex=None, return_last_feat=False):
        """setr_up_head.py:92-111 -> logits [B, num_classes, H, W] fp32 (no autograd through this path)."""
        if PatchMix_N != 0 or return_last_feat:
            raise S4FError('PatchMix un-shuffle / return_last_feat belong to the "ours" additions (SURVEY §8f-1)')
        x = self._transform_inputs(x)
        if not x.is_cuda:
            raise S4FError('SETRUPHead runs on the HIP kernels only (GPU tensors required)')
        store = self._ensure_store(x.device)
        tokens, grid = self._tokens_of(x)
        with torch.no_grad():
            logits, (Bn, h, w), _ = head_forward(tokens.detach(), (hp := self._hp(grid)), store, training=hp['training'], save=False)
            s = self.up_scale
            out = torch.empty(Bn, self.num_classes, h * s, w * s, device=x.device, dtype=torch.float32)
            K.up_logits_nchw(logits, out, Bn, h, w, self.num_classes, LOGIT_LD, s)
        return out

    def logits_lowres(self, inputs):
        """low-resolution logits [B*h*w, LOGIT_LD] + geometry (teacher path: the upsample is fused downstream)"""
        x = self._transform_inputs(inputs)
        store = self._ensure_store(x.device)
        tokens, grid = self._tokens_of(x)
        with torch.no_grad():
            logits, geom, _ = head_forward(tokens.detach(), (hp := self._hp(grid)), store, training=hp['training'], save=False)
        return logits, geom

    def fused_loss(self, inputs, labels_u8, loss_weight, ncr_teacher_lo=None, token_maps=None):
        """loss_weight * CE(mean over all pixels) of this head on `inputs`, as one fused node.
        token_maps (fwd, bwd): un-shuffle the tokens of a patch-shuffled image first (forward(..., PatchMix_N, PatchMixIndex) of
        the reference, decode_head.py:186-212, 242-249); ncr_teacher_lo: also return the negative-class-ranking loss against
        the teacher's low-resolution logits -> (loss, loss_ncr)."""
        x = self._transform_inputs(inputs)
        store = self._ensure_store(x.device)
        tokens, grid = self._tokens_of(x)
        if token_maps is not None:
            from .functional import TokenGatherFn
            tokens = TokenGatherFn.apply(tokens, token_maps[0], token_maps[1])
        return HeadLossFn.apply(tokens, labels_u8, loss_weight, self._hp(grid), store, ncr_teacher_lo, *self._params())

    def _loss_labels(self, img_metas, gt_semantic_seg):
        if img_metas and 'PatchMix_N' in img_metas[0]:
            raise S4FError('PatchMix un-shuffle belongs to the "ours" additions (SURVEY §8f-1)')
        ld = self.loss_decode
        if isinstance(ld, nn.ModuleList) or not isinstance(ld, CrossEntropyLoss) or ld.class_weight is not None \
                or ld.reduction != 'mean' or ld.avg_non_ignore:
            raise S4FError('the fused head loss implements CrossEntropyLoss(reduction="mean", avg_non_ignore=False)')
        labels = gt_semantic_seg
        if labels.dim() == 4:
            labels = labels.squeeze(1)
        if labels.dtype != torch.uint8:
            labels = labels.to(torch.uint8)
        return labels.contiguous()

    @staticmethod
    def fused_losses_lockstep(calls):
        """calls = [(head, inputs, labels_u8, loss_weight)] of structurally identical heads (or of one head, several times):
        ONE autograd node whose SyncBN layers exchange their statistics together (functional.MultiHeadLossFn).  Returns
        the loss tensors in call order."""
        from .functional import MultiHeadLossFn
        toks, metas, prms = [], [], []
        store = None
        for hd, inputs, labels_u8, lw in calls:
            x = hd._transform_inputs(inputs)
            store = hd._ensure_store(x.device)
            tokens, grid = hd._tokens_of(x)
            p_ = hd._params()
            toks.append(tokens)
            metas.append((lw, hd._hp(grid), len(p_), labels_u8))
            prms.extend(p_)
        return list(MultiHeadLossFn.apply(store, metas, *toks, *prms))

    @staticmethod
    def forward_train_lockstep(heads, inputs, img_metas, gt_semantic_seg, train_cfg):
        """forward_train of several structurally identical heads on the same inputs and labels, in lockstep; returns the
        per-head loss dicts in order."""
        labels = [hd._loss_labels(img_metas, gt_semantic_seg) for hd in heads]
        losses = SETRUPHead.fused_losses_lockstep([(hd, inputs, lb, hd.loss_decode.loss_weight) for hd, lb in zip(heads, labels)])
        return [{hd.loss_decode.loss_name: l} for hd, l in zip(heads, losses)]

    def forward_train(self, inputs, img_metas, gt_semantic_seg, train_cfg):
        """decode_head.py:225-259 + losses (:318-355) in one fused node."""
        if img_metas and 'PatchMix_N' in img_metas[0]:
            raise S4FError('PatchMix un-shuffle belongs to the "ours" additions (SURVEY §8f-1)')
        ld = self.loss_decode
        if isinstance(ld, nn.ModuleList) or not isinstance(ld, CrossEntropyLoss) or ld.class_weight is not None \
                or ld.reduction != 'mean' or ld.avg_non_ignore:
            raise S4FError('the fused head loss implements CrossEntropyLoss(reduction="mean", avg_non_ignore=False)')
        labels = gt_semantic_seg
        if labels.dim() == 4:
            labels = labels.squeeze(1)
        if labels.dtype != torch.uint8:
            labels = labels.to(torch.uint8)
        return {ld.loss_name: self.fused_loss(inputs, labels.contiguous(), ld.loss_weight)}

    def forward_get_logits(self, inputs, train_cfg, img_metas=None):
        return self.forward(inputs)

    def forward_test(self, inputs, img_metas, test_cfg, return_last_feat=False):
        return self.forward(inputs, return_last_feat=return_last_feat)

    def cls_seg(self, feat):
        raise S4FError('cls_seg is fused into the head kernels')

    def losses(self, seg_logit, seg_label):
        """decode_head.py:318-355 on already materialised full-size logits (API compatibility path)."""
        if seg_logit.shape[2:] != seg_label.shape[2:]:
            raise S4FError('losses(): only the identity resize is on the hot path')
        loss = dict()
        lds = [self.loss_decode] if not isinstance(self.loss_decode, nn.ModuleList) else self.loss_decode
        for ld in lds:
            v = ld(seg_logit, seg_label.squeeze(1), weight=None, ignore_index=self.ignore_index)
            loss[ld.loss_name] = v if ld.loss_name not in loss else loss[ld.loss_name] + v
        return loss
