"""CrossEntropyLoss with the reference's constructor / forward signature and reduction semantics
(reference mmseg/models/losses/cross_entropy_loss.py:12-63,188-297 and losses/utils.py:48-80), computed by the
s4f_ce_fwd / s4f_ce_bwd HIP kernels (softmax path only: use_sigmoid / use_mask are not on the SETR hot path)."""
import torch
import torch.nn as nn
from torch.autograd import Function

from . import kernels as K
from ._lib import S4FError
from .registry import LOSSES


class _CEFn(Function):
    @staticmethod
    def forward(ctx, logits, label, class_weight, ignore_index):
        N, C = logits.shape[0], logits.shape[1]
        spatial = 1
        for d in logits.shape[2:]:
            spatial *= d
        lg = logits.to(torch.float32).contiguous()
        lb = label.to(torch.int64).contiguous()
        loss = torch.empty(lb.shape, device=logits.device, dtype=torch.float32)
        K.ce_fwd(lg, lb, class_weight, loss, N, C, spatial, ignore_index)
        ctx.save_for_backward(lg, lb, class_weight if class_weight is not None else torch.empty(0))
        ctx.meta = (N, C, spatial, ignore_index, class_weight is not None)
        return loss

    @staticmethod
    def backward(ctx, dloss):
        lg, lb, cw = ctx.saved_tensors
        N, C, spatial, ignore_index, has_cw = ctx.meta
        dlogits = torch.empty_like(lg)
        K.ce_bwd(lg, lb, cw if has_cw else None, dloss.to(torch.float32).contiguous(), dlogits, N, C, spatial, ignore_index)
        return dlogits, None, None, None


def reduce_loss(loss, reduction):
    """losses/utils.py:28-45"""
    if reduction == 'none':
        return loss
    if reduction == 'mean':
        return loss.mean()
    if reduction == 'sum':
        return loss.sum()
    raise ValueError(reduction)


def weight_reduce_loss(loss, weight=None, reduction='mean', avg_factor=None):
    """losses/utils.py:48-80"""
    if weight is not None:
        assert weight.dim() == loss.dim()
        if weight.dim() > 1:
            assert weight.size(1) == 1 or weight.size(1) == loss.size(1)
        loss = loss * weight
    if avg_factor is None:
        loss = reduce_loss(loss, reduction)
    else:
        if reduction == 'mean':
            eps = torch.finfo(torch.float32).eps
            loss = loss.sum() / (avg_factor + eps)
        elif reduction != 'none':
            raise ValueError('avg_factor can not be used with reduction="sum"')
    return loss


def cross_entropy(pred, label, weight=None, class_weight=None, reduction='mean', avg_factor=None, ignore_index=-100,
                  avg_non_ignore=False):
    """cross_entropy_loss.py:12-63: per-element CE with ignored elements contributing 0, then
    weight_reduce_loss (avg_non_ignore=False => mean over ALL elements, Q5)."""
    if not pred.is_cuda:
        raise S4FError('CrossEntropyLoss runs on the HIP kernels only (GPU tensors required)')
    loss = _CEFn.apply(pred, label, class_weight, int(ignore_index))
    if (avg_factor is None) and avg_non_ignore and reduction == 'mean':
        avg_factor = label.numel() - (label == ignore_index).sum().item()
    if weight is not None:
        weight = weight.float()
    return weight_reduce_loss(loss, weight=weight, reduction=reduction, avg_factor=avg_factor)


@LOSSES.register_module()
class CrossEntropyLoss(nn.Module):
    def __init__(self, use_sigmoid=False, use_mask=False, reduction='mean', class_weight=None, loss_weight=1.0,
                 loss_name='loss_ce', avg_non_ignore=False):
        super().__init__()
        assert (use_sigmoid is False) or (use_mask is False)
        if use_sigmoid or use_mask:
            raise S4FError('use_sigmoid / use_mask are outside the SETR hot path')
        self.use_sigmoid, self.use_mask = use_sigmoid, use_mask
        self.reduction, self.loss_weight = reduction, loss_weight
        self.class_weight = class_weight
        self.avg_non_ignore = avg_non_ignore
        self.cls_criterion = cross_entropy
        self._loss_name = loss_name

    def extra_repr(self):
        return f'avg_non_ignore={self.avg_non_ignore}'

    def forward(self, cls_score, label, weight=None, avg_factor=None, reduction_override=None, ignore_index=-100, **kwargs):
        assert reduction_override in (None, 'none', 'mean', 'sum')
        reduction = reduction_override if reduction_override else self.reduction
        if self.class_weight is not None:
            class_weight = cls_score.new_tensor(self.class_weight, dtype=torch.float32)
        else:
            class_weight = None
        return self.loss_weight * self.cls_criterion(cls_score, label, weight, class_weight=class_weight,
                                                     reduction=reduction, avg_factor=avg_factor,
                                                     avg_non_ignore=self.avg_non_ignore, ignore_index=ignore_index,
                                                     **kwargs)

    @property
    def loss_name(self):
        return self._loss_name
