"""Fused SGD over the flat arenas + the reference's optimiser construction and LR schedule.

build_optimizer mirrors mmseg/core/builder.py:22-33 with mmcv's DefaultOptimizerConstructor semantics as used by
configs/setr/*.py: SGD(momentum 0.9, weight_decay 0), one param group PER parameter, lr x10 where a custom key
('head') occurs in the parameter name; parameters that do not require grad (the EMA teacher) are listed too and
never updated.  step() collapses the per-parameter groups back into one launch per arena range whenever the
groups of that range share their lr (they always do under the poly schedule).
PolyLR: mmcv PolyLrUpdaterHook(by_epoch=False): lr_t = (lr_0 - min_lr) * (1 - t/T)^power + min_lr.
"""
import torch

from . import kernels as K
from ._lib import S4FError


class S4FSGD(torch.optim.Optimizer):
    def __init__(self, model, lr, momentum=0.9, weight_decay=0.0, paramwise_cfg=None, dampening=0, nesterov=False):
        if weight_decay != 0 or dampening != 0 or nesterov:
            raise S4FError('the fused SGD implements momentum SGD with weight_decay=0 (as the SETR configs use)')
        custom = (paramwise_cfg or {}).get('custom_keys', {})
        keys = sorted(sorted(custom.keys()), key=len, reverse=True)
        groups = []
        for name, p_ in model.named_parameters():
            g = {'params': [p_], 'name': name}
            if p_.requires_grad:
                for k in keys:
                    if k in name:
                        g['lr'] = lr * custom[k].get('lr_mult', 1.)
                        break
            groups.append(g)
        super().__init__(groups, dict(lr=lr, momentum=momentum, weight_decay=weight_decay))
        self.model = model
        self._ranges = None

    def _plan(self):
        """[(group_name, a, b, [param_group indices])] over the student arena"""
        store = self.model.student_store
        if store is None:
            raise S4FError('optimizer.step() before the first forward: the arenas do not exist yet')
        pg_of = {id(g['params'][0]): i for i, g in enumerate(self.param_groups)}
        plan = []
        for gname, rng in store.group_ranges.items():
            a, b = rng['params']
            idx = [pg_of[id(e.module._parameters[e.attr])] for e in store.entries
                   if e.is_param and e.group == gname and id(e.module._parameters[e.attr]) in pg_of]
            plan.append((gname, a, b, idx))
        return store, plan

    @torch.no_grad()
    def step(self, closure=None, grad_scale=1.0):
        loss = closure() if closure is not None else None
        from .functional import join_side_streams
        join_side_streams()          # weight-gradient kernels run on a side stream
        store, plan = self._plan()
        first = store.first_sgd_step
        for gname, a, b, idx in plan:
            lrs = {self.param_groups[i]['lr'] for i in idx}
            moms = {self.param_groups[i]['momentum'] for i in idx}
            if len(lrs) == 1 and len(moms) == 1:
                pt = store.flat_t[a:b] if store.flat_t is not None else None
                K.sgd_momentum(store.flat[a:b], store.grad[a:b], store.mom[a:b], pt, b - a, lrs.pop(), moms.pop(),
                               grad_scale, first, store.dtype)
            else:
                for e in store.entries:
                    if not (e.is_param and e.group == gname):
                        continue
                    g = self.param_groups[[i for i in idx if self.param_groups[i]['params'][0] is e.module._parameters[e.attr]][0]]
                    n = (e.numel + 63) // 64 * 64
                    pt = store.flat_t[e.off:e.off + n] if store.flat_t is not None else None
                    K.sgd_momentum(store.flat[e.off:e.off + n], store.grad[e.off:e.off + n], store.mom[e.off:e.off + n], pt,
                                   n, g['lr'], g['momentum'], grad_scale, first, store.dtype)
        store.first_sgd_step = False
        if store.flat_t is not None and store._T_items:
            store.sync_T(eager=True)      # transposed operand shadows follow the bf16 shadow the SGD kernels just wrote
        return loss

    def zero_grad(self, set_to_none=False):
        store = self.model.student_store
        if store is not None and store.grad is not None:
            store.zero_grad()
        else:
            super().zero_grad(set_to_none=set_to_none)


def build_optimizer(model, cfg):
    """mmseg/core/builder.py:22-33.  cfg: dict(type='SGD', lr, momentum, weight_decay, paramwise_cfg=...)."""
    cfg = dict(cfg)
    if cfg.pop('type', 'SGD') != 'SGD':
        raise S4FError('only SGD is on the hot path')
    cfg.pop('constructor', None)
    if hasattr(model, 'module'):
        model = model.module
    return S4FSGD(model, **cfg)


class PolyLR:
    """mmcv PolyLrUpdaterHook, by_epoch=False (configs/_base_/schedules/schedule_80k_pascal_1over8.py:5)."""

    def __init__(self, optimizer, max_iters, power=0.9, min_lr=1e-4):
        self.optimizer, self.max_iters, self.power, self.min_lr = optimizer, max_iters, power, min_lr
        for g in optimizer.param_groups:
            g.setdefault('initial_lr', g['lr'])
        self.base_lr = [g['initial_lr'] for g in optimizer.param_groups]

    def get_lr(self, base_lr, it):
        coeff = (1 - it / self.max_iters) ** self.power
        return (base_lr - self.min_lr) * coeff + self.min_lr

    def step(self, it):
        """call before the iteration `it` (0-based), as the hook's before_train_iter does"""
        for g, b in zip(self.optimizer.param_groups, self.base_lr):
            g['lr'] = self.get_lr(b, it)
