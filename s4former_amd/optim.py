"""Fused SGD over the flat arenas + the reference's optimiser construction and LR schedule.

build_optimizer mirrors mmseg/core/builder.py:22-33 with mmcv's DefaultOptimizerConstructor semantics as used by
configs/setr/*.py: SGD(momentum 0.9, weight_decay 0), one param group PER parameter, lr x10 where a custom key
('head') occurs in the parameter name; parameters that do not require grad (the EMA teacher) are listed too and
never updated.  step() collapses the per-parameter groups back into one launch per arena range whenever the
groups of that range share their lr (they always do under the poly schedule).
PolyLR: mmcv PolyLrUpdaterHook(by_epoch=False): lr_t = (lr_0 - min_lr) * (1 - t/T)^power + min_lr.
"""
import os

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
        self._plan_cache = None
        self._eager = None           # (store, reducer, grad_scale) when ranges are stepped during backward
        self._eager_done = []        # [(a, b)] arena ranges already stepped in this iteration
        self._stream = None
        # Extension (off by default: torch leaves .grad untouched by step()): the SGD kernels write zeros over the gradient
        # ranges they consume, and the zero_grad() of the next iteration - a 400 MB fill alone on the GPU at the start of the
        # step - has nothing left to do (ParamStore.grad_clean).  bench.py switches it on.
        self.fused_zero_grad = os.environ.get('S4F_FUSED_ZERO_GRAD', '0') == '1'

    def _plan(self):
        """[(group_name, a, b, [param_group indices])] over the student arena"""
        store = self.model.student_store
        if store is None:
            raise S4FError('optimizer.step() before the first forward: the arenas do not exist yet')
        if self._plan_cache is not None and self._plan_cache[0] is store:
            return self._plan_cache
        pg_of = {id(g['params'][0]): i for i, g in enumerate(self.param_groups)}
        plan = []
        for gname, rng in store.group_ranges.items():
            a, b = rng['params']
            idx = [pg_of[id(e.module._parameters[e.attr])] for e in store.entries
                   if e.is_param and e.group == gname and id(e.module._parameters[e.attr]) in pg_of]
            plan.append((gname, a, b, idx))
        self._plan_cache = (store, plan)
        return store, plan

    def _plan_covers_all_params(self, store, plan):
        key = id(plan)
        if getattr(self, '_cover_cache', (None, None))[0] != key:
            ok = all(any(a <= e.off and e.off + e.numel <= b for _, a, b, _ in plan)
                     for e in store.entries if e.is_param and e.module._parameters[e.attr].requires_grad)
            self._cover_cache = (key, ok)
        return self._cover_cache[1]

    # ------------------------------------------------------------------ eager mode
    def attach_eager(self, store, reducer=None, grad_scale=1.0):
        """Step every arena range as soon as its gradient is final (ParamStore.range_done: the end of an encoder layer's
        or a head's backward) instead of after the whole backward: the HBM-bound update then runs behind the MFMA-bound
        rest of the backward pass on its own stream.  Same kernels, same operands, same result; step() covers what is
        left.  With a GradReducer the range is all-reduced first and stepped on the communication stream behind it."""
        self._eager = (store, reducer, float(grad_scale))
        store.on_range_done = self._range_done
        return self

    def _uniform(self, idx):
        lrs = {self.param_groups[i]['lr'] for i in idx}
        moms = {self.param_groups[i]['momentum'] for i in idx}
        return (lrs.pop(), moms.pop()) if len(lrs) == 1 and len(moms) == 1 else None

    @torch.no_grad()
    def _range_done(self, a, b):
        store, reducer, scale = self._eager
        if any(a < db and da < b for da, db, _ in self._eager_done):
            raise S4FError(f'parameter range [{a}, {b}) was reported final twice before optimizer.step() (gradient '
                           'accumulation over several backward passes needs S4F_EAGER_SGD=0)')
        handle = None
        if reducer is not None:
            n0 = len(reducer._handles)
            reducer._range_done(a, b)
            if len(reducer._handles) > n0:
                handle = reducer._handles[-1]
        _, plan = self._plan()
        # the parts of [a, b) that are parameters of ONE learning-rate setting each (a span may cover several adjacent head groups
        # and the running statistics between them: those have no gradient and are not stepped)
        pieces = []
        for _, ga, gb, idx in plan:
            lo, hi = max(a, ga), min(b, gb)
            if lo < hi:
                lm = self._uniform(idx)
                if lm is not None:                   # (mixed learning rates inside a group: left to step())
                    pieces.append((lo, hi, lm))
        if not pieces or store.grad is None:
            return
        from .functional import extra_streams, side_stream
        if handle is not None and reducer._stream is not None:
            stream = reducer._stream                 # behind the all-reduce of this very range
        else:
            # a stream of its own (32.45 vs 32.65 ms per step on the weight-gradient stream; as the 4th stream used it shares
            # the weight gradients' hardware queue - on the head streams' queues it was measured slower) - unless the chain runs
            # on a high-priority stream: a fifth stream of normal priority then starves (32 -> 40 ms, tools/exp/prio_ab.sh)
            cur = torch.cuda.current_stream()
            if os.environ.get('S4F_EAGER_STREAM', 'new') == 'new' and cur.priority >= 0:
                if self._stream is None:
                    self._stream = torch.cuda.Stream()
                stream = self._stream
            else:
                stream = side_stream(store.flat.device)
            cur = torch.cuda.current_stream()
            if stream != cur:
                stream.wait_stream(cur)
            for st in extra_streams():
                if st != stream and st != cur:
                    stream.wait_stream(st)
        with torch.cuda.stream(stream):
            if handle is not None:
                handle.wait()
            for lo, hi, lm in pieces:
                pt = store.flat_t[lo:hi] if store.flat_t is not None else None
                K.sgd_momentum(store.flat[lo:hi], store.grad[lo:hi], store.mom[lo:hi], pt, hi - lo, lm[0], lm[1], scale,
                               store.first_sgd_step, store.dtype, zero_grad=self.fused_zero_grad)
            store.sync_T_range(a, b)                 # the transposed operand shadows of this range, behind its update
            hook = getattr(self.model, '_ema_behind_update', None)
            if hook is not None and not (handle is not None and reducer._stream is not None):
                # round 5: the teacher's NEXT value of this range, into its second arena
                for lo, hi, _ in pieces:
                    hook(lo, hi)
            ema_ev = None
            if hook is not None and handle is not None and reducer._stream is not None:
                ema_ev = torch.cuda.Event()
                ema_ev.record()
        if ema_ev is not None:
            # Round 6 (N > 1): NOT on the communication stream - the next bucket's all-reduce is queued there and would wait behind
            # 3 fp32 arenas + the bf16 shadow of EMA traffic per range.  The out-of-place EMA runs on the optimiser's own stream
            # behind an event recorded after this range's SGD; step() joins that stream before the pending update is published.
            if self._stream is None:
                self._stream = torch.cuda.Stream()
            self._stream.wait_event(ema_ev)
            with torch.cuda.stream(self._stream):
                for lo, hi, _ in pieces:
                    hook(lo, hi)
            self._eager_done.append((pieces[0][0], pieces[0][0], self._stream))      # (empty range: only the stream is joined)
        for lo, hi, _ in pieces:
            self._eager_done.append((lo, hi, stream))

    @torch.no_grad()
    def step(self, closure=None, grad_scale=1.0):
        loss = closure() if closure is not None else None
        from .functional import join_side_streams
        join_side_streams()          # weight-gradient kernels run on a side stream
        store, plan = self._plan()
        first = store.first_sgd_step
        done = sorted((a, b) for a, b, _ in self._eager_done)
        cur = torch.cuda.current_stream()
        for st in {id(x[2]): x[2] for x in self._eager_done}.values():
            cur.wait_stream(st)
        self._eager_done = []
        for gname, a, b, idx in plan:
            lm = self._uniform(idx)
            if lm is not None:
                pos = a                               # what the eager mode has not stepped yet
                for da, db in [d for d in done if a <= d[0] and d[1] <= b] + [(b, b)]:
                    if da > pos:
                        pt = store.flat_t[pos:da] if store.flat_t is not None else None
                        K.sgd_momentum(store.flat[pos:da], store.grad[pos:da], store.mom[pos:da], pt, da - pos, lm[0], lm[1],
                                       grad_scale, first, store.dtype, zero_grad=self.fused_zero_grad)
                    pos = max(pos, db)
            else:
                for e in store.entries:
                    if not (e.is_param and e.group == gname):
                        continue
                    g = self.param_groups[[i for i in idx if self.param_groups[i]['params'][0] is e.module._parameters[e.attr]][0]]
                    n = (e.numel + 63) // 64 * 64
                    pt = store.flat_t[e.off:e.off + n] if store.flat_t is not None else None
                    K.sgd_momentum(store.flat[e.off:e.off + n], store.grad[e.off:e.off + n], store.mom[e.off:e.off + n], pt,
                                   n, g['lr'], g['momentum'], grad_scale, first, store.dtype, zero_grad=self.fused_zero_grad)
        store.first_sgd_step = False
        fin = getattr(self.model, '_ema_finish', None)
        if fin is not None:
            fin()                                    # what the eager ranges did not cover (running statistics, ranges stepped here)
        # every parameter range of the plan has been consumed and zeroed; the flag may only be raised if the plan really covers every
        # trainable parameter of the arena (a gradient outside it would survive the skipped zero_grad() and accumulate silently)
        store.grad_clean = bool(self.fused_zero_grad) and self._plan_covers_all_params(store, plan)
        if store.flat_t is not None and store._T_items:
            store.sync_T(eager=True)      # transposed operand shadows follow the bf16 shadow the SGD kernels just wrote
        return loss

    # ------------------------------------------------------------------ checkpoint contents
    # The reference's runner saves optimizer.state_dict() (mmcv CheckpointHook, save_optimizer=True): torch's layout
    # {'state': {param index: {'momentum_buffer': tensor}}, 'param_groups': [...]}.  The momentum lives in the arena
    # (ParamStore.mom), so the dict is assembled from / scattered into it here; a resumed run continues with its momentum.
    def state_dict(self):
        sd = super().state_dict()
        store = self.model.student_store
        if store is None or store.mom is None or store.first_sgd_step:
            return sd
        idx_of = {id(g['params'][0]): i for i, g in enumerate(self.param_groups)}
        state = {}
        for e in store.entries:
            if not e.is_param:
                continue
            prm = e.module._parameters[e.attr]
            if prm.requires_grad and id(prm) in idx_of:
                state[idx_of[id(prm)]] = {'momentum_buffer': store._view(store.mom, e).detach().clone()}
        sd['state'] = state
        return sd

    def load_state_dict(self, state_dict):
        state = state_dict.get('state', {})
        super().load_state_dict(dict(state_dict, state={}))
        self._plan_cache = None
        if not state:
            return
        store = self.model.student_store
        if store is None or store.mom is None:
            raise S4FError('optimizer.load_state_dict() with momentum buffers needs the arenas: call '
                           'model.ensure_engine(device) (or run one forward) first')
        by_idx = {i: g['params'][0] for i, g in enumerate(self.param_groups)}
        with torch.no_grad():
            store.mom.zero_()
            for k, st in state.items():
                buf = st.get('momentum_buffer') if isinstance(st, dict) else None
                if buf is None:
                    continue
                e = store.entry(by_idx[int(k)])
                store._view(store.mom, e).copy_(buf.to(store.mom.device))
        store.first_sgd_step = False

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
