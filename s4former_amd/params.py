"""Flat parameter arenas (MI355X-first memory layout of the model state).

All parameters (and the BatchNorm running statistics) of a replica live in ONE fp32 arena, with a second arena
of identical layout for gradients, one for SGD momentum, and a bf16 shadow arena for the MFMA operands.  The
nn.Parameters the mmseg-style modules expose are views into the arena, so state_dict()/load_state_dict()/
optimizers keep working, while EMA (reference encoder_decoder.py:1044-1066), SGD and the gradient all-reduce
each become one launch over a contiguous range instead of a Python loop over ~200 tensors.

3x3 conv weights are stored physically as [co][ky][kx][ci] (torch channels_last strides on the logical
[co,ci,ky,kx] parameter): that is the B-operand layout of the implicit-GEMM kernels.
"""
import os

import torch
import torch.nn as nn

from . import kernels as K
from ._lib import BF16, S4FError

ALIGN = 64  # elements


def _is_conv3x3(t):
    return t.dim() == 4 and t.shape[2] == 3 and t.shape[3] == 3


class _Entry:
    __slots__ = ('module', 'attr', 'is_param', 'name', 'group', 'shape', 'numel', 'off', 'cl', 'v_phys', 'v_grad', 'v_shadow',
                 'v_T', 'v_log', 'alt')


# A/B switch (round 4): hand a parameter group's pending spans to the reducer as soon as its last range is final
GROUP_FLUSH = os.environ.get('S4F_GROUP_FLUSH', '1') != '0'


class ParamStore:
    """groups: list of (group_name, module, prefix).  Parameters come first inside each group, then the BN
    running_mean / running_var buffers (num_batches_tracked stays an ordinary buffer)."""

    def __init__(self, groups, with_grad=True):
        self.groups = groups
        self.with_grad = with_grad
        self.entries = []
        self.by_id = {}
        self.flat = self.flat_t = self.grad = self.mom = None
        self.dtype = None
        self.group_ranges = {}       # group -> dict(params=(a,b), all=(a,b))
        self._versions = None
        self._shadow_dirty = True
        self._dirty_marks = 0
        self.first_sgd_step = True
        self._collect()

    # ------------------------------------------------------------------ layout
    def _collect(self):
        off = 0
        for gname, module, prefix in self.groups:
            g0 = off
            plist = [(n, p_, True) for n, p_ in module.named_parameters()]
            blist = [(n, b, False) for n, b in module.named_buffers()
                     if b is not None and b.dtype == torch.float32 and n.split('.')[-1] in ('running_mean', 'running_var')]
            pend = None
            for lst in (plist, blist):
                for n, t, is_param in lst:
                    e = _Entry()
                    mod = module
                    parts = n.split('.')
                    for s in parts[:-1]:
                        mod = getattr(mod, s)
                    e.module, e.attr, e.is_param = mod, parts[-1], is_param
                    e.name = f'{prefix}.{n}' if prefix else n
                    e.group = gname
                    e.shape = tuple(t.shape)
                    e.numel = t.numel()
                    e.cl = _is_conv3x3(t)
                    e.off = off
                    off += (e.numel + ALIGN - 1) // ALIGN * ALIGN
                    self.entries.append(e)
                if lst is plist:
                    pend = off
            self.group_ranges[gname] = dict(params=(g0, pend), all=(g0, off))
        self.total = off

    def _tensor(self, e):
        return e.module._parameters[e.attr] if e.is_param else e.module._buffers[e.attr]

    def _view(self, arena, e):
        v = arena[e.off:e.off + e.numel]
        if e.cl:
            co, ci, kh, kw = e.shape
            return v.view(co, kh, kw, ci).permute(0, 3, 1, 2)
        return v.view(e.shape)

    def layout_signature(self, upto_group=None):
        sig = []
        for e in self.entries:
            sig.append((e.shape, e.off, e.is_param))
        return sig

    # ------------------------------------------------------------------ build / validity
    def build(self, device, dtype_code):
        """(re)allocate the arenas on `device` and re-point every parameter/buffer into them"""
        self.dtype = dtype_code
        flat = torch.zeros(self.total, device=device, dtype=torch.float32)
        for e in self.entries:
            src = self._tensor(e).detach()
            self._view(flat, e).copy_(src.to(device))
        self.flat = flat
        self.flat_t = torch.empty(self.total, device=device, dtype=torch.bfloat16) if dtype_code == BF16 else None
        if self.with_grad:
            self.grad = torch.zeros(self.total, device=device, dtype=torch.float32)
            self.mom = torch.zeros(self.total, device=device, dtype=torch.float32)
        self.by_id = {}
        for e in self.entries:
            v = self._view(flat, e)
            if e.is_param:
                prm = e.module._parameters[e.attr]
                prm.data = v
                if self.with_grad and prm.requires_grad:
                    prm.grad = self._view(self.grad, e)
                self.by_id[id(prm)] = e
            else:
                e.module._buffers[e.attr] = v
                self.by_id[id(v)] = e
        # per-entry physical views, made once (slicing costs a few host microseconds and the step asks ~300 times)
        for e in self.entries:
            e.v_phys = flat[e.off:e.off + e.numel]
            e.v_grad = self.grad[e.off:e.off + e.numel] if self.grad is not None else None
            src = self.flat_t if self.flat_t is not None else flat
            e.v_shadow = src[e.off:e.off + e.numel]
        self.generation = getattr(self, 'generation', 0) + 1
        self._alt = None             # (flat, flat_t) of the second arena pair (enable_double)
        self.flat_T = None           # transposed bf16 shadows (same offsets as flat_t), allocated on first use
        self._T_items = {}           # entry offset -> (R, T, C)
        self._T_table = None         # (device int64 table, host item list)
        self._T_fresh = False
        self._T_event = None
        self._shadow_dirty = True
        self._versions = None
        self.first_sgd_step = True
        return self

    # ------------------------------------------------------------------ second arena pair (round 5: the double-buffered teacher)
    def enable_double(self):
        """allocate a second (fp32, T shadow) arena pair of the same layout and the views of every entry in it.  swap() then
        makes it the arena the modules' parameters / buffers, phys() and shadow() point into - a pointer flip per entry, no copy."""
        if self._alt is not None:
            return
        flat2 = torch.empty_like(self.flat)
        flat_t2 = torch.empty_like(self.flat_t) if self.flat_t is not None else None
        self._alt = (flat2, flat_t2)
        for e in self.entries:
            e.v_log = self._view(self.flat, e)
            src2 = flat_t2 if flat_t2 is not None else flat2
            e.alt = (flat2[e.off:e.off + e.numel], src2[e.off:e.off + e.numel], self._view(flat2, e))

    def other(self):
        """(flat, flat_t) of the arena pair that is NOT visible"""
        return self._alt

    def swap(self):
        if self.with_grad or self._alt is None:
            raise S4FError('swap() needs enable_double() on a store without gradients (the teacher)')
        flat2, flat_t2 = self._alt
        self._alt = (self.flat, self.flat_t)
        self.flat, self.flat_t = flat2, flat_t2
        for e in self.entries:
            p2, s2, l2 = e.alt
            e.alt = (e.v_phys, e.v_shadow, e.v_log)
            e.v_phys, e.v_shadow, e.v_log = p2, s2, l2
            self._tensor(e).data = l2               # same tensor objects (ids, caches keyed by them, versions): only the storage moves

    def is_valid(self, device):
        if self.flat is None or self.flat.device != torch.device(device):
            return False
        for e in (self.entries[0], self.entries[-1]):
            t = self._tensor(e)
            if t.data_ptr() != self.flat.data_ptr() + 4 * e.off:
                return False
        return True

    def ensure(self, device, dtype_code):
        if not self.is_valid(device) or self.dtype != dtype_code:
            self.build(device, dtype_code)
        return self

    def ensure_grads(self):
        """re-attach .grad views (a foreign zero_grad(set_to_none=True) detaches them; None means zero)"""
        if not self.with_grad:
            return
        for e in self.entries:
            if not e.is_param:
                continue
            prm = e.module._parameters[e.attr]
            if not prm.requires_grad:
                continue
            gv = self._view(self.grad, e)
            if prm.grad is None or prm.grad.data_ptr() != gv.data_ptr():
                if prm.grad is None:
                    gv.zero_()
                else:
                    gv.copy_(prm.grad)
                prm.grad = gv

    # ------------------------------------------------------------------ access for kernels
    def entry(self, t):
        e = self.by_id.get(id(t))
        if e is None:
            raise S4FError('tensor is not managed by this ParamStore')
        return e

    def phys(self, t):
        """fp32 physical (contiguous) view of a parameter/buffer"""
        return self.entry(t).v_phys

    def grad_phys(self, t):
        return self.entry(t).v_grad

    def shadow(self, t):
        """operand-typed physical view (bf16 shadow arena in perf mode, the fp32 master in parity mode)"""
        return self.entry(t).v_shadow

    # ------------------------------------------------------------------ shadow maintenance
    def _version_sum(self):
        return sum(self._tensor(e)._version for e in self.entries if e.is_param)

    def _state_version_sum(self):
        """like _version_sum, but over EVERY arena entry - the BatchNorm running statistics as well: what a pending
        double-buffered teacher update must be keyed by (a buffer-only load_state_dict / reset_running_stats is a foreign write
        too).  Writes through raw pointers do not move a version counter: they announce themselves with mark_dirty()."""
        return sum(self._tensor(e)._version for e in self.entries) + self._dirty_marks

    def mark_dirty(self):
        self._shadow_dirty = True
        self._T_fresh = False
        self._dirty_marks += 1

    # ------------------------------------------------------------------ transposed operand shadows (bf16 mode)
    def _T_rebuild_table(self):
        items, total = [], 0
        for off in sorted(self._T_items):
            R, T, C = self._T_items[off]
            items.append((off, off, R, T, C, total))
            total += T * (R // 64) * (C // 64)
        dev = torch.tensor([v for it in items for v in it], dtype=torch.int64, device=self.flat.device)
        self._T_table = (dev, items)

    def _T_subtable(self, offs):
        """item table (device int64, host list) of the registered transposes at the given arena offsets; cached"""
        key = tuple(offs)
        cache = self.__dict__.setdefault('_T_sub', {})
        if key not in cache:
            items, total = [], 0
            for off in offs:
                R, T, C = self._T_items[off]
                items.append((off, off, R, T, C, total))
                total += T * (R // 64) * (C // 64)
            dev = torch.tensor([v for it in items for v in it], dtype=torch.int64, device=self.flat.device)
            cache[key] = (dev, items)
        return cache[key]

    def sync_T_range(self, a, b):
        """the transposed shadows of the weights inside arena range [a, b), on the current stream (the eager optimiser calls this
        right behind the SGD launch of the range: the one whole-arena transpose launch at the end of step() then only covers what
        was not stepped eagerly)"""
        if self.flat_t is None or not self._T_items or self.flat_T is None:
            return
        offs = [off for off in sorted(self._T_items) if a <= off < b]
        if not offs:
            return
        dev, items = self._T_subtable(offs)
        K.transpose_many(self.flat_t, self.flat_T, dev, items)
        self.__dict__.setdefault('_T_done', set()).update(offs)

    def sync_T(self, eager=False):
        """(re)make every registered transposed shadow from the bf16 shadow arena in ONE launch on the current stream.
        eager=True: called on the optimiser's stream right after the SGD kernels (ranges already transposed behind their eager
        update - sync_T_range - are left out); every stream of the next step forks from that stream, so no event is needed.
        Otherwise users wait for the recorded event."""
        if not self._T_items:
            self._T_fresh = True
            return
        done = self.__dict__.pop('_T_done', set()) if eager else set()
        self.__dict__.pop('_T_done', None)
        if done:
            rest = [off for off in sorted(self._T_items) if off not in done]
            if rest:
                dev, items = self._T_subtable(rest)
                K.transpose_many(self.flat_t, self.flat_T, dev, items)
        else:
            if self._T_table is None:
                self._T_rebuild_table()
            K.transpose_many(self.flat_t, self.flat_T, self._T_table[0], self._T_table[1])
        self._T_fresh = True
        if eager:
            self._T_event = None
        else:
            self._T_event = torch.cuda.Event()
            self._T_event.record()

    def shadow_T(self, t):
        """bf16 transposed shadow of a 2-D weight [R, C] -> [C, R], or of a 3x3 conv weight (physically [co][tap][ci])
        -> [ci][tap][co]: the B operand with which the input-gradient GEMM runs row-major x row-major."""
        if self.flat_t is None:
            raise S4FError('transposed shadows exist in bf16 mode only')
        e = self.entry(t)
        if e.off not in self._T_items:
            if e.cl:
                co, ci = e.shape[0], e.shape[1]
                dims = (co, 9, ci)
            elif len(e.shape) == 2:
                dims = (e.shape[0], 1, e.shape[1])
            else:
                raise S4FError(f'shadow_T: unsupported parameter shape {e.shape}')
            if dims[0] % 64 or dims[2] % 64:
                raise S4FError(f'shadow_T: dims {dims} are not multiples of 64')
            if self.flat_T is None:
                self.flat_T = torch.zeros(self.total, device=self.flat.device, dtype=torch.bfloat16)
            self._T_items[e.off] = dims
            self._T_table = None
            self.__dict__.pop('_T_sub', None)
            self._T_fresh = False
            e.v_T = self.flat_T[e.off:e.off + e.numel]
        if not self._T_fresh:
            self.sync_shadow()
            self.sync_T()
        elif self._T_event is not None:
            torch.cuda.current_stream().wait_event(self._T_event)
        return e.v_T

    def sync_shadow(self):
        """refresh the bf16 shadow if anything but our own fused kernels touched the masters"""
        if self.flat_t is None:
            return
        v = self._version_sum()
        if self._shadow_dirty or v != self._versions:
            K.cast(self.flat, self.flat_t, BF16)
            self._shadow_dirty = False
            self._versions = v
            self._T_fresh = False

    def node_done(self):
        """called by every autograd node of this replica at the end of its backward (hook for the data-parallel
        gradient reducer: when the last pending node has run, the gradient arena is final)"""
        self.grad_clean = False                      # a node has accumulated into the gradient arena (also nodes that never range_acquire)
        cb = getattr(self, 'on_node_done', None)
        if cb is not None:
            cb()

    def range_of(self, tensors):
        """[a, b) arena range spanned by the given parameters (padded to the arena alignment)"""
        es = [self.entry(t) for t in tensors]
        a = min(e.off for e in es)
        b = max(e.off + (e.numel + ALIGN - 1) // ALIGN * ALIGN for e in es)
        return a, b

    def range_done(self, a, b):
        """the gradients of arena range [a, b) are final for this step; the data-parallel reducer / the eager optimiser
        hook in here to start that bucket's all-reduce / update while backward continues.

        coalesce_min > 0 (set by dist.setup_data_parallel at N > 1): ranges are MERGED with their already-final neighbours of the
        same parameter group and handed on only when the merged span reaches coalesce_min elements - three encoder layers per
        bucket instead of one all-reduce (and one SGD launch) per layer: 41 -> <= 8 collectives per step.  A group whose last
        pending range has reported is flushed whatever its size (the heads: their collectives run under the backbone's
        backward); anything never reported at all is left to GradReducer.reduce_() / optimizer.step(), which cover every range
        not handed on."""
        cb = getattr(self, 'on_range_done', None)
        if cb is None:
            return
        cmin = getattr(self, 'coalesce_min', 0)
        if cmin <= 0:
            cb(a, b)
            return
        ep = getattr(self, 'step_epoch', 0)
        if getattr(self, '_co_epoch', None) != ep:
            self._co, self._co_epoch = [], ep           # spans left pending by the previous step were covered by reduce_() / step()
        ga, gb = self._flush_unit(a, b, cmin)
        # a span that ends where a group's parameters end also takes the group's tail (BatchNorm running statistics: no
        # gradients, zeros in the gradient arena) - so that the spans of adjacent head groups touch and merge instead of
        # leaving ten collectives (five heads + five 1,024-element tails at the end of the step) where one will do
        for rng in self.group_ranges.values():
            if b == rng['params'][1] and a >= rng['all'][0]:
                if os.environ.get('S4F_CHECK_FLUSH') and self.grad is not None and rng['all'][1] > b:
                    # debug: the tail (running statistics: no gradients) rides along in the all-reduce and must be zeros
                    assert float(self.grad[b:rng['all'][1]].abs().max()) == 0.0, 'gradient arena: non-zero tail slots in a flushed span'
                b = rng['all'][1]
                break
        # merge with pending neighbours inside the group
        merged = True
        while merged:
            merged = False
            for i, (pa, pb) in enumerate(self._co):
                if ga <= pa and pb <= gb and (pb == a or b == pa):
                    a, b = min(a, pa), max(b, pb)
                    del self._co[i]
                    merged = True
                    break
        if b - a >= cmin:
            cb(a, b)
        else:
            self._co.append((a, b))
        # round 4: a flush unit (_flush_unit: a parameter group, or the run of small adjacent head groups) none of whose ranges is
        # still pending (no node of this step will write into it any more) is handed on AT ONCE, whatever its size - the reference's DDP reducer fires a bucket as soon as it is ready
        # (mmseg/apis/train.py:129-138).  The decode head (3.5 M) and the auxiliary heads (9.5 M) never reach coalesce_min; parked
        # until reduce_() their all-reduce + SGD sat on the critical path behind the whole backward although their gradients are
        # final before the backbone's backward starts.
        pend = getattr(self, '_pend', None) or {}
        if GROUP_FLUSH and not any(ga <= pa and pb <= gb for (pa, pb) in pend):
            mine = sorted(sp for sp in self._co if ga <= sp[0] and sp[1] <= gb)
            if mine:
                self._co = [sp for sp in self._co if not (ga <= sp[0] and sp[1] <= gb)]
                for sa, sb in mine:
                    cb(sa, sb)

    def _flush_unit(self, a, b, cmin):
        """[ua, ub): the stretch of the arena inside which spans are merged and which is handed on as soon as none of its ranges
        is pending: a parameter group, or a run of ADJACENT groups that are each smaller than coalesce_min (the decode head and the
        four auxiliary heads, 13 M elements together: one all-reduce + one SGD for all of them instead of five)."""
        key = (cmin, len(self.group_ranges))
        if getattr(self, '_units_key', None) != key:
            units, run = [], None
            for rng in sorted(r['all'] for r in self.group_ranges.values()):
                small = rng[1] - rng[0] < cmin
                if small and run is not None and run[1] == rng[0]:
                    run[1] = rng[1]
                elif small:
                    run = [rng[0], rng[1]]
                    units.append(run)
                else:
                    run = None
                    units.append([rng[0], rng[1]])
            self._units, self._units_key = [tuple(u) for u in units], key
        for ua, ub in self._units:
            if ua <= a and b <= ub:
                return ua, ub
        return a, b

    # A range is FINAL when the last node that accumulates into it has run its backward.  A node announces itself in its
    # forward (range_acquire) and signs off at the end of its backward (range_release); the counts restart with every step
    # (step_epoch, bumped by the segmentor).  A layer used by two passes of one step (masked + plain student pass, gradient
    # accumulation) is therefore reported once, after the second backward.  A graph that is never run backward leaves its
    # count above zero: the range is then simply not reported early and falls to reduce_() / step(), which cover the rest.
    def range_acquire(self, rng):
        self.grad_clean = False                      # a backward pass will accumulate into the gradient arena
        ep = getattr(self, 'step_epoch', 0)
        if getattr(self, '_pend_epoch', None) != ep:
            self._pend, self._pend_epoch = {}, ep
        self._pend[rng] = self._pend.get(rng, 0) + 1

    def range_release(self, rng):
        pend = getattr(self, '_pend', None)
        if pend is None or rng not in pend:
            return
        pend[rng] -= 1
        if pend[rng] == 0:
            del pend[rng]
            self.range_done(*rng)

    def zero_grad(self):
        if self.grad is not None and not getattr(self, 'grad_clean', False):
            self.grad.zero_()

    def named_entries(self):
        return [(e.name, e) for e in self.entries]
