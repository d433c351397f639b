"""Data-parallel plumbing: one process per GPU over torch.distributed (backend 'nccl' = RCCL over xGMI).

The only exchange on the data path is the mean of the student gradients (reference: MMDistributedDataParallel,
mmseg/apis/train.py:129-138).  Because every gradient of a replica lives in ONE flat fp32 arena, the reducer
all-reduces contiguous arena ranges (large buckets suit the point-to-point xGMI links) on a side stream as soon
as the last autograd node of the step has run, and the 1/world scaling is folded into the fused SGD kernel.
SyncBN statistics (2*C floats per BN call) are all-reduced inside the head nodes (functional.py); the log
scalars are reduced in one batched all-reduce (encoder_decoder.BaseSegmentor._parse_losses)."""
import datetime
import os

import torch
import torch.distributed as dist

from ._lib import S4FError


def init_distributed(backend=None, timeout_s=None):
    """reads RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* set by torch.distributed.run; returns (rank, local, world).
    timeout_s (or S4F_DIST_TIMEOUT_S, default 600): a collective that a peer never joins RAISES after that long instead of
    hanging for the backends' 10 - 30 minute defaults."""
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local = int(os.environ.get('LOCAL_RANK', '0'))
    if world > 1 and not dist.is_initialized():
        if backend is None:
            backend = os.environ.get('S4F_DIST_BACKEND') or ('nccl' if torch.cuda.is_available() else 'gloo')
        if timeout_s is None:
            timeout_s = float(os.environ.get('S4F_DIST_TIMEOUT_S', '600'))
        if torch.cuda.is_available():
            torch.cuda.set_device(local % max(1, torch.cuda.device_count()))     # RCCL binds its communicator to the current device
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', '29500')
        os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')    # dmabuf IPC: what RCCL's intra-node transport needs on this driver
        dist.init_process_group(backend=backend, rank=rank, world_size=world, timeout=datetime.timedelta(seconds=timeout_s))
    return rank, local, world


def world_size():
    return dist.get_world_size() if (dist.is_available() and dist.is_initialized()) else 1


def collectives_active():
    """True when the data path exchanges anything (N > 1).  tools/exp/rehearsal.py overrides this and the two issue()
    hooks (GradReducer.issue, functional._Exchange.issue) to rehearse the N > 1 control flow on one GPU."""
    return world_size() > 1


class GradReducer:
    """Sum-all-reduce of the gradient arena (the division by world is applied by the optimiser:
    S4FSGD.step(grad_scale=1/world)).

    attach(store): overlap with backward.  Every encoder layer reports its arena range as soon as its backward node
    has run (ParamStore.range_done); that bucket (7.1 M fp32 = 28 MB, one contiguous range) is all-reduced at once on
    a communication stream while the earlier layers are still computing.  reduce_() then covers what is left (heads,
    patch embedding) in `bucket_mb` chunks.  Backward visits the layers in the same order on every rank, so the
    collectives line up.  Works on any flat tensor / backend, so it is covered by gloo tests on CPU."""

    def __init__(self, bucket_mb=128, side_stream=True):
        self.bucket = int(bucket_mb * 1024 * 1024 // 4)
        self.side_stream = side_stream
        self._stream = None
        self._handles = []
        self._done = []          # ranges already launched in this step
        self._store = None
        self.launches = 0            # collectives issued so far (bench.py reports the per-step count)
        # Round 6 - the first steps of a multi-rank run check themselves (S4F_CHECK_FLUSH = number of checked steps, default 3,
        # 0 = off): before every gradient collective the ranks exchange (sequence number in the step, span [a, b), "the span's
        # tail slots are zeros") and a rank that is about to all-reduce ANOTHER span than its peers raises an S4FError that names
        # both - instead of the hang, or the silently wrong gradient, a divergent flush order ends in.  Costs one 4-word
        # all-reduce + a host sync per collective in those steps only.
        self.check_steps = int(os.environ.get('S4F_CHECK_FLUSH', '3') or 0)
        self._seq = 0
        self.timing = False          # record an event pair per collective (latency_summary)
        self._timed = []
        self._lat = []

    def attach(self, store):
        self._store = store
        store.on_range_done = self._range_done
        return self

    def broadcast_(self, flat, src=0):
        if world_size() > 1:
            dist.broadcast(flat, src=src)

    def _comm_ctx(self, t):
        use_side = self.side_stream and t.is_cuda
        if not use_side:
            return _Null()
        if self._stream is None:
            self._stream = torch.cuda.Stream()
        cur = torch.cuda.current_stream()
        if cur != self._stream:
            self._stream.wait_stream(cur)
        from .functional import extra_streams
        for st in extra_streams():
            if st != self._stream:                    # (the communication stream may BE the weight-gradient stream)
                self._stream.wait_stream(st)          # weight-gradient kernels / heads run on their own streams
        return torch.cuda.stream(self._stream)

    @staticmethod
    def issue(t):
        """asynchronous sum-all-reduce of one arena range; returns the work handle"""
        return dist.all_reduce(t, async_op=True)

    def _verify_span(self, t, a, b):
        """checked steps: every rank must be about to reduce the SAME span as collective number `_seq` of this step"""
        world, rank = world_size(), dist.get_rank()
        tail_ok = 1
        st = self._store
        if st is not None and st.grad is not None and t.data_ptr() == st.grad[a:b].data_ptr():
            for rng in (getattr(st, 'group_ranges', None) or {}).values():          # the BatchNorm statistics slots a flushed span carries have no gradient
                lo, hi = max(a, rng['params'][1]), min(b, rng['all'][1])
                if lo < hi and float(st.grad[lo:hi].abs().max()) != 0.0:
                    tail_ok = 0
        tab = torch.zeros(world, 4, dtype=torch.int64, device=t.device)
        tab[rank] = torch.tensor([self._seq, a, b, tail_ok], dtype=torch.int64, device=t.device)
        dist.all_reduce(tab)
        rows = tab.cpu().tolist()
        if any(r[:3] != rows[0][:3] for r in rows):
            raise S4FError('data-parallel gradient exchange: the ranks are about to all-reduce DIFFERENT spans of the gradient arena '
                           f'(collective {self._seq} of this step; per rank (seq, a, b, tail zeros): {rows}).  The backward passes of the '
                           'replicas reported their final ranges in different orders - every rank must run the same graph '
                           '(same tags, same batch composition); S4F_GROUP_FLUSH=0 / S4F_BUCKET_MIN_ELEMS=0 select simpler schedules')
        if not all(r[3] for r in rows):
            raise S4FError(f'data-parallel gradient exchange: span [{a}, {b}) carries non-zero values in slots that hold no gradient '
                           f'(BatchNorm statistics) on ranks {[i for i, r in enumerate(rows) if not r[3]]}')

    def _launch(self, t, span=None):
        with self._comm_ctx(t):
            if self.check_steps > 0 and span is not None and collectives_active() and dist.is_initialized():
                self._verify_span(t, *span)
            e0 = None
            if self.timing and t.is_cuda:
                e0 = torch.cuda.Event(enable_timing=True)
                e0.record()
            self._handles.append(self.issue(t))
            if e0 is not None:
                self._timed.append((t.numel(), e0, self._handles[-1]))
        self._seq += 1
        self.launches += 1

    def latency_summary(self):
        """per-collective latency of the steps recorded while `timing` was on: enqueue on the communication stream -> the stream
        has passed the collective (event pairs; includes the wait for the range's producers), after a device synchronise"""
        if not self._lat:
            return None
        torch.cuda.synchronize()
        ms = [(n, e0.elapsed_time(e1)) for n, e0, e1 in self._lat]
        self._lat = []
        return dict(collectives=len(ms), mean_ms=round(sum(m for _, m in ms) / len(ms), 3), max_ms=round(max(m for _, m in ms), 3),
                    mbytes=[round(4 * n / 1e6, 1) for n, _ in ms][:16], ms=[round(m, 3) for _, m in ms][:16])

    def _range_done(self, a, b):
        if not collectives_active() or self._store is None or self._store.grad is None:
            return
        if any(a < db and da < b for da, db in self._done):
            raise S4FError(f'gradient range [{a}, {b}) was reported final twice in one step (a backward pass ran after the '
                           'range had been handed to the reducer: gradient accumulation over several forward_train calls '
                           'is not supported with an attached reducer)')
        self._launch(self._store.grad[a:b], (a, b))
        self._done.append((a, b))

    def reduce_(self, flat_grad):
        """launch the all-reduces of every range not yet reduced in this step; call wait() before the optimiser step"""
        if not collectives_active():
            self._done = []
            return
        n = flat_grad.numel()
        pos = 0
        gaps = []
        for a, b in sorted(self._done):
            if a > pos:
                gaps.append((pos, a))
            pos = max(pos, b)
        if pos < n:
            gaps.append((pos, n))
        for a, b in gaps:
            for c in range(a, b, self.bucket):
                self._launch(flat_grad[c:min(b, c + self.bucket)], (c, min(b, c + self.bucket)))
        self._done = []

    def wait(self):
        if self._timed:
            # (timing on) the end of every recorded collective on the communication stream, in issue order
            ctx = torch.cuda.stream(self._stream) if self._stream is not None else _Null()
            with ctx:
                for n, e0, h in self._timed:
                    h.wait()
                    e1 = torch.cuda.Event(enable_timing=True)
                    e1.record()
                    self._lat.append((n, e0, e1))
            self._timed = []
        for h in self._handles:
            h.wait()
        self._handles = []
        if self._stream is not None:
            torch.cuda.current_stream().wait_stream(self._stream)
        # end of a step's exchange: the sequence restarts, one checked step less
        self._seq = 0
        if self.check_steps > 0 and collectives_active():
            self.check_steps -= 1

    def grad_scale(self):
        return 1.0 / world_size()


def setup_data_parallel(model, optimizer, device, reducer=None):
    """What a training loop does ONCE after `model.to(device)` (bench.py and the world-2 tests share it):
    build the arenas, make the replicas identical (broadcast of rank 0's student and teacher arenas), hook the gradient
    reducer into the per-range `range_done` notifications and, unless S4F_EAGER_SGD=0, the optimiser's eager per-range step.

    N > 1 defaults: all-reduce of a span of the gradient arena during backward as soon as it is final (three encoder layers per
    bucket; a head group as soon as its last range has reported: ParamStore.range_done), the SGD of a span behind its
    all-reduce, both on a communication stream of the reducer's own, and the empirical first-use order of the package's streams
    (S4F_STREAM_ORDER, below).  Opt-in: the heads advancing in lockstep (S4F_AUX_LOCKSTEP=1 / S4F_DECODE_LOCKSTEP=1: one SyncBN
    exchange per head layer for all of them; slower through a one-rank RCCL group on the round-4 tree, unmeasured with peers).
    None of it has run with a peer on hardware yet - which is why the first steps check themselves
    (GradReducer.check_steps) and why a training loop may call autotune_schedule() during warm-up (bench.py does).
    Returns the reducer; per step:  backward -> join_side_streams() -> reducer.reduce_(store.grad) -> reducer.wait() ->
    optimizer.step(grad_scale=reducer.grad_scale())."""
    reducer = reducer if reducer is not None else GradReducer()
    model.ensure_engine(device)
    reducer.broadcast_(model.student_store.flat)
    if model.teacher_store is not None:
        reducer.broadcast_(model.teacher_store.flat)
    model.student_store.mark_dirty()
    if model.teacher_store is not None:
        model.teacher_store.mark_dirty()
    reducer.attach(model.student_store)
    if collectives_active():
        # three encoder layers (3 x 7.09 M fp32 = 85 MB) per gradient bucket: 4 all-reduces for the backbone during backward
        # + the rest (heads, patch embedding) in 128 MB buckets at the end, instead of one collective per layer and head (41)
        model.student_store.coalesce_min = int(os.environ.get('S4F_BUCKET_MIN_ELEMS', str(20_000_000)))
    if collectives_active() and os.environ.get('S4F_STREAM_ORDER', '1') != '0' and torch.device(device).type == 'cuda':
        # First-use order of the package's streams (the runtime binds a stream to one of its four hardware queues at its first
        # use; two streams on one queue serialise): RCCL's stream (used by the broadcasts above), decode, aux, two throw-away
        # streams, weight gradients.  An EMPIRICAL setting, kept for what it measures through a one-rank RCCL group
        # (tools/exp/rccl_ab3.sh): 31.7 -> 29.9 ms per step on one box (28.7 ms without a process group; round 2: 33.95 -> 32.58).
        # Round 3 wrapped it in a model of the queue pattern with a self-check that never passed; round 4 first deleted both and
        # lost 2 ms at N > 1, then restored the order alone: no model, no claim about who shares a queue, and the reducer keeps a
        # communication stream of its own.
        from .functional import check_stream_layout, pretouch_streams
        pretouch_streams(device, ['decode', 'aux', 'burn', 'burn', 'side'])
        # Round 5: what the order is FOR is measured, once, right here (a few hundred microseconds): do the chain's stream, the two
        # head streams, the weight-gradient stream and the reducer's stream run concurrently pair by pair?  The result rides on
        # the reducer (bench.py prints it, tests/test_zz_dist_gpu.py reads it); a runtime on which the incantation has stopped
        # working shows up as a warning and a slower - never a wrong - step.
        reducer.stream_layout = check_stream_layout(device, extra=[('comm', getattr(reducer, '_stream', None))])
        if 'error' in reducer.stream_layout:
            import warnings
            warnings.warn(f"s4former_amd.dist: the stream-layout check could not run ({reducer.stream_layout['error']}); nothing is "
                          'known about which streams share a hardware queue', RuntimeWarning)
        elif not reducer.stream_layout.get('ok'):
            import warnings
            info = set(reducer.stream_layout.get('informational', ()))
            bad = {k: v for k, v in reducer.stream_layout.get('pairs', {}).items() if v >= 1.5 and k not in info}
            warnings.warn('s4former_amd.dist: the first-use stream order did not separate these stream pairs onto different hardware '
                          f'queues (ratio of a pair of spin kernels to one: {bad or reducer.stream_layout}); the step is correct but '
                          'streams that share a queue serialise (S4F_STREAM_ORDER=0 skips the pre-touch)', RuntimeWarning)
    if os.environ.get('S4F_EAGER_SGD', '1') != '0':
        # parameter ranges are updated as soon as their (all-reduced) gradient is final, behind the rest of backward
        optimizer.attach_eager(model.student_store, reducer if collectives_active() else None, reducer.grad_scale())
    return reducer


def autotune_schedule(model, reducer, run_step, steps=10):
    """Round 6: the N > 1 schedule knobs whose defaults rest on ONE-rank evidence are MEASURED on the job's own ranks during
    warm-up, and the fastest setting is kept (reference: the DDP wrap of mmseg/apis/train.py:129-138 has one fixed schedule):

      * head lockstep: off / auxiliary heads / auxiliary + decode head (S4F_AUX_LOCKSTEP, S4F_DECODE_LOCKSTEP: 32 -> 20 -> 12 SyncBN
        exchanges per step against the overlap the heads lose with each other),
      * gradient bucket size: ParamStore.coalesce_min and half of it (fewer, larger collectives against an earlier start).

    run_step(): one complete training step (every rank calls it the same number of times).  Each candidate runs one untimed +
    `steps` timed steps between barriers; the ranks agree on the MAX time per candidate (one all-reduce), so every rank keeps the
    same setting.  A knob the user has set in the environment is left alone; S4F_AUTOTUNE_SCHEDULE=0 skips everything.  Every
    setting this can pick is parity-tested at world 2 (tests/test_zz_dist_gpu.py).  Returns (and keeps on reducer.schedule) the
    record {candidate: ms per step, 'chosen': {...}} or None at N = 1."""
    import time
    if not collectives_active() or os.environ.get('S4F_AUTOTUNE_SCHEDULE', '1') == '0':
        return None
    store = model.student_store
    dev = store.flat.device

    def timed():
        run_step()                                   # the first step of a setting pays its one-off allocations / cache misses
        if dev.type == 'cuda':
            torch.cuda.synchronize()
        dist.barrier()
        t0 = time.perf_counter()
        for _ in range(steps):
            run_step()
        if dev.type == 'cuda':
            torch.cuda.synchronize()
        t = torch.tensor([time.perf_counter() - t0], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return 1e3 * float(t) / steps

    rec, chosen = {}, {}
    if 'S4F_AUX_LOCKSTEP' not in os.environ and 'S4F_DECODE_LOCKSTEP' not in os.environ:
        cands = {'lockstep=off': ('0', '0'), 'lockstep=aux': ('1', '0'), 'lockstep=aux+decode': ('1', '1')}
        try:
            for name, (a, d) in cands.items():
                os.environ['S4F_AUX_LOCKSTEP'], os.environ['S4F_DECODE_LOCKSTEP'] = a, d
                rec[name] = round(timed(), 3)
            best = min(cands, key=lambda k: rec[k])
        except Exception:
            os.environ.pop('S4F_AUX_LOCKSTEP', None)
            os.environ.pop('S4F_DECODE_LOCKSTEP', None)
            raise
        os.environ['S4F_AUX_LOCKSTEP'], os.environ['S4F_DECODE_LOCKSTEP'] = cands[best]
        chosen['lockstep'] = best.split('=')[1]
    if 'S4F_BUCKET_MIN_ELEMS' not in os.environ and getattr(store, 'coalesce_min', 0) > 1:
        c0 = store.coalesce_min
        cands = {f'bucket_min_elems={c0}': c0, f'bucket_min_elems={c0 // 2}': c0 // 2}
        for name, c in cands.items():
            store.coalesce_min = c
            rec[name] = round(timed(), 3)
        best = min(cands, key=lambda k: rec[k])
        store.coalesce_min = cands[best]
        chosen['bucket_min_elems'] = cands[best]
    rec['chosen'] = chosen
    reducer.schedule = rec
    return rec


class _Null:
    def __enter__(self):
        return self

    def __exit__(self, *a):
        return False
