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

    def _launch(self, t):
        with self._comm_ctx(t):
            self._handles.append(self.issue(t))
        self.launches += 1

    def _range_done(self, a, b):
        if not collectives_active() or self._store is None or self._store.grad is None:
            return
        if any(a < db and da < b for da, db in self._done):
            raise S4FError(f'gradient range [{a}, {b}) was reported final twice in one step (a backward pass ran after the '
                           'range had been handed to the reducer: gradient accumulation over several forward_train calls '
                           'is not supported with an attached reducer)')
        self._launch(self._store.grad[a:b])
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
                self._launch(flat_grad[c:min(b, c + self.bucket)])
        self._done = []

    def wait(self):
        for h in self._handles:
            h.wait()
        self._handles = []
        if self._stream is not None:
            torch.cuda.current_stream().wait_stream(self._stream)

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
    None of it has run with a peer on hardware yet.
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
        if not reducer.stream_layout.get('ok'):
            import warnings
            bad = {k: v for k, v in reducer.stream_layout.get('pairs', {}).items() if v >= 1.5}
            warnings.warn('s4former_amd.dist: the first-use stream order did not separate these stream pairs onto different hardware '
                          f'queues (ratio of a pair of spin kernels to one: {bad or reducer.stream_layout}); the step is correct but '
                          'streams that share a queue serialise (S4F_STREAM_ORDER=0 skips the pre-touch)', RuntimeWarning)
    if os.environ.get('S4F_EAGER_SGD', '1') != '0':
        # parameter ranges are updated as soon as their (all-reduced) gradient is final, behind the rest of backward
        optimizer.attach_eager(model.student_store, reducer if collectives_active() else None, reducer.grad_scale())
    return reducer


class _Null:
    def __enter__(self):
        return self

    def __exit__(self, *a):
        return False
