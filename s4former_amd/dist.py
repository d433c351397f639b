"""Data-parallel plumbing: one process per GPU over torch.distributed (backend 'nccl' = RCCL over xGMI).

The only exchange on the data path is the mean of the student gradients (reference: MMDistributedDataParallel,
mmseg/apis/train.py:129-138).  Because every gradient of a replica lives in ONE flat fp32 arena, the reducer
all-reduces contiguous arena ranges (large buckets suit the point-to-point xGMI links) on a side stream as soon
as the last autograd node of the step has run, and the 1/world scaling is folded into the fused SGD kernel.
SyncBN statistics (2*C floats per BN call) are all-reduced inside the head nodes (functional.py); the log
scalars are reduced in one batched all-reduce (encoder_decoder.BaseSegmentor._parse_losses)."""
import os

import torch
import torch.distributed as dist


def init_distributed(backend=None):
    """reads RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* set by torch.distributed.run; returns (rank, local, world)"""
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local = int(os.environ.get('LOCAL_RANK', '0'))
    if world > 1 and not dist.is_initialized():
        if backend is None:
            backend = os.environ.get('S4F_DIST_BACKEND') or ('nccl' if torch.cuda.is_available() else 'gloo')
        if torch.cuda.is_available():
            torch.cuda.set_device(local % max(1, torch.cuda.device_count()))
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', '29500')
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, local, world


def world_size():
    return dist.get_world_size() if (dist.is_available() and dist.is_initialized()) else 1


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

    def _launch(self, t):
        with self._comm_ctx(t):
            if _rehearsal() is not None:
                self._handles.append(_StandInWork(t))
            else:
                self._handles.append(dist.all_reduce(t, async_op=True))

    def _range_done(self, a, b):
        if (world_size() == 1 and _rehearsal() is None) or self._store is None or self._store.grad is None:
            return
        self._launch(self._store.grad[a:b])
        self._done.append((a, b))

    def reduce_(self, flat_grad):
        """launch the all-reduces of every range not yet reduced in this step; call wait() before the optimiser step"""
        if world_size() == 1 and _rehearsal() is None:
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


def _rehearsal():
    """one-GPU rehearsal of the N > 1 control flow (bench.py, S4F_STREAM_LAYOUT=test): the stream standing in for RCCL's"""
    if world_size() > 1:
        return None
    from . import functional as F_
    return F_.STANDIN


class _StandInWork:
    """what an asynchronous all-reduce does to the streams, without peers: RCCL's stream waits for the issuing stream, passes
    over the buffer twice (a ring all-reduce reads and writes it about that often), and wait() makes the caller's stream
    wait for it"""

    def __init__(self, t):
        st = _rehearsal()
        st.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(st):
            t.mul_(1.0)
            t.mul_(1.0)
        self._st = st

    def wait(self):
        torch.cuda.current_stream().wait_stream(self._st)


class _Null:
    def __enter__(self):
        return self

    def __exit__(self, *a):
        return False
