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
            backend = 'nccl' if torch.cuda.is_available() else 'gloo'
        if backend == 'nccl':
            torch.cuda.set_device(local)
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', '29500')
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, local, world


def world_size():
    return dist.get_world_size() if (dist.is_available() and dist.is_initialized()) else 1


class GradReducer:
    """Sum-all-reduce of the gradient arena in `bucket_mb` chunks (the division by world is applied by the optimiser:
    S4FSGD.step(grad_scale=1/world)).  Works on any flat tensor, so it is covered by gloo tests on CPU."""

    def __init__(self, bucket_mb=128, side_stream=True):
        self.bucket = int(bucket_mb * 1024 * 1024 // 4)
        self.side_stream = side_stream
        self._stream = None
        self._handles = []

    def broadcast_(self, flat, src=0):
        if world_size() > 1:
            dist.broadcast(flat, src=src)

    def reduce_(self, flat_grad):
        """launch the all-reduces; call wait() before the optimiser step"""
        if world_size() == 1:
            return
        n = flat_grad.numel()
        use_side = self.side_stream and flat_grad.is_cuda
        if use_side:
            if self._stream is None:
                self._stream = torch.cuda.Stream()
            self._stream.wait_stream(torch.cuda.current_stream())
            ctx = torch.cuda.stream(self._stream)
        else:
            ctx = _Null()
        with ctx:
            for a in range(0, n, self.bucket):
                self._handles.append(dist.all_reduce(flat_grad[a:min(n, a + self.bucket)], async_op=True))

    def wait(self):
        for h in self._handles:
            h.wait()
        self._handles = []
        if self._stream is not None:
            torch.cuda.current_stream().wait_stream(self._stream)

    def grad_scale(self):
        return 1.0 / world_size()


class _Null:
    def __enter__(self):
        return self

    def __exit__(self, *a):
        return False
