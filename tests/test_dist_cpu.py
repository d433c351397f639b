"""world_size-2 / 4 / 8 gloo tests (CPU; SURVEY App. C asks for 2 - 8 processes) of the N>1 path: bucketed gradient all-reduce + 1/world scaling, batched log-scalar
reduction of _parse_losses, SyncBN statistics = statistics of the concatenated global batch, identical replicas."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _plain(v):
    """tensors by VALUE through the queue (torch shares tensor storages through file descriptors served by the sending process:
    a worker that has exited by the time the parent unpickles leaves nothing to connect to - seen at world 8)"""
    if isinstance(v, torch.Tensor):
        return ('__tensor__', v.detach().cpu().numpy().copy())
    if isinstance(v, (tuple, list)):
        return type(v)(_plain(x) for x in v)
    if isinstance(v, dict):
        return {k: _plain(x) for k, x in v.items()}
    return v


def _unplain(v):
    if isinstance(v, tuple) and len(v) == 2 and isinstance(v[0], str) and v[0] == '__tensor__':
        return torch.from_numpy(v[1])
    if isinstance(v, (tuple, list)):
        return type(v)(_unplain(x) for x in v)
    if isinstance(v, dict):
        return {k: _unplain(x) for k, x in v.items()}
    return v


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    from s4former_amd.dist import GradReducer, init_distributed, world_size
    from s4former_amd.encoder_decoder import BaseSegmentor
    r, l, w = init_distributed(backend='gloo')
    assert (r, w) == (rank, world) and world_size() == world
    res = {}
    # 1. gradient arena: bucketed sum all-reduce, mean via grad_scale
    g = torch.Generator().manual_seed(100 + rank)
    grad = torch.randn(300_001, generator=g)
    red = GradReducer(bucket_mb=0.25, side_stream=False)      # 65536-element buckets -> 5 buckets
    red.reduce_(grad)
    red.wait()
    res['grad'] = (grad * red.grad_scale()).clone()
    # 1b. overlapped buckets: ranges reported during "backward", the rest in reduce_()
    class FakeStore:
        pass
    st = FakeStore()
    st.grad = torch.randn(300_001, generator=torch.Generator().manual_seed(200 + rank))
    red2 = GradReducer(bucket_mb=0.25, side_stream=False).attach(st)
    st.on_range_done(200_000, 260_000)
    st.on_range_done(100_000, 200_000)
    red2.reduce_(st.grad)
    red2.wait()
    res['grad2'] = (st.grad * red2.grad_scale()).clone()
    # 1c. coalesced buckets (ParamStore.range_done with coalesce_min): layer ranges reported one by one in backward order are
    # handed to the reducer three at a time; what stays below the size is covered by reduce_()
    from s4former_amd.params import ParamStore
    ps = ParamStore.__new__(ParamStore)
    ps.group_ranges = {'backbone': dict(params=(0, 240_000), all=(0, 240_000)), 'head': dict(params=(240_000, 300_032), all=(240_000, 300_032))}
    ps.grad = torch.randn(300_032, generator=torch.Generator().manual_seed(300 + rank))
    ps.coalesce_min, ps.step_epoch = 60_000, 1
    red3 = GradReducer(bucket_mb=0.25, side_stream=False).attach(ps)
    launched = []
    issue0 = GradReducer.issue
    spans = []                                         # (offset, length) of every span handed to the reducer, in order
    red3.issue = lambda t: (launched.append(t.numel()), spans.append((t.storage_offset(), t.numel())), issue0(t))[2]
    # forward: every node announces its range (heads last, as in the model); backward: the heads sign off first
    layer_rngs = [(k * 20_000, (k + 1) * 20_000) for k in range(12)]
    head_rngs = [(240_000, 250_000), (250_000, 300_032)]                 # "decode head", "auxiliary heads": both far below the size
    for rng in layer_rngs + head_rngs:
        ps.range_acquire(rng)
    ps.range_release(head_rngs[0])                     # first head final: its group still has a pending range -> parked
    parked = list(launched)
    ps.range_release(head_rngs[1])                     # group complete: handed on although 60,032 < ... regardless of the size
    heads_done = list(launched)
    for rng in reversed(layer_rngs):                    # twelve "layers" of 20,000 elements, last first
        ps.range_release(rng)
    early = list(launched)
    red3.reduce_(ps.grad)
    red3.wait()
    res['grad3'] = (ps.grad * red3.grad_scale()).clone()
    res['buckets3'] = (early, list(launched), parked, heads_done)
    res['spans3'] = list(spans)
    # 2. parameters broadcast from rank 0
    p = torch.full((1000,), float(rank))
    red.broadcast_(p, src=0)
    res['bcast'] = float(p.abs().max())
    # 3. batched log-scalar reduction
    seg = BaseSegmentor()
    losses = {'decode.loss_ce': torch.tensor(1.0 + rank), 'aux_0.loss_ce': torch.tensor(0.5 * (rank + 1)), 'mask_ratio': torch.tensor(0.25 * rank)}
    loss, log_vars = seg._parse_losses(losses)
    res['loss_local'] = float(loss)
    res['log_vars'] = dict(log_vars)
    # 4. SyncBN statistics: all-reduced (sum, sumsq) == statistics of the concatenated batch
    gx = torch.Generator().manual_seed(7 + rank)
    x = torch.randn(64, 8, generator=gx) * (1 + rank) + rank
    sums = torch.cat([x.sum(0), (x * x).sum(0)])
    dist.all_reduce(sums)
    n = 64 * world
    mean = sums[:8] / n
    var = sums[8:] / n - mean * mean
    res['bn'] = (mean, var, x)
    q.put((rank, _plain(res)))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize('world', [2, 4, 8])
def test_world_gloo(world):
    port = _free_port()
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p_ in procs:
        p_.start()
    out = {}
    for _ in range(world):
        r, res = q.get(timeout=240)
        out[r] = _unplain(res)
    for p_ in procs:
        p_.join(timeout=60)
        assert p_.exitcode == 0

    def mean_of(seed0, n):
        return sum(torch.randn(n, generator=torch.Generator().manual_seed(seed0 + r)) for r in range(world)) / world
    for key, seed0, n in (('grad', 100, 300_001), ('grad2', 200, 300_001), ('grad3', 300, 300_032)):
        assert torch.allclose(out[0][key], mean_of(seed0, n), atol=2e-6), key
        for r in range(1, world):
            assert torch.equal(out[0][key], out[r][key]), (key, r)       # replicas bit-identical
    early, all_, parked, heads_done = out[0]['buckets3']
    for r in range(1, world):
        # every rank hands the SAME spans to the reducer in the SAME order (a collective per span: a different order would pair
        # the all-reduces of different spans, or hang)
        assert out[r]['spans3'] == out[0]['spans3'], (r, out[r]['spans3'], out[0]['spans3'])
        assert out[r]['buckets3'] == out[0]['buckets3']
    # round 4: a head group is handed to the reducer as soon as its LAST range is final - before the first backbone bucket -
    # whatever its size; 12 layers -> 4 buckets of 3 during "backward"
    assert parked == [], parked
    assert sum(heads_done) == 60_032 and len(heads_done) <= 2, heads_done
    assert early[:len(heads_done)] == heads_done and early[len(heads_done):] == [60_000] * 4, early
    assert sum(all_) == 300_032 and len(all_) == len(early), all_   # every element exactly once, nothing left for reduce_()
    assert all(out[r]['bcast'] == 0.0 for r in range(world))
    # local loss stays local (it is what backward runs on); logged values are the rank mean
    mr = (world - 1) / 2.0                              # mean rank
    for r in range(world):
        assert out[r]['loss_local'] == 1.5 * (r + 1)
        lv = out[r]['log_vars']
        assert abs(lv['decode.loss_ce'] - (1.0 + mr)) < 1e-6 and abs(lv['aux_0.loss_ce'] - 0.5 * (mr + 1)) < 1e-6
        assert abs(lv['mask_ratio'] - 0.25 * mr) < 1e-6 and abs(lv['loss'] - 1.5 * (mr + 1)) < 1e-6
    xc = torch.cat([out[r]['bn'][2] for r in range(world)])
    assert torch.allclose(out[0]['bn'][0], xc.mean(0), atol=1e-5)
    assert torch.allclose(out[0]['bn'][1], xc.var(0, unbiased=False), atol=1e-4 * world)
    for r in range(1, world):
        assert torch.equal(out[0]['bn'][0], out[r]['bn'][0])


def _diverge_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    from s4former_amd._lib import S4FError
    from s4former_amd.dist import GradReducer, init_distributed
    init_distributed(backend='gloo', timeout_s=60)
    g = torch.ones(4096)
    res = {}
    # (1) the same span on every rank: checked, reduced, one checked step used up
    red = GradReducer(side_stream=False)
    assert red.check_steps == 3
    red._launch(g[0:1024], (0, 1024))
    red.wait()
    res['same'] = (float(g[0]), red.check_steps)
    # (2) another span on the last rank: every rank gets the error BEFORE the collective is issued (no hang, no wrong sum)
    a, b = (1024, 2048) if rank < world - 1 else (2048, 3072)
    try:
        red._launch(g[a:b], (a, b))
        res['diverged'] = 'no error'
    except S4FError as e:
        res['diverged'] = ('DIFFERENT spans' in str(e), str([0, a, b, 1]) in str(e))
    res['untouched'] = float(g[1024:3072].sum())
    # (3) S4F_CHECK_FLUSH = 0 steps: nothing is exchanged beside the collective itself
    red0 = GradReducer(side_stream=False)
    red0.check_steps = 0
    red0._launch(g[3072:4096], (3072, 4096))
    red0.wait()
    res['unchecked'] = float(g[3072])
    q.put((rank, res))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize('world', [2, 4])
def test_a_divergent_flush_order_raises_on_every_rank(world):
    """round 6, GradReducer.check_steps (reference: the fixed bucket order of the DDP wrap, mmseg/apis/train.py:129-138): in the
    first steps of a multi-rank run the ranks exchange the span they are about to all-reduce; a rank with ANOTHER span makes every
    rank raise an S4FError that names the spans - before the mismatched collective is issued"""
    port = _free_port()
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    procs = [ctx.Process(target=_diverge_worker, args=(r, world, port, q)) for r in range(world)]
    for p_ in procs:
        p_.start()
    out = {}
    for _ in range(world):
        r, res = q.get(timeout=120)
        out[r] = res
    for p_ in procs:
        p_.join(timeout=60)
        assert p_.exitcode == 0
    for r in range(world):
        assert out[r]['same'] == (float(world), 2), out[r]
        assert out[r]['diverged'] == (True, True), out[r]
        assert out[r]['untouched'] == 2048.0, out[r]                # the mismatched spans were never reduced
        assert out[r]['unchecked'] == float(world)
