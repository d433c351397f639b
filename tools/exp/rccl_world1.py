#!/usr/bin/env python
"""The N > 1 data path of bench.py through REAL RCCL on one GPU: a process group of ONE rank over backend 'nccl', with the
package's "is anything exchanged" switch forced on, so that every collective of the step (per-range gradient all-reduce
issued asynchronously on the communication stream during backward, SyncBN statistics exchange inside the head nodes, the
batched log-scalar all-reduce, the eager SGD behind the reducer's work handles) goes through ProcessGroupNCCL / RCCL's
enqueue path, its internal stream and its work objects.  A one-rank all-reduce moves no data across xGMI - this checks the
API usage (dtypes, contiguity, stream ordering of async work, handle.wait() inside backward), not bandwidth or scaling.

  python tools/exp/rccl_world1.py -- <bench.py arguments>"""
import datetime
import os
import runpy
import sys

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

import s4former_amd.dist as D          # noqa: E402
import s4former_amd.functional as F_   # noqa: E402

CALLS = dict(grad=0, bn=0, sizes=[])


def _reduce(self):
    if self.buf is not None:
        CALLS['bn'] += 1
        self.issue(self.buf)               # the real dist.all_reduce (world stays 1: the BN counts must)
    self.buf = None


if __name__ == '__main__':
    argv = [a for a in sys.argv[1:] if a != '--']
    os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
    os.environ.setdefault('MASTER_PORT', '29531')
    os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    torch.cuda.set_device(0)
    dist.init_process_group('nccl', rank=0, world_size=1, timeout=datetime.timedelta(seconds=120))
    only = os.environ.get('RCCL1_ONLY', '')          # 'grad' | 'bn': force only one kind of exchange (cost attribution)
    D.collectives_active = lambda: True
    issue0 = D.GradReducer.issue

    def _issue(t):
        CALLS['grad'] += 1
        CALLS['sizes'].append(int(t.numel()))
        return issue0(t)
    if only != 'bn':
        D.GradReducer.issue = staticmethod(_issue)
    else:
        class _Done:
            def wait(self):
                pass
        D.GradReducer.issue = staticmethod(lambda t: _Done())
    if only != 'grad':
        F_._Exchange.reduce = _reduce
    sys.argv = [os.path.join(ROOT, 'bench.py')] + argv
    try:
        runpy.run_path(sys.argv[0], run_name='__main__')
    finally:
        print(f'[rccl_world1] backend {dist.get_backend()}, collectives issued: gradient ranges {CALLS["grad"]}, '
              f'SyncBN exchanges {CALLS["bn"]}', flush=True)
        print('[rccl_world1] elements of the last 32 gradient collectives:', CALLS['sizes'][-32:], flush=True)
        dist.destroy_process_group()
