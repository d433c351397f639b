#!/usr/bin/env python
"""One-GPU rehearsal of the N > 1 control flow of bench.py (round-1 experiment): every collective of the data
path is replaced by a stand-in kernel on a stream standing in for RCCL's, so that the stream / hardware-queue effects of
the exchanges can be timed without peers.  Test scaffolding: it lives here, not in the package - it overrides the three
hooks the package exposes (dist.collectives_active, dist.GradReducer.issue, functional._Exchange.issue).

  python tools/exp/rehearsal.py [--layout] -- <bench.py arguments>        (S4F_AUX_LOCKSTEP=1 etc. as for a real N > 1 run)"""
import os
import runpy
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

import s4former_amd.dist as D          # noqa: E402
import s4former_amd.functional as F_   # noqa: E402

STANDIN = None


def _standin():
    global STANDIN
    if STANDIN is None:
        STANDIN = torch.cuda.Stream()
        with torch.cuda.stream(STANDIN):      # first use: the stand-in takes the hardware queue RCCL's stream would
            torch.empty(1 << 20, device='cuda').fill_(1.0)
    return STANDIN


class _Work:
    """what an asynchronous all-reduce does to the streams, without peers: RCCL's stream waits for the issuing stream,
    passes over the buffer twice, and wait() makes the caller's stream wait for it"""

    def __init__(self, t):
        st = _standin()
        st.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(st):
            t.mul_(1.0)
            t.mul_(1.0)

    def wait(self):
        torch.cuda.current_stream().wait_stream(_standin())


def _exchange(buf):
    cur = torch.cuda.current_stream()
    st = _standin()
    st.wait_stream(cur)
    with torch.cuda.stream(st):
        buf.add_(0.0)
    cur.wait_stream(st)


def _reduce(self):
    if self.buf is not None:
        _exchange(self.buf)
    self.buf = None


if __name__ == '__main__':
    argv = sys.argv[1:]
    if '--layout' in argv:
        argv.remove('--layout')
    if '--' in argv:
        argv.remove('--')
    D.collectives_active = lambda: True
    D.GradReducer.issue = staticmethod(lambda t: _Work(t))
    F_._Exchange.reduce = _reduce          # world stays 1 (the BN counts must), the exchange is issued regardless
    sys.argv = [os.path.join(ROOT, 'bench.py')] + argv
    runpy.run_path(sys.argv[0], run_name='__main__')
