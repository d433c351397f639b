"""N > 1 path of bench.py on ONE GPU: two ranks over gloo (RCCL needs one GPU per rank; the collectives, the SyncBN
statistics exchange, the per-layer gradient buckets and the rank-0-only reporting are the same code).  Guards against
rank-asymmetric collectives (a profiled extra step on rank 0 only once dead-locked every N > 1 run)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_two_ranks_gloo():
    env = dict(os.environ, S4F_DIST_BACKEND='gloo', S4F_BENCH_WATCHDOG='240', HSA_ENABLE_IPC_MODE_LEGACY='0')
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2', '--master-addr', '127.0.0.1',
           '--master-port', '29533', os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--steps', '1', '--warmup', '1',
           '--no-cpu-baseline']
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=420)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith('{"metric"')]
    assert len(lines) == 1, r.stdout[-2000:]
    out = json.loads(lines[0])
    assert out['n_gpus'] == 2 and out['value'] > 0 and out['roofline'] is not None
    assert abs(out['losses']['loss']) < 1e3
