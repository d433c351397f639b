"""N > 1 path on ONE GPU: two ranks over gloo (RCCL needs one GPU per rank; the collectives, the SyncBN statistics
exchange, the per-range gradient all-reduce during backward, the eager SGD behind it and the rank-0-only reporting are the
same code).  Runs LAST (file name) and on the TINY model with hard time caps: a multi-process test must never gate the
kernel-parity tests.  A stalled rank dumps its stacks to a file and exits; the dumps are printed on failure.

Asserted (reference mmseg/apis/train.py:129-138, segmentors/base.py:257-272):
  * 2-rank losses (mean over ranks) == 1-rank losses of the concatenated global batch   (SyncBN == BN over the concatenation)
  * state after two steps (weights, BN running statistics, EMA teacher) == the 1-rank run's
  * replicas bit-identical: student arena, SGD momentum arena, teacher arena
for the default N > 1 path and for the opt-in variants (lockstep heads, stream layout, no eager SGD)."""
import glob
import json
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
WORKER = os.path.join(ROOT, 'tests', 'dist_worker.py')
CAP = 120          # seconds per launched job


def _dumps(d):
    out = []
    for f in sorted(glob.glob(os.path.join(d, 'watchdog_rank*.txt'))):
        txt = open(f).read()
        if txt.strip():
            out.append(f'---- {os.path.basename(f)}\n{txt[-6000:]}')
    return '\n'.join(out)


def _run(cmd, env, out_dir):
    try:
        r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=CAP)
    except subprocess.TimeoutExpired as e:
        so = (e.stdout or b'').decode(errors='replace') if isinstance(e.stdout, bytes) else (e.stdout or '')
        se = (e.stderr or b'').decode(errors='replace') if isinstance(e.stderr, bytes) else (e.stderr or '')
        pytest.fail(f'timed out after {CAP} s: {" ".join(cmd[-8:])}\n{so[-2000:]}\n{se[-4000:]}\n{_dumps(out_dir)}')
    assert r.returncode == 0, f'rc {r.returncode}\n{r.stdout[-2000:]}\n{r.stderr[-4000:]}\n{_dumps(out_dir)}'
    return r


def _run_ranks(n, port, script_args, env, out_dir):
    """n ranks as plain child processes with the RANK / WORLD_SIZE / MASTER_* environment torch.distributed.run would set
    (no elastic agent in between: fewer moving parts under a test harness); every child gets the hard time cap, a stalled
    group is killed by PID and the per-rank stack dumps are printed"""
    import time
    procs = []
    for r in range(n):
        e = dict(env, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port),
                 OMP_NUM_THREADS='1')
        procs.append(subprocess.Popen([sys.executable] + script_args, cwd=ROOT, env=e, stdout=subprocess.PIPE,
                                      stderr=subprocess.STDOUT, text=True))
    deadline = time.time() + CAP
    outs = [''] * n
    try:
        for i, pr in enumerate(procs):
            try:
                outs[i], _ = pr.communicate(timeout=max(1.0, deadline - time.time()))
            except subprocess.TimeoutExpired:
                for q in procs:
                    if q.poll() is None:
                        q.kill()
                for k, q in enumerate(procs):
                    try:
                        outs[k] = (outs[k] or '') + (q.communicate(timeout=10)[0] or '')
                    except Exception:
                        pass
                pytest.fail(f'{n} ranks did not finish within {CAP} s\n' + '\n'.join(o[-2000:] for o in outs) + '\n' + _dumps(out_dir))
    finally:
        for q in procs:
            if q.poll() is None:
                q.kill()
    bad = [(i, pr.returncode) for i, pr in enumerate(procs) if pr.returncode != 0]
    assert not bad, f'ranks failed: {bad}\n' + '\n'.join(o[-3000:] for o in outs) + '\n' + _dumps(out_dir)
    return outs


def _torchrun(nproc, port, script_args):
    return [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(nproc), '--master-addr',
            '127.0.0.1', '--master-port', str(port)] + script_args


def _env(**extra):
    env = dict(os.environ, S4F_DIST_BACKEND='gloo', HSA_ENABLE_IPC_MODE_LEGACY='0', S4F_AUTOTUNE='0')
    env.update(extra)
    return env


def _load(d, rank):
    z = np.load(os.path.join(d, f'rank{rank}.npz'), allow_pickle=False)
    return {k: z[k] for k in z.files}


@pytest.fixture(scope='module')
def single(tmp_path_factory):
    """the expected result: the whole global batch in ONE process (fp32 parity mode), per flag set"""
    res = {}
    for flags in ('pasa', 'plain'):
        d = str(tmp_path_factory.mktemp(f'w1_{flags}'))
        _run([sys.executable, WORKER, '--out', d, '--flags', flags], _env(), d)
        res[flags] = _load(d, 0)
    return res


VARIANTS = {
    'default': {},
    'no_eager_sgd': dict(S4F_EAGER_SGD='0'),
    'lockstep_heads': dict(S4F_AUX_LOCKSTEP='1', S4F_DECODE_LOCKSTEP='1'),
    'lockstep_aux_only': dict(S4F_AUX_LOCKSTEP='1', S4F_DECODE_LOCKSTEP='0'),       # (round 6: a setting autotune_schedule may pick)
    # round 3: final ranges merged two tiny layers at a time (the production size merges three DeiT-B layers), and the round-2
    # schedule (one collective per range).  Round 4: the head lockstep is opt-in again (the default runs one head after the other);
    # the empirical first-use order of the streams has its own switch
    'coalesced_buckets': dict(S4F_BUCKET_MIN_ELEMS='1500000'),
    'round2_schedule': dict(S4F_AUX_LOCKSTEP='0', S4F_DECODE_LOCKSTEP='0', S4F_BUCKET_MIN_ELEMS='1'),
    'natural_stream_order': dict(S4F_STREAM_ORDER='0'),
}


@pytest.mark.parametrize('variant,flags', [('default', 'pasa'), ('default', 'plain'), ('no_eager_sgd', 'pasa'),
                                           ('lockstep_heads', 'pasa'), ('lockstep_aux_only', 'plain'), ('coalesced_buckets', 'pasa'),
                                           ('round2_schedule', 'plain'), ('natural_stream_order', 'plain')])
def test_two_ranks_equal_one_rank_on_the_concatenated_batch(variant, flags, single, tmp_path):
    d = str(tmp_path)
    port = 29600 + sorted(VARIANTS).index(variant) * 2 + (flags == 'plain')
    _run_ranks(2, port, [WORKER, '--out', d, '--flags', flags], _env(**VARIANTS[variant]), d)
    r0, r1, ref = _load(d, 0), _load(d, 1), single[flags]
    assert json.loads(str(r0['meta']))['world'] == 2
    # replicas bit-identical after two steps
    for k in ('student_sha', 'mom_sha', 'teacher_sha'):
        assert str(r0[k]) == str(r1[k]), f'{k} differs between the ranks'
    msgs = []
    for it in range(2):
        keys = [str(k) for k in ref[f'it{it}_loss_keys']]
        assert [str(k) for k in r0[f'it{it}_loss_keys']] == keys
        # logged values are already the mean over the ranks (batched all-reduce of _parse_losses): identical on both ranks
        assert np.array_equal(r0[f'it{it}_loss_vals'], r1[f'it{it}_loss_vals'])
        for k, got, want in zip(keys, r0[f'it{it}_loss_vals'], ref[f'it{it}_loss_vals']):
            if abs(got - want) > 1e-5 * abs(want) + 1e-7:
                msgs.append(f'it{it} {k}: 2 ranks {got:.7f} vs 1 rank {want:.7f}')
        # the loss backward runs on stays local; its rank mean is the logged total
        lm = 0.5 * (float(r0[f'it{it}_local_loss']) + float(r1[f'it{it}_local_loss']))
        tot = float(r0[f'it{it}_loss_vals'][keys.index('loss')])
        assert abs(lm - tot) <= 1e-6 * abs(tot)
    # state after two optimiser steps: weights, BN running statistics (SyncBN == BN of the concatenated batch), EMA teacher
    assert [str(k) for k in r0['state_keys']] == [str(k) for k in ref['state_keys']]
    for k, got, want, gs, ws in zip(ref['state_keys'], r0['state_abs_sum'], ref['state_abs_sum'], r0['state_sum'], ref['state_sum']):
        if abs(got - want) > 2e-5 * abs(want) + 1e-7 or abs(gs - ws) > 2e-5 * abs(want) + 1e-6:
            msgs.append(f'state {k}: |.|_1 {got:.7f} vs {want:.7f}, sum {gs:.7f} vs {ws:.7f}')
    assert np.array_equal(r0['nbt'], ref['nbt'])
    assert not msgs, '\n'.join(msgs[:20])


def test_two_ranks_pick_their_schedule_by_measurement_and_check_their_first_steps(single, tmp_path):
    """Round 6 (reference: the one fixed DDP schedule of mmseg/apis/train.py:129-138).  (1) The first steps of a multi-rank run
    verify, before every gradient collective, that all ranks are about to reduce the same span (GradReducer.check_steps, default
    3): the two recorded steps run under that check and must still equal the one-rank run.  (2) dist.autotune_schedule times the
    head-lockstep settings and two bucket sizes on the job's own ranks and keeps the fastest: every candidate is timed, both
    ranks keep the SAME setting, and the replicas are still bit-identical after the tuning steps.  Each setting it may pick is
    held against the one-rank run by the variants above (default, lockstep_aux_only, lockstep_heads, coalesced_buckets)."""
    d = str(tmp_path)
    _run_ranks(2, 29650, [WORKER, '--out', d, '--flags', 'plain', '--autotune'], _env(), d)
    r0, r1, ref = _load(d, 0), _load(d, 1), single['plain']
    for it in range(2):
        assert np.allclose(r0[f'it{it}_loss_vals'], ref[f'it{it}_loss_vals'], rtol=1e-5, atol=1e-7), it
    s0, s1 = json.loads(str(r0['schedule'])), json.loads(str(r1['schedule']))
    assert s0 == s1, (s0, s1)                                        # the ranks agreed (MAX over ranks per candidate)
    assert {'lockstep=off', 'lockstep=aux', 'lockstep=aux+decode'} <= set(s0) and sum(k.startswith('bucket_min_elems=') for k in s0) == 2, s0
    assert s0['chosen']['lockstep'] in ('off', 'aux', 'aux+decode') and s0['chosen']['bucket_min_elems'] > 0
    assert all(v > 0 for k, v in s0.items() if k != 'chosen')
    assert int(r0['tuned_steps']) == int(r1['tuned_steps']) == 5 * 3            # 5 candidates x (1 untimed + 2 timed) steps
    for k in ('tuned_student_sha', 'tuned_mom_sha', 'tuned_teacher_sha'):
        assert str(r0[k]) == str(r1[k]), f'{k} differs between the ranks after the tuning steps'
    assert np.isfinite(float(r0['tuned_last_loss']))
    assert int(r0['check_steps_left']) == 0                          # the three checked steps are behind us


def test_a_divergent_flush_order_is_reported_not_hung(tmp_path):
    """the check itself: two ranks that hand DIFFERENT spans to the reducer as their first collective get an S4FError naming both
    spans (tests/dist_worker.py is not involved: a bare GradReducer over a flat tensor, gloo)"""
    d = str(tmp_path)
    code = (
        "import os, sys, torch, torch.distributed as dist\n"
        "sys.path.insert(0, %r)\n"
        "from s4former_amd.dist import GradReducer, init_distributed\n"
        "from s4former_amd._lib import S4FError\n"
        "rank, local, world = init_distributed(timeout_s=30)\n"
        "g = torch.ones(4096, device='cuda')\n"
        "r = GradReducer()\n"
        "a, b = (0, 1024) if rank == 0 else (1024, 2048)\n"
        "try:\n"
        "    r._launch(g[a:b], (a, b))\n"
        "    r.wait()\n"
        "    print('NO-ERROR')\n"
        "except S4FError as e:\n"
        "    print('CAUGHT', 'DIFFERENT spans' in str(e), flush=True)\n"
        "dist.barrier()\n" % ROOT)
    outs = _run_ranks(2, 29652, ['-c', code], _env(), d)
    assert all('CAUGHT True' in o for o in outs), outs


def test_two_ranks_bf16_replicas_stay_identical(tmp_path):
    """perf mode through the same path: finite losses, replicas bit-identical"""
    d = str(tmp_path)
    _run_ranks(2, 29620, [WORKER, '--out', d, '--flags', 'plain', '--dtype', 'bf16'], _env(), d)
    r0, r1 = _load(d, 0), _load(d, 1)
    for k in ('student_sha', 'mom_sha', 'teacher_sha'):
        assert str(r0[k]) == str(r1[k]), f'{k} differs between the ranks'
    assert np.all(np.isfinite(r0['it1_loss_vals']))


def test_one_rank_rccl_group_runs_the_data_path(single, tmp_path):
    """the collectives of the step through backend 'nccl' (RCCL) itself - a group of ONE rank, every exchange forced on: the
    result must equal the plain single-process run (all-reduce over one rank is the identity), and the gradient ranges and
    SyncBN exchanges must really have been issued.  Skipped (not failed) when RCCL cannot initialise on the box."""
    d = str(tmp_path)
    env = _env(S4F_BUCKET_MIN_ELEMS='1')          # every final range goes out at once: the overlapped per-range path (the tiny
    env.pop('S4F_DIST_BACKEND')                   # model's layers are far below the production bucket size of three DeiT-B layers)
    try:
        r = subprocess.run([sys.executable, WORKER, '--out', d, '--flags', 'plain', '--rccl-one-rank'], cwd=ROOT, env=env,
                           capture_output=True, text=True, timeout=CAP)
    except subprocess.TimeoutExpired as e:
        pytest.fail(f'one-rank RCCL run timed out after {CAP} s\n{e.stdout}\n{e.stderr}\n{_dumps(d)}')
    if r.returncode == 77:
        pytest.skip('RCCL did not initialise on this box: ' + r.stdout[-300:])
    assert r.returncode == 0, f'rc {r.returncode}\n{r.stdout[-2000:]}\n{r.stderr[-4000:]}\n{_dumps(d)}'
    got, ref = _load(d, 0), single['plain']
    assert json.loads(str(got['meta']))['backend'] == 'nccl'
    # round 5: the first-use stream order is MEASURED at set-up (pairwise concurrency of the package's streams): the record must
    # be there and complete; whether every pair came apart is this runtime's business (a warning, not an error)
    lay = json.loads(str(got['stream_layout']))
    assert lay is not None and 'error' not in lay and len(lay['pairs']) >= 6 and isinstance(lay['ok'], bool), lay
    n_grad, n_bn = (int(v) for v in got['issued'])
    assert n_grad >= 2 * 4 and n_bn >= 2 * 10, (n_grad, n_bn)       # per step: >= one range per encoder layer, >= one exchange per BN call
    # (two processes: the fp32 atomics of the split-K sums differ in their last bits from run to run - the bounds of the
    #  two-rank comparison above)
    for it in range(2):
        assert np.allclose(got[f'it{it}_loss_vals'], ref[f'it{it}_loss_vals'], rtol=1e-5, atol=1e-7), it
    assert np.allclose(got['state_abs_sum'], ref['state_abs_sum'], rtol=2e-5, atol=1e-7)
    assert np.array_equal(got['nbt'], ref['nbt'])


def test_bench_two_ranks_gloo(tmp_path):
    """bench.py's own N > 1 control flow (rank-symmetric profiled step, rank-0-only reporting, barriers) on the tiny workload,
    launched exactly as the driver launches it (python -m torch.distributed.run)"""
    d = str(tmp_path)
    env = _env(S4F_BENCH_WATCHDOG='90', S4F_WATCHDOG_DIR=d, S4F_DIST_TIMEOUT_S='60', S4F_AUTOTUNE_STEPS='2')
    r = _run(_torchrun(2, 29630, [os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--steps', '2', '--warmup', '1',
                                  '--workload', 'tiny', '--no-cpu-baseline']), env, d)
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith('{"metric"')]
    assert len(lines) == 1, r.stdout[-2000:]
    out = json.loads(lines[0])
    assert out['n_gpus'] == 2 and out['value'] > 0 and out['roofline'] is not None
    assert out['config']['dist_backend'] == 'gloo' and out['config']['ranks_seen'] == 2
    assert 'pairs' in (out['config']['stream_layout'] or {}), out['config']
    # round 6: the schedule picked by measurement during warm-up and the per-collective latency are in the line
    assert out['config']['schedule'] and out['config']['schedule']['chosen'], out['config']
    assert out['config']['grad_collective_latency']['collectives'] >= 2, out['config']
    assert out['config']['flush_order_checked_steps'] == 3
    assert abs(out['losses']['loss']) < 1e3
