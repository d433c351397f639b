"""The package's kernel-path / schedule switches (SWITCHES.md) against the reference's golden, one switch at a time.

The defaults are what the rest of the GPU suite runs; this file runs the tiny S4Former step - the paper's full method
(`step_mt_ours`: PASA with its masked and plain student pass, CutMix / PatchShuffle, NCR, EMA teacher: the fixture that touches
the most code) - once per switch setting in a FRESH process (most switches are read when the package is imported) and holds every
run to the golden made from the reference's own code with the bounds of tests/test_step_gpu.py::test_step_vs_golden: named losses,
per-parameter gradient norms, state after two optimiser steps.  A switch that selects another kernel or another schedule must
not select another result."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
WORKER = os.path.join(ROOT, 'tests', 'step_worker.py')
GOLD = os.path.join(ROOT, 'tests', 'golden')

# (dtype, environment): every entry is one non-default setting of SWITCHES.md that changes which kernels / streams / launch paths run
MATRIX = [
    ('bf16', dict(S4F_FUSED_LAUNCH='0')),                       # per-kernel launches instead of s4f_encoder_layer_fwd / _bwd
    ('bf16', dict(S4F_ATTN_BWD_FUSED='0')),                     # two-kernel attention backward
    ('bf16', dict(S4F_ATTN_FWD2='0')),                          # round-1 attention forward
    ('bf16', dict(S4F_ATTN_NW='2')),                            # two waves per attention block
    ('bf16', dict(S4F_EMA_DOUBLE='0')),                         # in-place EMA at the head of forward_train
    ('bf16', dict(S4F_EMA_DOUBLE='0', S4F_EMA_OVERLAP='1')),    # EMA of the arena's tail on the side stream
    ('bf16', dict(S4F_TAP_SPLIT='0')),                          # autograd slices instead of the shared tap-gradient buffer
    ('bf16', dict(S4F_SIDE_STREAM='0', S4F_HEAD_STREAMS='0')),  # the whole step on one stream
    ('bf16', dict(S4F_LAYER_WG_SIDE='0')),                      # the layers' grouped weight gradient inside the chain
    ('bf16', dict(S4F_GELU_Q8='1')),                            # 8-bit gelu'
    ('bf16', dict(S4F_FOLD_COLSUM='0')),                        # separate column-sum / BatchNorm-statistics passes
    ('bf16', dict(S4F_FUSE_CLS_FWD='0', S4F_FUSE_CLS_GRAD='0', S4F_FUSE_CLS_WGRAD='0', S4F_SKIP_MASKED_COPY='0')),   # unfused last head stage
    ('bf16', dict(S4F_UNSUP_STREAM='decode2')),                 # a third head stream for the pseudo-labelled decode call
    ('bf16', dict(S4F_AUTOTUNE='0')),                           # shipped GEMM table only, automatic choice for unknown signatures
    ('bf16', dict(S4F_TEACHER_PRECISE='1')),                    # round 6: fp32 teacher under the bf16 student
    ('bf16', dict(S4F_TEACHER_LAST_FP32='1')),                  # fp32 operands in the teacher head's last stage
    ('fp32', dict(S4F_FUSED_LAUNCH='0')),
    ('fp32', dict(S4F_SIDE_STREAM='0', S4F_HEAD_STREAMS='0')),
    ('fp32', dict(S4F_TAP_SPLIT='0', S4F_EMA_DOUBLE='0')),
]
LTOL = {'fp32': (1e-4, 1e-3), 'bf16': (3e-3, 1.2e-2)}         # tests/test_step_gpu.py::test_step_vs_golden
GTOL = {'fp32': (1e-3, 5e-3), 'bf16': (6e-2, 7e-2)}
WTOL = {'fp32': 2e-4, 'bf16': 2e-2}


@pytest.mark.parametrize('dtype,env', MATRIX, ids=[f"{d}-{'+'.join(f'{k[4:]}={v}' for k, v in e.items())}" for d, e in MATRIX])
def test_a_switch_selects_another_path_not_another_result(dtype, env, tmp_path):
    out = str(tmp_path / 'r.npz')
    e = dict(os.environ)
    for k in list(e):
        if k.startswith('S4F_') and k not in ('S4F_LIB',):
            del e[k]                                               # the matrix entry is the ONLY non-default switch of the run
    e.update(env)
    r = subprocess.run([sys.executable, WORKER, '--name', 'mt_ours', '--dtype', dtype, '--out', out], cwd=ROOT, env=e,
                       capture_output=True, text=True, timeout=240)
    assert r.returncode == 0, f'{env}: rc {r.returncode}\n{r.stdout[-1500:]}\n{r.stderr[-3000:]}'
    got = np.load(out, allow_pickle=False)
    z = np.load(os.path.join(GOLD, 'step_mt_ours.npz'), allow_pickle=False)
    msgs = []
    for it in range(2):
        ref = dict(zip([str(k) for k in z[f'it{it}_loss_keys']], z[f'it{it}_loss_vals']))
        mine = dict(zip([str(k) for k in got[f'it{it}_loss_keys']], got[f'it{it}_loss_vals']))
        assert sorted(k for k in ref if 'loss' in k) == sorted(k for k in mine if 'loss' in k and k != 'loss')
        for k, v in ref.items():
            if 'loss' in k and abs(mine[k] - v) > LTOL[dtype][it] * abs(v):
                msgs.append(f'it{it} {k}: {mine[k]:.6f} vs {v:.6f}')
        gref = dict(zip([str(k) for k in z[f'it{it}_gn_keys']], z[f'it{it}_gn_vals']))
        gmine = dict(zip([str(k) for k in got[f'it{it}_gn_keys']], got[f'it{it}_gn_vals']))
        assert sorted(gref) == sorted(gmine)
        worst = max((abs(gmine[k] - v) / (abs(v) + 1e-12), k) for k, v in gref.items())
        if worst[0] > GTOL[dtype][it]:
            msgs.append(f'it{it} gradient norm {worst[1]}: rel {worst[0]:.2e}')
    for k, ref_v, v in zip(got['final_keys'], z['final_abs_sum'], got['final_abs_sum']):
        if abs(v - ref_v) > WTOL[dtype] * abs(ref_v) + 1e-9:
            msgs.append(f'final |{k}|_1: {v:.6f} vs {ref_v:.6f}')
    assert not msgs, f'{json.dumps(env)}:\n' + '\n'.join(msgs[:20])
