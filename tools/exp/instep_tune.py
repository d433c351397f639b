#!/usr/bin/env python
"""In-step check of the GEMM variant table: the shipped choices were timed kernel by kernel; here one signature at a time is
switched to an alternative variant and the WHOLE step is timed (bench.py as a child process with S4F_TUNE_CACHE=<temp table>),
with the unmodified table re-run regularly to follow the box's drift.
  python tools/exp/instep_tune.py <shard> <nshards> [workload]          (GPU box; ~20 s per run)"""
import json
import os
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
TABLE = os.path.join(ROOT, 's4former_amd', 'tuned_gfx950.json')


def run(table_path, workload):
    env = dict(os.environ, S4F_TUNE_CACHE=table_path)
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--steps', '24', '--warmup', '8', '--no-cpu-baseline',
                        '--no-kernel-profile', '--workload', workload], env=env, capture_output=True, text=True, timeout=300)
    for line in r.stdout.splitlines():
        if line.startswith('{"metric"'):
            d = json.loads(line)
            return d['ms_per_step'] if abs(d['losses']['loss']) < 1e3 else None
    return None


def main():
    shard, nshards = int(sys.argv[1]), int(sys.argv[2])
    workload = sys.argv[3] if len(sys.argv) > 3 else 'semi'
    rows = 16400 if workload == 'semi' else None
    table = json.load(open(TABLE))
    cands = []
    for k, v in table.items():
        key = eval(k)
        if len(key) < 14:
            continue
        a_mode, b_mode, M, N, K = key[:5]
        atomic = key[9]
        if atomic or a_mode not in (0, 2) or b_mode != 0:
            continue
        if workload == 'semi' and a_mode == 0 and M not in (16400, 8200):
            continue
        if workload == 'sup' and a_mode == 0 and M != 8200:
            continue
        if workload == 'semi768' and a_mode == 0 and M not in (18440, 9220):
            continue
        if workload == 'semi768' and a_mode == 2 and key[7] and key[7][0] != 4:
            continue
        if workload == 'sup' and a_mode == 2 and key[7] and key[7][0] != 8:
            continue
        if workload == 'semi' and a_mode == 2 and key[7] and key[7][0] != 8:
            continue
        alts = [h for h in (2, 4, 8, 9, 10) if h != v[0]]
        if N % 256:
            alts = [h for h in alts if h not in (4, 10)]
        if N % 192 or a_mode != 0:
            alts = [h for h in alts if h not in (8, 9)]
        for h in alts:
            cands.append((k, v, h))
    if os.environ.get('TUNE_ATOMIC'):          # split-K problems: neighbouring split factors instead of other variants
        cands = []
        for k, v in table.items():
            key = eval(k)
            if key[0] == 'wgrad_grouped':
                if (workload == 'semi' and key[1][2] != 16400) or (workload == 'semi768' and key[1][2] != 18440) or \
                        (workload == 'sup' and key[1][2] != 8200):
                    continue
                cands += [(k, v, (v[0], s_)) for s_ in (1, 3, 4) if s_ != v[1]]
            elif len(key) >= 14 and key[9]:
                if workload == 'semi' and ((key[7] and key[7][0] != 8) or key[4] in (18440, 147456, 589824, 36864, 9216, 8200)):
                    continue
                if workload == 'semi768' and not ((key[7] and key[7][0] == 4) or key[4] in (18440, 147456, 589824)):
                    continue
                if workload == 'sup' and not (key[4] == 8200 or (key[0], key[1]) == (2, 0)):
                    continue
                cands += [(k, v, (v[0], s_)) for s_ in sorted({max(1, v[1] // 2), v[1] * 2, max(1, (v[1] * 3) // 4), (v[1] * 3) // 2}) if s_ != v[1]]
    cands = cands[shard::nshards]
    print(f'{len(cands)} candidates in shard {shard}/{nshards}', flush=True)
    base = []
    with tempfile.TemporaryDirectory() as td:
        tp = os.path.join(td, 't.json')
        for i, (k, v, h) in enumerate(cands):
            if i % 6 == 0:
                json.dump(table, open(tp, 'w'))
                b = run(tp, workload)
                base.append(b)
                print(f'   baseline {b}', flush=True)
            mod = dict(table)
            mod[k] = list(h) if isinstance(h, tuple) else [h, v[1]]
            json.dump(mod, open(tp, 'w'))
            ms = run(tp, workload)
            key = eval(k)
            d = None if (ms is None or base[-1] is None) else ms - base[-1]
            if isinstance(h, tuple):
                print(f'{str(key[:5])[:60]:60s}: {v} -> {list(h)}: {ms} ms ({"n/a" if d is None else f"{d:+.3f}"})', flush=True)
                continue
            print(f'{str(key[:5]):34s} act {key[8]} f32 {int(key[10])} t {int(key[11])} resid {int(key[12])}: {v[0]:2d} -> {h:2d}: {ms} ms '
                  f'({"n/a" if d is None else f"{d:+.3f}"})', flush=True)


if __name__ == '__main__':
    main()
