"""GPU-side schedule of one step WITHOUT a tracer attached, from the HIP events bench.py records around every C-ABI launch
(bench.py --timeline file.json): per stream busy time, time with 0 / 1 / 2 / 3 streams busy, the gaps on the chain's
stream and a slice-by-slice view.  ATen kernels are not in it (only C-ABI launches carry events): a gap on a stream is
either an ATen kernel, a wait on another stream, or the host not having issued the next launch yet.
python tools/event_timeline.py <file.json> [slice_ms]"""
import collections
import json
import sys

E = json.load(open(sys.argv[1]))
dt = float(sys.argv[2]) if len(sys.argv) > 2 else 0.5
streams = []
for e in E:
    if e[2] not in streams:
        streams.append(e[2])
sname = {s: chr(ord('a') + i) for i, s in enumerate(streams)}
t_end = max(e[3] + e[4] for e in E)
t_beg = min(e[3] for e in E)
print(f'{len(E)} launches, {t_end - t_beg:.2f} ms from the first start to the last end, streams {len(streams)}')
per = collections.defaultdict(float)
for n, tag, st, t0, d in E:
    per[st] += d
print('busy ms per stream', {sname[s]: round(v, 2) for s, v in per.items()})
ev = []
for n, tag, st, t0, d in E:
    ev.append((t0, 1, st))
    ev.append((t0 + d, -1, st))
ev.sort()
busy = collections.Counter()
act = collections.Counter()
last = t_beg
for t, k, st in ev:
    nb = sum(1 for v in act.values() if v > 0)
    busy[nb] += t - last
    last = t
    act[st] += k
print('ms with n streams busy:', {n: round(v, 2) for n, v in sorted(busy.items())})


def fam(n, tag):
    if n == 's4f_gemm':
        return {0: 'G', 1: 'wg', 2: 'CV'}.get(tag[0] if tag else 0, 'G')
    for key, f in (('gemm_grouped', 'WG'), ('attention_fwd', 'Af'), ('attention_bwd', 'Ab'), ('layernorm', 'ln'), ('bn_', 'bn'),
                   ('upce', 'ce'), ('sgd', 'SGD'), ('ema', 'EMA'), ('colsum', 'cs'), ('transpose', 'tr'), ('layer_', 'L'),
                   ('head_', 'H')):
        if key in n:
            return f
    return 'x'


nsl = int((t_end - t_beg) / dt) + 1
for i in range(nsl):
    a, b = t_beg + i * dt, t_beg + (i + 1) * dt
    cell = collections.defaultdict(lambda: collections.defaultdict(float))
    for n, tag, st, t0, d in E:
        ov = min(t0 + d, b) - max(t0, a)
        if ov > 0:
            cell[st][fam(n, tag)] += ov / dt
    parts = []
    for s in streams:
        if s in cell:
            tot = sum(cell[s].values())
            top = sorted(cell[s].items(), key=lambda kv: -kv[1])[:3]
            parts.append(f'{sname[s]}:{tot:4.2f} ' + ','.join(f'{f}{v:.1f}' for f, v in top))
    print(f'{i * dt:6.1f} ms | ' + ' | '.join(parts))
