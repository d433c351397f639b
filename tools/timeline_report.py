"""Timeline of one multi-stream step from a rocprofv3 --kernel-trace rocpd database: for every slice of the step, which
kernel families are on the GPU, how many queues are busy, and the time nothing / exactly one queue runs.
python tools/timeline_report.py <results.db> [steps_back] [slice_ms]"""
import collections
import re
import sqlite3
import sys

db = sys.argv[1]
k = int(sys.argv[2]) if len(sys.argv) > 2 else 2
dt = float(sys.argv[3]) if len(sys.argv) > 3 else 0.5
c = sqlite3.connect(db)
rows = list(c.execute("select name, start, end, stream_id, queue_id from kernels order by start"))
# step boundaries: the two patch-embedding gathers of a step (teacher, student) are its first kernels - with the double-buffered
# teacher (round 5) the EMA is no longer one launch at the head of the step
em = [r[1] for r in rows if 'ema_kernel' in r[0]]
if len(em) < k + 2:
    im = sorted(r[1] for r in rows if 'im2col16' in r[0])
    em = im[0::2]
t0, t1 = em[-k - 1], em[-k]
R = [r for r in rows if t0 <= r[1] < t1]


def fam(n):
    n = re.sub(r'\(anonymous namespace\)::', '', n).replace('void ', '')
    m = re.match(r'_ZN12_GLOBAL__N_1\d+([a-z_0-9]+?)I', n)
    if m:
        n = m.group(1)
    for key, f in (('gemm6_grouped', 'WG'), ('gemm2_grouped', 'WG'), ('gemm6', 'wg'), ('gemm5_kernel<2', 'CV'), ('gemm', 'G'),
                   ('attn_fwd', 'Af'), ('attn_dq', 'Aq'), ('attn_dkv', 'Ak'), ('ln_', 'ln'), ('bn_', 'bn'), ('upce', 'ce'),
                   ('sgd', 'SGD'), ('ema', 'EMA'), ('colsum', 'cs'), ('transpose', 'tr')):
        if key in n:
            return f
    return 'x'


queues = sorted({r[4] for r in R})
qname = {q: chr(ord('a') + i) for i, q in enumerate(queues)}
print(f'step {(t1 - t0) / 1e6:.2f} ms, {len(R)} kernels, queues {qname}')
# sweep: time with n queues busy
ev = []
for n, s, e, st, q in R:
    ev.append((s, 1, q))
    ev.append((e, -1, q))
ev.sort()
busy = collections.Counter()
act = collections.Counter()
last = t0
for t, d, q in ev:
    nb = sum(1 for v in act.values() if v > 0)
    busy[nb] += t - last
    last = t
    act[q] += d
print('ms with n queues busy:', {n: round(v / 1e6, 2) for n, v in sorted(busy.items())})
per = collections.defaultdict(float)
for r in R:
    per[r[4]] += (r[2] - r[1]) / 1e6
print('busy ms per queue', {qname[q]: round(v, 2) for q, v in per.items()})
nsl = int((t1 - t0) / 1e6 / dt) + 1
for i in range(nsl):
    a, b = t0 + i * dt * 1e6, t0 + (i + 1) * dt * 1e6
    cell = collections.defaultdict(lambda: collections.defaultdict(float))
    for n, s, e, st, q in R:
        ov = min(e, b) - max(s, a)
        if ov > 0:
            cell[q][fam(n)] += ov / (dt * 1e6)
    parts = []
    for q in queues:
        if q in cell:
            tot = sum(cell[q].values())
            top = sorted(cell[q].items(), key=lambda kv: -kv[1])[:3]
            parts.append(f'{qname[q]}:{tot:4.2f} ' + ','.join(f'{f}{v:.1f}' for f, v in top))
    print(f'{i * dt:6.1f} ms | ' + ' | '.join(parts))

# optional: list every kernel between two times (ms from step start): python tools/timeline_report.py db k dt t_from t_to
if len(sys.argv) > 5:
    a, b = t0 + float(sys.argv[4]) * 1e6, t0 + float(sys.argv[5]) * 1e6
    print(f'--- kernels starting in [{sys.argv[4]}, {sys.argv[5]}] ms')
    for n, s, e, st, q in R:
        if a <= s < b:
            nn = re.sub(r'\(anonymous namespace\)::', '', n).replace('void ', '')
            print(f'{(s - t0) / 1e6:8.3f} +{(e - s) / 1e3:7.1f}us {qname[q]} {nn[:90]}')
