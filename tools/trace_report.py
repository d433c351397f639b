"""Per-step report from a rocprofv3 --kernel-trace rocpd database: per-queue busy time, per-kernel time on the critical
(busiest) queue, idle gaps.  python tools/trace_report.py <results.db> [steps_back]"""
import collections
import re
import sqlite3
import sys

db = sys.argv[1]
c = sqlite3.connect(db)
rows = list(c.execute("select name, start, end, stream_id, queue_id from kernels order by start"))
# (step boundaries: the EMA launch at the head of a step - or, with the double-buffered teacher of round 5, the first of the step's two patch-embedding gathers)
em = [r[1] for r in rows if 'ema_kernel' in r[0]]
if len(em) < 4:
    em = sorted(r[1] for r in rows if 'im2col16' in r[0])[0::2]
k = int(sys.argv[2]) if len(sys.argv) > 2 else 2
t0, t1 = em[-k - 1], em[-k]
R = [r for r in rows if t0 <= r[1] < t1]
print(f'step {1e-6 * (t1 - t0):.2f} ms, {len(R)} kernels')
per = collections.defaultdict(float)
for r in R:
    per[r[4]] += (r[2] - r[1]) / 1e6
print('busy ms per queue', {q: round(v, 2) for q, v in per.items()})
mainq = max(per, key=per.get)


def short(n):
    n = re.sub(r'\(anonymous namespace\)::', '', n).replace('void ', '')
    m = re.match(r'_ZN12_GLOBAL__N_1\d+([a-z_0-9]+?)I', n)
    if m:
        return m.group(1) + ' ' + n[-28:]
    return n[:64]


for q in per:
    agg = collections.defaultdict(lambda: [0, 0.0])
    for n, s, e, st, qq in R:
        if qq == q:
            a = agg[short(n)]
            a[0] += 1
            a[1] += (e - s) / 1e6
    print(f'--- queue {q} ({"critical" if q == mainq else "side"})')
    for n, v in sorted(agg.items(), key=lambda kv: -kv[1][1])[:28]:
        print(f'  {n:66s} {v[0]:4d} {v[1]:7.3f} ms  avg {1e3 * v[1] / v[0]:7.1f} us')
iv = sorted((r[1], r[2]) for r in R if r[4] == mainq)
gaps = sum(max(0, iv[i + 1][0] - max(x[1] for x in iv[:i + 1][-4:])) for i in range(len(iv) - 1))
print(f'critical queue: idle between kernels {gaps / 1e6:.2f} ms')
