"""Ordered per-launch listing of one step from a rocprofv3 --kernel-trace rocpd database (serialized runs: every kernel in
issue order with its duration, grid and gap to its predecessor).  python tools/launch_list.py <results.db> [steps_back]"""
import re
import sqlite3
import sys

c = sqlite3.connect(sys.argv[1])
rows = list(c.execute("select name,start,end,grid_x,grid_y,grid_z,workgroup_x,queue_id from kernels order by start"))
# (step boundaries: the EMA launch at the head of a step - or, with the double-buffered teacher of round 5, the first of the step's two patch-embedding gathers)
em = [r[1] for r in rows if 'ema_kernel' in r[0]]
if len(em) < 4:
    em = sorted(r[1] for r in rows if 'im2col16' in r[0])[0::2]
k = int(sys.argv[2]) if len(sys.argv) > 2 else 2
t0, t1 = em[-k - 1], em[-k]
R = [r for r in rows if t0 <= r[1] < t1]


def short(n):
    n = re.sub(r'\(anonymous namespace\)::', '', n).replace('void ', '')
    m = re.match(r'_ZN12_GLOBAL__N_1\d+([a-z_0-9]+?)I(.*)', n)
    if m:
        return m.group(1) + ' ' + m.group(2)[:24]
    return n[:56]


print(f'step {(t1 - t0) / 1e6:.2f} ms, {len(R)} kernels')
prev = None
for n, s, e, gx, gy, gz, wx, q in R:
    gap = (s - prev) / 1e3 if prev is not None else 0.0
    prev = e if prev is None else max(prev, e)
    print(f'{(s - t0) / 1e6:8.3f} ms  {(e - s) / 1e3:8.1f} us  gap {gap:6.1f}  q{q} blk {gx // max(wx, 1):6d}x{gy}x{gz:<3d} {short(n)}')
