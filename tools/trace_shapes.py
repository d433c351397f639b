"""Per-(kernel, grid) time table of one step from a rocprofv3 --kernel-trace rocpd database (serialized runs:
S4F_SIDE_STREAM=0 S4F_HEAD_STREAMS=0).  python tools/trace_shapes.py <results.db> [steps_back] [filter]"""
import collections
import re
import sqlite3
import sys

c = sqlite3.connect(sys.argv[1])
rows = list(c.execute("select name,start,end,grid_x,grid_z,workgroup_x,vgpr_count,accum_vgpr_count,lds_size from kernels order by start"))
# (step boundaries: the EMA launch at the head of a step - or, with the double-buffered teacher of round 5, the first of the step's two patch-embedding gathers)
em = [r[1] for r in rows if 'ema_kernel' in r[0]]
if len(em) < 4:
    em = sorted(r[1] for r in rows if 'im2col16' in r[0])[0::2]
k = int(sys.argv[2]) if len(sys.argv) > 2 else 2
flt = sys.argv[3] if len(sys.argv) > 3 else ''
t0, t1 = em[-k - 1], em[-k]
R = [r for r in rows if t0 <= r[1] < t1]
agg = collections.defaultdict(lambda: [0, 0.0])


def short(n):
    n = re.sub(r'\(anonymous namespace\)::', '', n).replace('void ', '')
    return n[:60]


for n, s, e, gx, gz, wx, v, av, lds in R:
    if flt in n:
        key = (short(n), gx // wx, gz, v, av, lds)
        agg[key][0] += 1
        agg[key][1] += (e - s) / 1e3
tot = 0
for key, v in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    tot += v[1]
    print(f'{key[0]:60s} blocks={key[1]:6d} z={key[2]:3d} vgpr={key[3]:3d} agpr={key[4]:3d} lds={key[5]:6d} '
          f'n={v[0]:3d} tot={v[1]:8.1f}us avg={v[1] / v[0]:7.1f}')
print(f'total {tot / 1e3:.3f} ms')
