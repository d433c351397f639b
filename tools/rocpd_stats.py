"""per-kernel (and per-grid) duration summary of a rocprofv3 .db (rocpd sqlite) file: python tools/rocpd_stats.py x_results.db [--by-grid]"""
import sqlite3
import sys


def main():
    path = sys.argv[1]
    by_grid = '--by-grid' in sys.argv
    db = sqlite3.connect(path)
    cur = db.cursor()
    tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
    kd = [t for t in tabs if 'rocpd_kernel_dispatch' in t][0]
    ks = [t for t in tabs if 'rocpd_info_kernel_symbol' in t][0]
    names = {r[0]: r[1] for r in cur.execute(f'select id, kernel_name from {ks}')}
    agg = {}
    for kid, st, en, gx, wx in cur.execute(f'select kernel_id, start, end, grid_size_x, workgroup_size_x from {kd}'):
        nm = names.get(kid, str(kid))
        key = (nm, gx // max(wx, 1)) if by_grid else (nm,)
        d = agg.setdefault(key, [0, 0.0, 1e30, 0.0])
        d[0] += 1
        d[1] += (en - st) / 1e3
        d[2] = min(d[2], (en - st) / 1e3)
        d[3] = max(d[3], (en - st) / 1e3)
    tot = sum(v[1] for v in agg.values())
    print(f'{"kernel":90s} {"blocks":>8s} {"calls":>6s} {"total us":>12s} {"avg us":>9s} {"min us":>9s} {"max us":>9s} {"%":>6s}')
    for key, v in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        nm = key[0][:90]
        blocks = str(key[1]) if by_grid else ''
        print(f'{nm:90s} {blocks:>8s} {v[0]:6d} {v[1]:12.1f} {v[1] / v[0]:9.1f} {v[2]:9.1f} {v[3]:9.1f} {100 * v[1] / tot:6.2f}')


if __name__ == '__main__':
    main()
