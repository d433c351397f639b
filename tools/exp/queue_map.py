"""which hardware queue does the k-th torch.cuda.Stream() land on?  Run under rocprofv3 --kernel-trace and read the queue ids:
rocprofv3 --kernel-trace -d out -o q -- python3 tools/exp/queue_map.py ; python3 tools/exp/queue_map.py report <db>"""
import sys

if len(sys.argv) > 2 and sys.argv[1] == 'report':
    import sqlite3
    c = sqlite3.connect(sys.argv[2])
    rows = list(c.execute("select name, grid_x, queue_id, stream_id from kernels order by start"))
    seen = {}
    for n, g, q, st in rows:
        if 'Fill' in n and g not in seen:
            seen[g] = (q, st)
    for g in sorted(seen):
        print(f'fill grid {g:6d} -> queue {seen[g][0]} stream {seen[g][1]}')
    sys.exit(0)

import torch

dev = torch.device('cuda', 0)
x = torch.empty(1 << 22, device=dev)
torch.cuda.synchronize()
x[:256 * 4 * 1].fill_(1.0)                 # null stream: the smallest grid
streams = [torch.cuda.Stream() for _ in range(12)]
for k, st in enumerate(streams):
    with torch.cuda.stream(st):
        x[:256 * 4 * (k + 2) * 16].fill_(float(k))
torch.cuda.synchronize()
hp = torch.cuda.Stream(priority=-1)
with torch.cuda.stream(hp):
    x[:256 * 4 * 400].fill_(3.0)
torch.cuda.synchronize()
