"""how often does every arena range report 'gradient final' per step? (debug aid) python tools/exp/range_count.py [gold]"""
import collections
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from tests.test_step_gpu import build_product, load_gold, run_product  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else 'mt_pasa'
z, meta = load_gold(name)
model, opt, sched = build_product(meta, 'fp32')
model.ensure_engine(torch.device('cuda', 0))
cnt = collections.Counter()
model.student_store.on_range_done = lambda a, b: cnt.update([(a, b)])
run_product(model, opt, sched, meta, iters=1)
print(name, 'flags', meta['flags'])
for (a, b), n in sorted(cnt.items()):
    print(f'range [{a}, {b}) reported {n} time(s)')
