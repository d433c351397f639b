"""locate where eager SGD and the plain step differ (debug aid): python tools/exp/eager_debug.py [dtype]"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from tests.test_step_gpu import build_product, load_gold, run_product  # noqa: E402

dtype = sys.argv[1] if len(sys.argv) > 1 else 'fp32'
z, meta = load_gold('mt_pasa')
ref = None
for trial in range(6):
    eager = trial > 0 and os.environ.get("NO_EAGER") is None
    model, opt, sched = build_product(meta, dtype)
    model.ensure_engine(torch.device('cuda', 0))
    if eager:
        opt.attach_eager(model.student_store)
    run_product(model, opt, sched, meta, iters=3)
    s = model.student_store
    cur = (s.flat.clone(), s.mom.clone())
    if ref is None:
        ref = cur
        continue
    for k, (a, b) in enumerate(zip(ref, cur)):
        d = (a - b).abs()
        bad = d > 2e-6 * float(a.abs().max())
        if bad.any():
            idx = bad.nonzero().flatten()
            names = {}
            for e in s.entries:
                n = int(((idx >= e.off) & (idx < e.off + e.numel)).sum())
                if n:
                    names[e.name] = (n, e.numel, float(d[e.off:e.off + e.numel].max()))
            print(f'trial {trial} arena {k}: {int(bad.sum())} elements differ, max {float(d.max()):.3e}:', names, flush=True)
        else:
            print(f'trial {trial} arena {k}: equal (max {float(d.max()):.2e})', flush=True)
