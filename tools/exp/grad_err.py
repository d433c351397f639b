#!/usr/bin/env python
"""per-parameter distance of the product's iteration-0 gradients (fp32 or bf16 mode) to the golden samples made by the reference
(fp32) and to the reference's fp64 evaluation; usage: python tools/exp/grad_err.py [scenario] [dtype]"""
import json, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import s4former_amd as S
from tests import common as C
name = sys.argv[1] if len(sys.argv) > 1 else 'sup'
dtype = sys.argv[2] if len(sys.argv) > 2 else 'fp32'
full = name.startswith('full')
z = np.load(os.path.join(ROOT, 'tests', 'golden', (name if full else f'step_{name}') + '.npz'))
meta = json.loads(str(z['meta']))
S.set_compute_dtype(dtype)
cfg = C.deit_b_cfg(img=meta['img'], num_classes=meta['num_classes'], **meta['flags']) if full else C.tiny_model_cfg(**meta['flags'])
model = S.build_segmentor(cfg); model.train(); C.load_filled(model, meta['seed_w'], meta['gain']); model.cuda()
opt = S.build_optimizer(model, dict(type='SGD', lr=meta['lr'], momentum=0.9, weight_decay=0.0, paramwise_cfg=dict(custom_keys={'head': dict(lr_mult=10.)})))
bkw = dict(img=meta['img'], num_classes=meta['num_classes'], block=32, border=8) if full else {}
imgs, gt, metas = C.make_batch(meta['seed_b'], meta['n_sup'], meta['n_unsup'], **bkw)
opt.zero_grad()
out = model.train_step(dict(img=imgs.cuda(), img_metas=metas, gt_semantic_seg=gt.cuda()), opt, iter=0)
out['loss'].backward(); torch.cuda.synchronize()
named = dict(model.named_parameters())
keys = [str(k) for k in z['it0_gn_keys']]
ns = z['it0_gs'].shape[1]
rows = []
for i, k in enumerate(keys):
    got = C.grad_sample(named[k].grad, ns).double().cpu().numpy()
    r32 = z['it0_gs'][i, :got.size].astype(np.float64); r64 = z['it0_gs64'][i, :got.size]
    gm = float(z['it0_gmax'][i])
    rows.append((np.abs(got - r64).max() / gm, np.abs(got - r32).max() / gm, np.abs(r32 - r64).max() / gm,
                 abs(float(named[k].grad.norm()) - z['it0_gn_vals'][i]) / z['it0_gn_vals'][i], k, int(np.abs(got - r64).argmax())))
rows.sort(reverse=True)
print(f'{name} {dtype}: |prod-ref64|/max  |prod-ref32|/max  |ref32-ref64|/max  norm rel   name   worst sample index')
for r in rows[:25]:
    print('%.2e  %.2e  %.2e  %.2e  %s  %d' % r)
print('median', np.median([r[0] for r in rows]), np.median([r[2] for r in rows]))
