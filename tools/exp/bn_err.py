#!/usr/bin/env python
"""which channels carry the error of a BN bias gradient? product (GPU, fp32 mode) vs oracle (CPU fp32 / fp64) on the tiny 'sup' scenario"""
import json, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import s4former_amd as S
from oracle import model as OM
from tests import common as C
z = np.load(os.path.join(ROOT, 'tests', 'golden', 'step_sup.npz')); meta = json.loads(str(z['meta']))
cfg = C.tiny_model_cfg(**meta['flags'])
imgs, gt, metas = C.make_batch(meta['seed_b'], meta['n_sup'], meta['n_unsup'])
def oracle(dbl):
    orc = OM.oracle_from_cfg(cfg); orc.train()
    orc.load_state_dict(C.fill_state([(k, tuple(v.shape)) for k, v in orc.state_dict().items()], meta['seed_w'], meta['gain']))
    x = imgs
    if dbl:
        orc.double(); x = imgs.double()
    acts = {}
    def hook(name):
        def f(m, i, o): acts[name] = (i[0].detach(), o.detach())
        return f
    for n, m in orc.named_modules():
        if isinstance(m, torch.nn.BatchNorm2d): m.register_forward_hook(hook(n))
    loss, _ = orc.parse_losses(orc.forward_train(x, [m['tag'] for m in metas], gt)); loss.backward()
    return {n: p.grad.double() for n, p in orc.named_parameters() if p.grad is not None}, acts
g32, acts = oracle(False); g64, _ = oracle(True)
S.set_compute_dtype('fp32')
model = S.build_segmentor(cfg); model.train(); C.load_filled(model, meta['seed_w'], meta['gain']); model.cuda()
opt = S.build_optimizer(model, dict(type='SGD', lr=meta['lr'], momentum=0.9, weight_decay=0.0, paramwise_cfg=dict(custom_keys={'head': dict(lr_mult=10.)})))
opt.zero_grad()
out = model.train_step(dict(img=imgs.cuda(), img_metas=metas, gt_semantic_seg=gt.cuda()), opt, iter=0)
out['loss'].backward(); torch.cuda.synchronize()
gp = {n: p.grad.double().cpu() for n, p in model.named_parameters() if p.grad is not None}
print('oracle32 vs oracle64 worst/max:', max(float((g32[k]-g64[k]).abs().max()/g64[k].abs().max()) for k in g32))
for k in sorted(gp, key=lambda k: -float((gp[k]-g64[k]).abs().max()/g64[k].abs().max()))[:6]:
    e = (gp[k]-g64[k]).abs(); print(k, 'err/max %.2e' % float(e.max()/g64[k].abs().max()), 'argmax', int(e.reshape(-1).argmax()), tuple(gp[k].shape))
k = 'auxiliary_head.2.up_convs.1.0.bn.bias'
e = (gp[k]-g64[k]).abs()/g64[k].abs().max()
bn = [n for n in acts if n.startswith('auxiliary_head.2.up_convs.1')][0]
xin, yout = acts[bn]
mu = xin.mean((0,2,3)); sd = xin.std((0,2,3)); act = (yout > 0).float().mean((0,2,3))
print('channel  err/max   dbeta64     mean      std     relu-active')
for c in e.argsort(descending=True)[:10].tolist():
    print(c, '%.2e %.3e %.3e %.3e %.3f' % (float(e[c]), float(g64[k][c]), float(mu[c]), float(sd[c]), float(act[c])))
print('median err', float(e.median()))
