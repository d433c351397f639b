"""__graft_entry__.smoke(): one tiny S4Former training step (student fwd/bwd, EMA, teacher pseudo-labels with PASA,
fused SGD) through the C-ABI HIP kernels on cuda:0, checked against the CPU oracle."""
import torch

import s4former_amd as S
from oracle import model as OM
from tests import common as C


def run_smoke(verbose=True):
    flags = dict(unsup_weight=1.0, attn_mask_seperate_head=True, attn_mask_weight=5, adaptive_attn_mask=True)
    cfg = C.tiny_model_cfg(**flags)
    S.set_compute_dtype('fp32')
    model = S.build_segmentor(cfg)
    model.train()
    vals = C.load_filled(model, 1999, 60.0)
    model.cuda()
    opt = S.build_optimizer(model, dict(type='SGD', lr=0.001, momentum=0.9, weight_decay=0.0,
                                        paramwise_cfg=dict(custom_keys={'head': dict(lr_mult=10.)})))
    imgs, gt, metas = C.make_batch(2024, 2, 2)
    opt.zero_grad()
    out = model.train_step(dict(img=imgs.cuda(), img_metas=metas, gt_semantic_seg=gt.cuda()), opt, iter=0)
    out['loss'].backward()
    opt.step()
    torch.cuda.synchronize()

    orc = OM.oracle_from_cfg(cfg)
    orc.train()
    orc.load_state_dict(vals, strict=True)
    loss, logv = orc.parse_losses(orc.forward_train(imgs, [m['tag'] for m in metas], gt))
    for k, v in logv.items():
        got, ref = out['log_vars'][k], float(v)
        assert abs(got - ref) <= 1e-4 * abs(ref), f'smoke: {k} = {got} vs oracle {ref}'
    assert abs(out['log_vars']['loss'] - float(loss)) <= 1e-4 * abs(float(loss))
    if verbose:
        print('smoke ok:', {k: round(v, 5) for k, v in out['log_vars'].items()})
    S.set_compute_dtype('bf16')
