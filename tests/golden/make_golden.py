"""Generate the golden vectors of the training step FROM THE REFERENCE'S OWN CODE (container only).

Runs the reference's EncoderDecoder.forward_train / _parse_losses / backward / torch.optim.SGD (hot-path files
imported by path under oracle/ref_harness.py) on the repo's deterministic tiny fixtures, checks that the CPU
oracle (oracle/model.py) reproduces every number, and writes tests/golden/step_*.npz.  Inputs and weights are
regenerated from seeds on the GPU box (tests/common.py), so only outputs are stored.

Usage (build container):  python tests/golden/make_golden.py
"""
import json
import os
import sys
import tempfile

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

from oracle import model as OM  # noqa: E402
from oracle import ref_harness as RH  # noqa: E402
from tests import common as C  # noqa: E402

SCENARIOS = {
    # name: (model flags, n_sup, n_unsup, base lr, teacher conv_seg gain)
    'sup': (dict(unsup_weight=0), 2, 0, 0.001, 1.0),
    'mt_literal': (dict(unsup_weight=1.0), 2, 2, 0.01, 60.0),
    'mt_pasa': (dict(unsup_weight=1.0, attn_mask_seperate_head=True, attn_mask_weight=5, adaptive_attn_mask=True), 2, 2,
                0.001, 60.0),
}
SEED_W, SEED_B = 1999, 2024


def run_steps(model, fwd, params_named, opt, set_lr, batches, n_iters=2):
    rec = []
    for it in range(n_iters):
        imgs, gt, metas = batches[it]
        set_lr(opt, it)
        opt.zero_grad()
        losses = fwd(imgs, gt, metas, it)
        loss = sum(v.mean() for k, v in losses.items() if 'loss' in k)
        loss.backward()
        r = dict(losses={k: float(v.mean()) for k, v in losses.items() if isinstance(v, torch.Tensor)}, loss=float(loss))
        r['grad_norms'] = {n: float(p.grad.norm()) for n, p in params_named() if p.grad is not None}
        r['grad_samples'] = {n: p.grad.flatten()[:: max(1, p.grad.numel() // 16)][:16].clone()
                             for n, p in params_named() if p.grad is not None and n.endswith(('conv_seg.weight', 'cls_token', 'ln1.weight'))}
        opt.step()
        rec.append(r)
    return rec


def main():
    assert RH.available(), 'reference tree needed'
    torch.set_num_threads(8)
    torch.manual_seed(0)
    for name, (flags, n_sup, n_unsup, lr, gain) in SCENARIOS.items():
        cfg = C.tiny_model_cfg(**flags)
        batches = [C.make_batch(SEED_B + it, n_sup, n_unsup) for it in range(2)]
        # ---------------- reference
        ref = RH.build_reference_segmentor(cfg)
        ref.train()
        vals = C.load_filled(ref, SEED_W, gain)
        ropt = OM.build_optimizer(ref, lr)
        cwd = os.getcwd()

        def ref_fwd(imgs, gt, metas, it):
            return ref.forward_train(imgs, metas, gt_semantic_seg=gt, iter=it)
        with tempfile.TemporaryDirectory() as td:
            os.chdir(td)
            try:
                rrec = run_steps(ref, ref_fwd, ref.named_parameters, ropt, OM.set_poly_lr, batches)
            finally:
                os.chdir(cwd)
        # ---------------- oracle
        orc = OM.oracle_from_cfg(cfg)
        orc.train()
        orc.load_state_dict(vals, strict=True)
        oopt = OM.build_optimizer(orc, lr)

        def orc_fwd(imgs, gt, metas, it):
            return orc.forward_train(imgs, [m['tag'] for m in metas], gt)
        orec = run_steps(orc, orc_fwd, orc.named_parameters, oopt, OM.set_poly_lr, batches)
        # ---------------- compare oracle vs reference
        # iteration 0 must agree to fp32 rounding; iteration 1 sits behind an SGD step (head lr up to 0.1) that
        # amplifies summation-order differences, so it gets a looser band.
        worst = [0.0, 0.0]
        for it in range(2):
            assert set(k for k in rrec[it]['losses'] if 'loss' in k) == set(k for k in orec[it]['losses'] if 'loss' in k), \
                (name, rrec[it]['losses'].keys(), orec[it]['losses'].keys())
            for k, v in rrec[it]['losses'].items():
                if 'loss' in k:
                    worst[it] = max(worst[it], abs(v - orec[it]['losses'][k]) / (abs(v) + 1e-12))
            for k, v in rrec[it]['grad_norms'].items():
                worst[it] = max(worst[it], abs(v - orec[it]['grad_norms'][k]) / (abs(v) + 1e-12))
        rsd, osd = ref.state_dict(), orc.state_dict()
        wsd = 0.0
        for k in rsd:
            if rsd[k].dtype.is_floating_point:
                wsd = max(wsd, float((rsd[k] - osd[k]).abs().max() / (rsd[k].abs().max() + 1e-12)))
        print(f'[{name}] oracle vs reference: worst relative deviation it0 {worst[0]:.3e} it1 {worst[1]:.3e} '
              f'final weights {wsd:.3e}', flush=True)
        assert worst[0] < 5e-6 and worst[1] < 1e-3 and wsd < 1e-4, f'oracle deviates from the reference in scenario {name}'
        # ---------------- save golden
        out = dict(meta=json.dumps(dict(scenario=name, flags=flags, n_sup=n_sup, n_unsup=n_unsup, lr=lr, gain=gain,
                                        seed_w=SEED_W, seed_b=SEED_B, torch=torch.__version__,
                                        input_sha=[C.sha(b[0]) for b in batches],
                                        weight_sha=C.sha(torch.cat([v.flatten().float() for v in vals.values()])))))
        for it in range(2):
            out[f'it{it}_loss_keys'] = np.array(list(rrec[it]['losses'].keys()))
            out[f'it{it}_loss_vals'] = np.array(list(rrec[it]['losses'].values()), dtype=np.float64)
            out[f'it{it}_loss'] = np.float64(rrec[it]['loss'])
            out[f'it{it}_gn_keys'] = np.array(list(rrec[it]['grad_norms'].keys()))
            out[f'it{it}_gn_vals'] = np.array(list(rrec[it]['grad_norms'].values()), dtype=np.float64)
            for n, t in rrec[it]['grad_samples'].items():
                out[f'it{it}_gs_{n}'] = t.numpy()
        out['final_sha_keys'] = np.array([k for k in rsd if rsd[k].dtype.is_floating_point])
        out['final_abs_sum'] = np.array([float(rsd[k].double().abs().sum()) for k in rsd if rsd[k].dtype.is_floating_point])
        if n_unsup:
            with torch.no_grad():
                ti = orc.last.get('teacher')
            # teacher pseudo-labels of the LAST iteration from the reference path itself
            ref.set_eval(True)
            with torch.no_grad():
                imgs, gt, metas = batches[1]
                tinfo = ref.extract_teacher_info_ema(imgs[n_sup + n_unsup:], metas[n_sup + n_unsup:])
            ref.set_train(True)
            lab = tinfo['hard_seg_label'].clone()
            lab[tinfo['conf_mask'] == 0] = 255
            out['teacher_label_final'] = lab.to(torch.uint8).numpy()
            out['teacher_mask_ratio_final'] = np.float64(tinfo['conf_mask'].float().mean())
            print(f'[{name}] mask_ratio {float(out["teacher_mask_ratio_final"]):.3f}', flush=True)
        np.savez_compressed(os.path.join(HERE, f'step_{name}.npz'), **out)
        print(f'[{name}] losses it0 {rrec[0]["losses"]}', flush=True)
    print('goldens written')


if __name__ == '__main__':
    main()
