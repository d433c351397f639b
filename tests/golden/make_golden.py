"""Generate the golden vectors of the training step FROM THE REFERENCE'S OWN CODE (container only).

Runs the reference's EncoderDecoder.forward_train / _parse_losses / backward / torch.optim.SGD (hot-path files
imported by path under oracle/ref_harness.py) on the repo's deterministic tiny fixtures, checks that the CPU
oracle (oracle/model.py) reproduces every number, and writes tests/golden/step_*.npz.  Inputs and weights are
regenerated from seeds on the GPU box (tests/common.py), so only outputs are stored.

Usage (build container):  python tests/golden/make_golden.py
"""
import copy
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
from oracle import ops as O  # noqa: E402
from oracle import ref_harness as RH  # noqa: E402
from tests import common as C  # noqa: E402

SCENARIOS = {
    # name: (model flags, n_sup, n_unsup, base lr, teacher conv_seg gain)
    'sup': (dict(unsup_weight=0), 2, 0, 0.001, 1.0),
    'mt_literal': (dict(unsup_weight=1.0), 2, 2, 0.01, 60.0),
    'mt_pasa': (dict(unsup_weight=1.0, attn_mask_seperate_head=True, attn_mask_weight=5, adaptive_attn_mask=True), 2, 2,
                0.001, 60.0),
    # configs/setr/..._MT_w_ours.py:236-256 on the tiny model (PatchMix_N = 2: 32-pixel blocks of the 64-pixel crops)
    'mt_ours': (dict(unsup_weight=1.0, attn_mask_seperate_head=True, attn_mask_weight=5, adaptive_attn_mask=True,
                     use_PatchShuffle_w_Cutmix=True, PatchMix_N=2, negative_class_ranking=True,
                     negative_class_ranking_mode='unsup_only'), 2, 2, 0.001, 60.0),
}
SEED_W, SEED_B = 1999, 2024
RELU_MARGIN = 2e-6     # iteration 0 of a fixture must not decide any ReLU by less than this (see relu_margin below)


def relu_margin(ref, fwd):
    """smallest |BatchNorm output| the reference feeds into a ReLU during one forward_train.  ReLU' is discontinuous at 0: an
    implementation whose pre-activation differs in the last bits takes the other branch there and legitimately returns a
    different gradient (one pixel's term; measured: 5e-3 of a BN bias gradient's maximum).  Like tied arg-max pixels, such
    elements carry no parity information, so the fixtures' batch seeds are chosen such that the reference decides none of
    its ReLUs by less than RELU_MARGIN at iteration 0 (its pre-activations are O(1), fp32 rounding ~1e-7)."""
    lo = [float('inf')]
    hooks = [m.register_forward_hook(lambda mod, i, o: lo.__setitem__(0, min(lo[0], float(o.detach().abs().min()))))
             for m in ref.modules() if isinstance(m, torch.nn.BatchNorm2d) and m.training]
    try:
        with torch.no_grad():
            fwd()
    finally:
        for h in hooks:
            h.remove()
    return lo[0]


NS = 64


def grad_sample(g, ns=NS):
    """ns elements of the flattened (logical layout) tensor at a fixed stride - the same rule on the GPU side"""
    f = g.detach().reshape(-1)
    step = max(1, f.numel() // ns)
    return f[::step][:ns].clone()


def fp64_grad_samples(cfg, seed_w, gain, batch, ns=NS, rng_seed=0):
    """iteration 0 of the reference evaluated in float64 (same float32 weights and inputs, widened): what the reference's own
    fp32 gradients are a rounding of.  Returns {name: (samples float64, max |g|)}."""
    imgs, gt, metas = batch
    ref = RH.build_reference_segmentor(cfg)
    ref.train()
    C.load_filled(ref, seed_w, gain)
    ref.double()
    torch.set_default_dtype(torch.float64)            # the reference builds its masks / constants in the default dtype
    cwd = os.getcwd()
    try:
        with tempfile.TemporaryDirectory() as td:
            os.chdir(td)
            try:
                seed_host_rng(rng_seed)
                losses = ref.forward_train(imgs.double(), copy.deepcopy(metas), gt_semantic_seg=gt, iter=0)
            finally:
                os.chdir(cwd)
        sum(v.mean() for k, v in losses.items() if 'loss' in k).backward()
    finally:
        torch.set_default_dtype(torch.float32)
    return {n: (grad_sample(p.grad, ns).numpy().astype(np.float64), float(p.grad.abs().max()))
            for n, p in ref.named_parameters() if p.grad is not None}


def seed_host_rng(seed):
    """the in-model augmentations draw from numpy's and torch's GLOBAL generators (generate_unsup_data.py): every iteration
    of every implementation starts from the same state"""
    np.random.seed(seed)
    torch.manual_seed(seed)


def run_steps(model, fwd, params_named, opt, set_lr, batches, n_iters=2, seed_b=0):
    rec = []
    for it in range(n_iters):
        imgs, gt, metas = batches[it]
        set_lr(opt, it)
        opt.zero_grad()
        seed_host_rng(seed_b + it)
        losses = fwd(imgs, gt, metas, it)
        loss = sum(v.mean() for k, v in losses.items() if 'loss' in k)
        loss.backward()
        r = dict(losses={k: float(v.mean()) for k, v in losses.items() if isinstance(v, torch.Tensor)}, loss=float(loss))
        r['grad_norms'] = {n: float(p.grad.norm()) for n, p in params_named() if p.grad is not None}
        # element samples of EVERY parameter's gradient (64 strided elements, logical = torch layout) + the tensor's max |g|
        r['grad_samples'] = {n: grad_sample(p.grad) for n, p in params_named() if p.grad is not None}
        r['grad_max'] = {n: float(p.grad.abs().max()) for n, p in params_named() if p.grad is not None}
        opt.step()
        rec.append(r)
    return rec


def main():
    assert RH.available(), 'reference tree needed'
    torch.set_num_threads(8)
    torch.manual_seed(0)
    for name, (flags, n_sup, n_unsup, lr, gain) in SCENARIOS.items():
        cfg = C.tiny_model_cfg(**flags)
        # ---------------- batch seed: the first one at which the reference takes no ReLU decision by less than RELU_MARGIN
        seed_b = SEED_B
        while True:
            probe = RH.build_reference_segmentor(cfg)
            probe.train()
            C.load_filled(probe, SEED_W, gain)
            b0 = C.make_batch(seed_b, n_sup, n_unsup)
            if flags.get('use_PatchShuffle_w_Cutmix'):
                # the fixture must exercise both augmentations at iteration 0
                seed_host_rng(seed_b)
                boxes, perms = O.draw_strong_aug(n_unsup, (64, 64), 0.5, 2, 0.5, 16 * flags['PatchMix_N'])
                if not any(b[1] > b[0] for b in boxes) or not any(p.tolist() != sorted(p.tolist()) for p in perms):
                    seed_b += 10
                    continue
            if flags.get('adaptive_attn_mask'):
                # ... nor cut the PASA top-k through tied patches (implementation-defined choice: make_golden_full.py)
                from tests.golden.make_golden_full import topk_boundary_tied
                n0 = n_sup + n_unsup
                tied = False
                for it in range(2):
                    bt = C.make_batch(seed_b + it, n_sup, n_unsup)
                    tied = tied or topk_boundary_tied(probe, bt[0][n0:], copy.deepcopy(bt[2][n0:]))
                if tied:
                    print(f'[{name}] batch seed {seed_b}: PASA top-k boundary tied', flush=True)
                    seed_b += 10
                    continue
            seed_host_rng(seed_b)
            cwd = os.getcwd()
            with tempfile.TemporaryDirectory() as td:
                os.chdir(td)
                try:
                    margin = relu_margin(probe, lambda: probe.forward_train(b0[0], copy.deepcopy(b0[2]), gt_semantic_seg=b0[1], iter=0))
                finally:
                    os.chdir(cwd)
            print(f'[{name}] batch seed {seed_b}: smallest |ReLU input| at iteration 0 = {margin:.2e}', flush=True)
            if margin >= RELU_MARGIN:
                break
            seed_b += 10
        batches = [C.make_batch(seed_b + it, n_sup, n_unsup) for it in range(2)]
        # ---------------- reference
        ref = RH.build_reference_segmentor(cfg)
        ref.train()
        vals = C.load_filled(ref, SEED_W, gain)
        ropt = OM.build_optimizer(ref, lr)
        cwd = os.getcwd()

        def ref_fwd(imgs, gt, metas, it):
            # (the reference writes PatchMix_N / PatchMixIndex INTO the img_metas dicts: every call gets its own copy, as every
            # iteration of a real run gets fresh metas from the data loader)
            return ref.forward_train(imgs, copy.deepcopy(metas), gt_semantic_seg=gt, iter=it)
        with tempfile.TemporaryDirectory() as td:
            os.chdir(td)
            try:
                rrec = run_steps(ref, ref_fwd, ref.named_parameters, ropt, OM.set_poly_lr, batches, seed_b=seed_b)
            finally:
                os.chdir(cwd)
        # ---------------- oracle
        orc = OM.oracle_from_cfg(cfg)
        orc.train()
        orc.load_state_dict(vals, strict=True)
        oopt = OM.build_optimizer(orc, lr)

        def orc_fwd(imgs, gt, metas, it):
            return orc.forward_train(imgs, [m['tag'] for m in metas], gt)
        orec = run_steps(orc, orc_fwd, orc.named_parameters, oopt, OM.set_poly_lr, batches, seed_b=seed_b)
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
                                        seed_w=SEED_W, seed_b=seed_b, relu_margin=margin, torch=torch.__version__,
                                        input_sha=[C.sha(b[0]) for b in batches],
                                        weight_sha=C.sha(torch.cat([v.flatten().float() for v in vals.values()])))))
        for it in range(2):
            out[f'it{it}_loss_keys'] = np.array(list(rrec[it]['losses'].keys()))
            out[f'it{it}_loss_vals'] = np.array(list(rrec[it]['losses'].values()), dtype=np.float64)
            out[f'it{it}_loss'] = np.float64(rrec[it]['loss'])
            out[f'it{it}_gn_keys'] = np.array(list(rrec[it]['grad_norms'].keys()))
            out[f'it{it}_gn_vals'] = np.array(list(rrec[it]['grad_norms'].values()), dtype=np.float64)
            # element samples in the order of gn_keys: [n_params, 64] (shorter tensors zero-padded) + max |g| per tensor
            gs = np.zeros((len(rrec[it]['grad_norms']), NS), dtype=np.float32)
            for i, n in enumerate(rrec[it]['grad_norms']):
                t = rrec[it]['grad_samples'][n].numpy()
                gs[i, :t.size] = t
            out[f'it{it}_gs'] = gs
            out[f'it{it}_gmax'] = np.array([rrec[it]['grad_max'][n] for n in rrec[it]['grad_norms']], dtype=np.float64)
        # the same iteration-0 gradients evaluated in float64: the tests bound the product's distance to THESE by a multiple of
        # the reference's own fp32 distance to them (its fp32 gradients are only good to ~5e-4 of a tensor's maximum)
        g64 = fp64_grad_samples(cfg, SEED_W, gain, batches[0], rng_seed=seed_b)
        gs64 = np.zeros((len(rrec[0]['grad_norms']), NS), dtype=np.float64)
        for i, n in enumerate(rrec[0]['grad_norms']):
            gs64[i, :g64[n][0].size] = g64[n][0]
        out['it0_gs64'] = gs64
        d = np.abs(out['it0_gs'].astype(np.float64) - gs64).max(axis=1) / out['it0_gmax']
        print(f'[{name}] reference fp32 vs its own fp64 evaluation, gradient elements / tensor max: median {np.median(d):.2e} '
              f'worst {d.max():.2e}', flush=True)
        out['final_sha_keys'] = np.array([k for k in rsd if rsd[k].dtype.is_floating_point])
        out['final_abs_sum'] = np.array([float(rsd[k].double().abs().sum()) for k in rsd if rsd[k].dtype.is_floating_point])
        if n_unsup:
            with torch.no_grad():
                ti = orc.last.get('teacher')
            # teacher pseudo-labels of the LAST iteration from the reference path itself
            ref.set_eval(True)
            with torch.no_grad():
                imgs, gt, metas = batches[1]
                tinfo = ref.extract_teacher_info_ema(imgs[n_sup + n_unsup:], copy.deepcopy(metas[n_sup + n_unsup:]))
            ref.set_train(True)
            lab = tinfo['hard_seg_label'].clone()
            lab[tinfo['conf_mask'] == 0] = 255
            out['teacher_label_final'] = lab.to(torch.uint8).numpy()
            out['teacher_mask_ratio_final'] = np.float64(tinfo['conf_mask'].float().mean())
            # how fragile each pixel's decision is in the reference's own arithmetic (encoder_decoder.py:888-901): the top-2
            # LOGIT margin (argmax flips only where it is ~0) and |p_max - threshold| (confidence flips only where it is ~0)
            z = tinfo['seg_logits']
            top2 = z.topk(2, dim=1).values
            out['teacher_margin_final'] = (top2[:, 0] - top2[:, 1]).numpy().astype(np.float32)
            out['teacher_pmax_final'] = torch.softmax(z, 1).max(1).values.numpy().astype(np.float32)
            out['teacher_logit_absmax_final'] = np.float64(z.abs().max())
            print(f'[{name}] mask_ratio {float(out["teacher_mask_ratio_final"]):.3f}', flush=True)
        np.savez_compressed(os.path.join(HERE, f'step_{name}.npz'), **out)
        print(f'[{name}] losses it0 {rrec[0]["losses"]}', flush=True)
    print('goldens written')


if __name__ == '__main__':
    main()
