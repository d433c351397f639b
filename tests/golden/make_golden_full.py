"""FULL-SIZE golden vectors from the reference's own code (container only; ~10 minutes of CPU):

  full_sup    DeiT-B / SETR-PUP 512x512, 21 classes, 2 labelled images            (BASELINE cfg1 / cfg2 shapes), iterations 0-1
  full_pasa   the same model, 2 + 2 images, attn_mask_seperate_head + adaptive    (cfg3 / cfg4 shapes, the paper's PASA step), iterations 0-1
  full_768    768x768, 19 classes, N = 2305 tokens, 1 + 1 images, one iteration      (cfg5 shapes)
  full_ours   full_pasa + CutMix / PatchShuffle (128-pixel blocks) + negative-class ranking  (the paper's full method), iterations 0-1
  full_sup8   8 labelled images, one iteration (BASELINE cfg2 at its real batch; no fp64 evaluation)
  full_semi8  8 + 8 images with PASA, one iteration (BASELINE cfg3 / cfg4 per GPU at its real batch; no fp64 evaluation)
  full_semi8_fwd  8 + 8 images with PASA, forward only (round 4: what fits of cfg3 at its real batch)
  full_768_semi4_fwd  768x768, 19 classes, 4 + 4 images with PASA, forward only (round 5: cfg5 at its real per-GPU batch)

Per scenario: every named loss, the total, per-parameter gradient L2 norms, 32 strided gradient elements of EVERY parameter
(+ the tensor's max |g|), |.|_1 of every state-dict tensor after the optimiser steps (student, BN statistics, EMA teacher),
the teacher's pseudo-labels of the last iteration with the reference's own tie set (pixels whose top-2 logit margin or
distance of p_max to the threshold is below 1e-3 of max |logit|).  Weights and inputs are regenerated from seeds on the GPU
box (tests/common.py): only outputs are stored.

Usage (build container):  python tests/golden/make_golden_full.py [scenario ...]
"""
import copy
import json
import os
import sys
import tempfile
import time

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

from oracle import model as OM  # noqa: E402
from oracle import ref_harness as RH  # noqa: E402
from tests import common as C  # noqa: E402

PASA = dict(unsup_weight=1.0, attn_mask_seperate_head=True, attn_mask_weight=5, adaptive_attn_mask=True)
SCENARIOS = {
    # name: (img, classes, flags, n_sup, n_unsup, lr, iterations with backward)
    'full_sup': (512, 21, dict(unsup_weight=0), 2, 0, 0.001, 2),
    'full_pasa': (512, 21, PASA, 2, 2, 0.001, 2),
    'full_768': (768, 19, PASA, 1, 1, 0.001, 1),
    # configs/setr/..._MT_w_ours.py:236-256 at full size: PASA + CutMix / PatchShuffle (PatchMix_N = 8: 128-pixel blocks) + NCR
    'full_ours': (512, 21, dict(PASA, use_PatchShuffle_w_Cutmix=True, PatchMix_N=8, negative_class_ranking=True,
                                negative_class_ranking_mode='unsup_only'), 2, 2, 0.001, 2),
    # round 3: the benchmarked configurations at their REAL batch (head BatchNorm statistics over 8 images, not 2), one iteration
    # with its backward.  No fp64 evaluation (it does not fit the build container's memory): the gradient gate of these
    # fixtures takes the reference's fp32-vs-fp64 distance per tensor from the small-batch fixture of the same model.
    'full_sup8': (512, 21, dict(unsup_weight=0), 8, 0, 0.001, 1),           # BASELINE cfg2
    'full_semi8': (512, 21, PASA, 8, 8, 0.001, 1),                          # BASELINE cfg3 / cfg4 per GPU
    'full_semi4': (512, 21, PASA, 4, 4, 0.001, 1),                          # fallback if 8 + 8 does not fit the container
    # round 4: the BENCHMARKED configuration (cfg3: 8 + 8 with PASA) at its real batch, FORWARD ONLY under torch.no_grad() - the
    # step with its backward needs ~67 GB in a 62 GB container; the forward pins all seven losses, mask_ratio and the teacher's
    # labels (+ tie set) at the batch the bench line is measured on
    'full_semi8_fwd': (512, 21, PASA, 8, 8, 0.001, 0),
    # round 5: cfg5 (Cityscapes crops: 768 x 768, 19 classes, 2305 tokens) at its REAL per-GPU batch of 4 + 4, forward only
    # (full_768 is 1 + 1 with its backward)
    'full_768_semi4_fwd': (768, 19, PASA, 4, 4, 0.001, 0),
}
NO_FP64 = {'full_sup8', 'full_semi8', 'full_semi4', 'full_semi8_fwd', 'full_768_semi4_fwd'}
SEED_W, SEED_B, NS = 1999, 3030, 32
FRAG = 1e-3          # the stored tie set covers every logit bound up to FRAG * max |logit|


def teacher_labels(ref, imgs, metas, th):
    ref.set_eval(True)
    with torch.no_grad():
        t = ref.extract_teacher_info_ema(imgs, metas)
    ref.set_train(True)
    lab = t['hard_seg_label'].clone()
    lab[t['conf_mask'] == 0] = 255
    z = t['seg_logits']
    top2 = z.topk(2, dim=1).values
    margin = (top2[:, 0] - top2[:, 1]).reshape(-1)
    pmax = torch.softmax(z, 1).max(1).values.reshape(-1)
    amax = float(z.abs().max())
    frag = ((margin < FRAG * amax) | ((pmax - th).abs() < 0.5 * FRAG * amax + 1e-7)).nonzero().reshape(-1)
    return dict(label=lab.to(torch.uint8).numpy(), ratio=float(t['conf_mask'].float().mean()), absmax=amax,
                frag_idx=frag.to(torch.int32).numpy(), frag_margin=margin[frag].numpy().astype(np.float32),
                frag_pmax=pmax[frag].numpy().astype(np.float32))


def topk_boundary_tied(ref, imgs, metas):
    """does the PASA selection (vit.py:519-535: topk of the per-patch unconfidence, largest=False, k = half the patches) cut
    through a group of EQUAL values for any image?  Which of the tied patches torch.topk returns is implementation-defined
    (CPU: libstdc++ nth_element, GPU: radix select), so a fixture with such a tie cannot pin the flagged rows."""
    ref.set_eval(True)
    with torch.no_grad():
        t = ref.extract_teacher_info_ema(imgs, metas)
    ref.set_train(True)
    conf = t['conf_mask'].float()
    B, H, W = conf.shape
    u = (1 - conf).view(B, H // 16, 16, W // 16, 16).sum((2, 4)).reshape(B, -1) / 256.0
    k = u.shape[1] // 2
    s = torch.sort(u, dim=1).values
    return bool((s[:, k - 1] == s[:, k]).any())


def calibrate_gain(ref, base_w, imgs, metas, target=0.5):
    """teacher conv_seg gain (bisection on the reference's own teacher pass) so that about half the pixels are confident"""
    lo, hi, gain = 1.0, 1e4, 1.0
    w = ref.decode_head_ema.conv_seg.weight
    for _ in range(12):
        gain = (lo * hi) ** 0.5
        with torch.no_grad():
            w.copy_(base_w * gain)
        r = teacher_labels(ref, imgs, metas, ref.unsup_confidence)['ratio']
        print(f'    gain {gain:9.2f} -> mask_ratio {r:.3f}', flush=True)
        if abs(r - target) < 0.08:
            break
        lo, hi = (gain, hi) if r < target else (lo, gain)
    return gain


def main():
    assert RH.available(), 'reference tree needed'
    torch.set_num_threads(8)
    names = sys.argv[1:] or list(SCENARIOS)
    for name in names:
        img, ncls, flags, n_sup, n_unsup, lr, iters = SCENARIOS[name]
        t0 = time.time()
        cfg = C.deit_b_cfg(img=img, num_classes=ncls, **flags)
        nb = max(iters, 1)
        seed_b = SEED_B
        batches = [C.make_batch(seed_b + it, n_sup, n_unsup, img=img, num_classes=ncls, block=32, border=8) for it in range(nb)]
        ref = RH.build_reference_segmentor(cfg)
        ref.train()
        vals = C.load_filled(ref, SEED_W, 1.0)
        gain = 1.0
        if n_unsup:
            n0 = n_sup + n_unsup
            gain = calibrate_gain(ref, vals['decode_head_ema.conv_seg.weight'].clone(), batches[0][0][n0:], batches[0][2][n0:])
            gain = float(np.float32(gain))
            vals = C.load_filled(ref, SEED_W, gain)          # exactly what the GPU side will build from (seed, gain)
            # At 1024 / 2304 patches with confidences in multiples of 1/256 the PASA top-k boundary is tied for almost every batch
            # (a tie-free seed for one iteration took 65 tries and did not survive the EMA step of the next): the full-size
            # fixtures are NOT tie-free; the fp32 test takes the reference CPU path's choice among tied patches
            # (S4F_TOPK_TIES=cpu) when it compares with them.
            print(f'[{name}] PASA top-k boundary tied at the first teacher pass: '
                  f'{[topk_boundary_tied(ref, b[0][n0:], b[2][n0:]) for b in batches]}', flush=True)
        opt = OM.build_optimizer(ref, lr)
        out = {}
        cwd = os.getcwd()
        with tempfile.TemporaryDirectory() as td:
            os.chdir(td)
            try:
                for it in range(nb):
                    imgs, gt, metas = batches[it]
                    OM.set_poly_lr(opt, it)
                    opt.zero_grad()
                    if flags.get('use_PatchShuffle_w_Cutmix'):
                        C.seed_host_rng(seed_b + it)      # the in-model augmentations draw from the global numpy / torch generators
                    with torch.set_grad_enabled(iters > 0):
                        losses = ref.forward_train(imgs, copy.deepcopy(metas), gt_semantic_seg=gt, iter=it)

                    loss = sum(v.mean() for k, v in losses.items() if 'loss' in k)
                    lk = [k for k, v in losses.items() if isinstance(v, torch.Tensor)]
                    out[f'it{it}_loss_keys'] = np.array(lk)
                    out[f'it{it}_loss_vals'] = np.array([float(losses[k].mean()) for k in lk], dtype=np.float64)
                    out[f'it{it}_loss'] = np.float64(float(loss))
                    print(f'[{name}] it{it} {dict(zip(lk, out[f"it{it}_loss_vals"].round(5)))} ({time.time() - t0:.0f} s)', flush=True)
                    if iters == 0:
                        break
                    loss.backward()
                    named = [(n, p) for n, p in ref.named_parameters() if p.grad is not None]
                    out[f'it{it}_gn_keys'] = np.array([n for n, _ in named])
                    out[f'it{it}_gn_vals'] = np.array([float(p.grad.norm()) for _, p in named], dtype=np.float64)
                    gs = np.zeros((len(named), NS), dtype=np.float32)
                    for i, (_, p) in enumerate(named):
                        t = C.grad_sample(p.grad, NS).numpy()
                        gs[i, :t.size] = t
                    out[f'it{it}_gs'] = gs
                    out[f'it{it}_gmax'] = np.array([float(p.grad.abs().max()) for _, p in named], dtype=np.float64)
                    opt.step()
            finally:
                os.chdir(cwd)
        if iters and name not in NO_FP64:
            from tests.golden.make_golden import fp64_grad_samples
            keys0 = [str(k) for k in out['it0_gn_keys']]
            del opt
            g64 = fp64_grad_samples(cfg, SEED_W, gain, batches[0], NS, rng_seed=seed_b)
            gs64 = np.zeros((len(keys0), NS), dtype=np.float64)
            for i, n in enumerate(keys0):
                gs64[i, :g64[n][0].size] = g64[n][0]
            out['it0_gs64'] = gs64
            d = np.abs(out['it0_gs'].astype(np.float64) - gs64).max(axis=1) / out['it0_gmax']
            print(f'[{name}] reference fp32 vs its own fp64 evaluation, gradient elements / tensor max: median {np.median(d):.2e} '
                  f'worst {d.max():.2e} ({time.time() - t0:.0f} s)', flush=True)
            opt = None
        rsd = ref.state_dict()
        fk = [k for k in rsd if rsd[k].dtype.is_floating_point]
        out['final_sha_keys'] = np.array(fk)
        out['final_abs_sum'] = np.array([float(rsd[k].double().abs().sum()) for k in fk])
        if n_unsup:
            imgs, gt, metas = batches[nb - 1]
            n0 = n_sup + n_unsup
            t = teacher_labels(ref, imgs[n0:], metas[n0:], ref.unsup_confidence)
            out['teacher_label_final'] = t['label']
            out['teacher_mask_ratio_final'] = np.float64(t['ratio'])
            out['teacher_logit_absmax_final'] = np.float64(t['absmax'])
            out['teacher_frag_idx'], out['teacher_frag_margin'], out['teacher_frag_pmax'] = t['frag_idx'], t['frag_margin'], t['frag_pmax']
            print(f'[{name}] mask_ratio {t["ratio"]:.3f}, tie set {t["frag_idx"].size} of {t["label"].size} pixels', flush=True)
        out['meta'] = json.dumps(dict(scenario=name, img=img, num_classes=ncls, flags=flags, n_sup=n_sup, n_unsup=n_unsup, lr=lr,
                                      gain=gain, seed_w=SEED_W, seed_b=seed_b, ns=NS, frag=FRAG, iters=iters, torch=torch.__version__,
                                      input_sha=[C.sha(b[0]) for b in batches],
                                      weight_sha=C.sha(torch.cat([vals[k].flatten().float() for k in list(vals)[:40]]))))
        np.savez_compressed(os.path.join(HERE, f'{name}.npz'), **out)
        print(f'[{name}] written ({time.time() - t0:.0f} s)', flush=True)
        del ref, opt, vals


if __name__ == '__main__':
    main()
