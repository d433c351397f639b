"""Do two independent B = 8 backbone forward chains on two streams share the chip better than one after the other?  (round 5: the
teacher pass beside the student's labelled half)   python tools/exp/two_chain_backbone_probe.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import s4former_amd as S  # noqa: E402
from s4former_amd.presets import setr_pup_model  # noqa: E402

S.set_compute_dtype('bf16')
torch.manual_seed(0)
model = S.build_segmentor(setr_pup_model(img=512, num_classes=21, unsup_weight=1.0, plain_mt_pseudo_loss=True))
model.init_weights()
model.train()
model.cuda()
model.ensure_engine(torch.device('cuda'))
x8a = torch.randn(8, 3, 512, 512, device='cuda')
x8b = torch.randn(8, 3, 512, 512, device='cuda')
x16 = torch.cat([x8a, x8b])
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()


def timeit(fn, iters=5):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


def one8():
    with torch.no_grad():
        model.extract_feat_ema(x8a)


def one16():
    with torch.no_grad():
        model.extract_feat_ema(x16)


def two8():
    cur = torch.cuda.current_stream()
    s1.wait_stream(cur); s2.wait_stream(cur)
    with torch.no_grad():
        with torch.cuda.stream(s1):
            model.extract_feat_ema(x8a)
        with torch.cuda.stream(s2):
            model.extract_feat_ema(x8b)
    cur.wait_stream(s1); cur.wait_stream(s2)


def seq8():
    with torch.no_grad():
        model.extract_feat_ema(x8a)
        model.extract_feat_ema(x8b)


print(f'backbone forward (no grad, bf16): one B=8 chain {timeit(one8):.3f} ms | one B=16 chain {timeit(one16):.3f} ms | '
      f'two B=8 chains on two streams {timeit(two8):.3f} ms | two B=8 chains back to back {timeit(seq8):.3f} ms', flush=True)
