"""When do the head calls execute on their streams, and when did the host issue them?  HIP events around every head call of one
step (two per call: no tracer, no per-kernel profiler), against host time stamps.  python tools/exp/head_marks.py"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import s4former_amd as S  # noqa: E402
import s4former_amd.functional as F_  # noqa: E402
from s4former_amd.dist import GradReducer  # noqa: E402
from s4former_amd.functional import join_side_streams  # noqa: E402
from s4former_amd.presets import MAX_ITERS, OPTIMIZER, setr_pup_model, synthetic_batch  # noqa: E402
import bench  # noqa: E402

dev = torch.device('cuda', 0)
S.set_compute_dtype('bf16')
torch.manual_seed(1999)
n_sup, n_unsup, img, ncls, flags, desc = bench.WORKLOADS['semi']
model = S.build_segmentor(setr_pup_model(img=img, num_classes=ncls, **flags))
model.init_weights(); model.train(); model.to(dev)
model.log_vars_as_tensors = True
opt = S.build_optimizer(model, dict(OPTIMIZER))
opt.fused_zero_grad = True
sched = S.PolyLR(opt, MAX_ITERS)
reducer = GradReducer()
batches = [synthetic_batch(1999 + i, n_sup, n_unsup, img=img, num_classes=ncls, device=dev) for i in range(2)]
model.ensure_engine(dev)
reducer.attach(model.student_store)
seg_gain, _ = bench.calibrate_teacher(model, batches[0], n_sup, n_unsup, 0.4)


def step(it):
    imgs, gt, metas = batches[it % 2]
    sched.step(it)
    opt.zero_grad()
    out = model.train_step(dict(img=imgs, img_metas=metas, gt_semantic_seg=gt), opt, iter=it)
    out['loss'].backward()
    join_side_streams()
    opt.step()


for it in range(8):
    step(it)
torch.cuda.synchronize()
for rep in range(2):
    F_.HEAD_MARKS = []
    e0 = torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    e0.record()
    step(8 + rep)
    e1 = torch.cuda.Event(enable_timing=True)
    e1.record()
    torch.cuda.synchronize()
    marks, F_.HEAD_MARKS = F_.HEAD_MARKS, None
    print(f'--- step: {e0.elapsed_time(e1):.2f} ms on the GPU; head calls (label, stream): GPU start - end | host issue start - end  (ms from step start)')
    streams = {}
    for label, st, a, b, h0, h1 in marks:
        sid = streams.setdefault(st, chr(ord('b') + len(streams)))
        print(f'  {label[0]} convs={label[1]} imgs={label[2]} on {sid}: GPU {e0.elapsed_time(a):6.2f} - {e0.elapsed_time(b):6.2f}  ({a.elapsed_time(b):5.2f}) | '
              f'host {(h0 - t0) * 1e3:6.2f} - {(h1 - t0) * 1e3:6.2f}')
