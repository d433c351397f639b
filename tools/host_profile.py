"""cProfile of the host side of the training step (enqueue only): python tools/host_profile.py [steps]"""
import cProfile
import os
import pstats
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import s4former_amd as S  # noqa: E402
from s4former_amd.dist import GradReducer  # noqa: E402
from s4former_amd.functional import join_side_streams  # noqa: E402
from s4former_amd.presets import MAX_ITERS, OPTIMIZER, setr_pup_model, synthetic_batch  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 5
dev = torch.device('cuda', 0)
S.set_compute_dtype('bf16')
torch.manual_seed(1999)
import bench  # noqa: E402
n_sup, n_unsup, img, ncls, flags, desc = bench.WORKLOADS['semi']
model = S.build_segmentor(setr_pup_model(img=img, num_classes=ncls, **flags))
model.init_weights(); model.train(); model.to(dev)
model.log_vars_as_tensors = True
opt = S.build_optimizer(model, dict(OPTIMIZER))
sched = S.PolyLR(opt, MAX_ITERS)
reducer = GradReducer()
batches = [synthetic_batch(1999 + i, n_sup, n_unsup, img=img, num_classes=ncls, device=dev) for i in range(2)]
model.ensure_engine(dev)
reducer.attach(model.student_store)


def step(it):
    imgs, gt, metas = batches[it % 2]
    sched.step(it)
    opt.zero_grad()
    out = model.train_step(dict(img=imgs, img_metas=metas, gt_semantic_seg=gt), opt, iter=it)
    out['loss'].backward()
    join_side_streams()
    reducer.reduce_(model.student_store.grad)
    reducer.wait()
    opt.step(grad_scale=reducer.grad_scale())
    return out


it = 0
for _ in range(4):
    step(it); it += 1
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(steps):
    step(it); it += 1
host = time.perf_counter() - t0
torch.cuda.synchronize()
print(f'host enqueue {1e3 * host / steps:.2f} ms/step, wall {1e3 * (time.perf_counter() - t0) / steps:.2f} ms/step')
pr = cProfile.Profile()
pr.enable()
for _ in range(steps):
    step(it); it += 1
pr.disable()
torch.cuda.synchronize()
st = pstats.Stats(pr)
st.sort_stats('tottime').print_stats(45)
st.sort_stats('cumtime').print_stats(60)
