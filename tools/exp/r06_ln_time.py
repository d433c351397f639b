"""round 6: LayerNorm backward alone at the encoder layer's shape (16 x 1025 rows of 768), per kernel form / grid cap
(S4F_LN_BWD_THIN = 0: two rows per wave, n: thin form with at most n blocks).  python tools/exp/r06_ln_time.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from s4former_amd import kernels as K  # noqa: E402

T = torch.bfloat16
dev = 'cuda'
Bn, N, E = 16, 1025, 768
M = Bn * N
dy = (torch.randn(M, E, device=dev) * 0.05).to(T)
x = torch.randn(Bn, N, E, device=dev)
mean = torch.zeros(M, device=dev); rstd = torch.ones(M, device=dev); gamma = torch.ones(E, device=dev)
dres = torch.randn(Bn, N, E, device=dev); dx = torch.empty(Bn, N, E, device=dev); dx_t = torch.empty(Bn, N, E, device=dev, dtype=T)
dg = torch.zeros(E, device=dev); db = torch.zeros(E, device=dev); dcs = torch.zeros(E, device=dev)


def f(cs=True):
    K.layernorm_bwd(dy, x, mean, rstd, gamma, dres, dx, dx_t, dg, db, M, E, 1, dcolsum=dcs if cs else None)


for cs in (True, False):
    for _ in range(3):
        f(cs)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        f(cs)
    e1.record()
    torch.cuda.synchronize()
    print(f"lib {os.path.basename(os.environ.get('S4F_LIB', 'default'))} S4F_LN_BWD_THIN={os.environ.get('S4F_LN_BWD_THIN', '-')} colsum={cs}: {e0.elapsed_time(e1) / 20 * 1e3:7.1f} us", flush=True)
