"""which Tensile / hipBLASLt kernels torch.matmul dispatches for the step's N = 768 shapes (run under rocprofv3 --kernel-trace --stats)"""
import torch
T = torch.bfloat16
for M in (16400, 8200):
    for N, K in ((768, 3072), (768, 2304), (768, 768), (2304, 768)):
        x = torch.randn(M, K, device='cuda').to(T); w = torch.randn(N, K, device='cuda').to(T); y = torch.empty(M, N, device='cuda', dtype=T)
        for _ in range(5):
            torch.matmul(x, w.t(), out=y)
torch.cuda.synchronize()
