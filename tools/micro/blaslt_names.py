"""Which library kernels torch.mm picks for the path's shapes (run under rocprofv3 --kernel-trace --stats; the kernel names carry the macro-tile and the schedule switches)."""
import torch
BF = torch.bfloat16
SHAPES = [(384, 17920, 1536), (1025, 4096, 1024), (560, 151680, 1536), (3408, 37888, 3584), (3408, 3584, 18944), (4096, 4096, 4096), (8192, 8192, 8192)]
for (M, N, K) in SHAPES:
    x = torch.randn(M, K, device='cuda').to(BF); w = (torch.randn(N, K, device='cuda') * 0.03).to(BF)
    for _ in range(5):
        y = torch.mm(x, w.t())
    torch.cuda.synchronize()
