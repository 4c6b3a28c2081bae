"""Lab: split-K factor of the N = 1536 GEMMs of the SFT step (M = 560): PARTIAL slabs + reduce vs the single-pass kernel."""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tools', 'micro'))
from vlaser_amd import ops, _lib as L
from kernel_lab import timeit, rnd
BF = torch.bfloat16
M, N = 560, 1536
for K in (1536, 2048, 8960, 17920):
    x = rnd(M, K, std=1.0); ws = [rnd(N, K) for _ in range(6)]
    out = torch.zeros(M, N, dtype=BF, device='cuda')
    res = []
    for bm in (32, 64):
        us = timeit([lambda w=w: ops.gemm(L.EPI_NONE, x, w, out=out, force_bm=bm) for w in ws])
        res.append(f'none/bm{bm} {us:.1f}')
    for S in (2, 3, 4, 5, 7, 8):
        if K % (S * 64): continue
        part = torch.zeros(S, M, N, device='cuda')
        us = timeit([lambda w=w: (ops.gemm(L.EPI_PARTIAL, x, w, out_f32=part, k_splits=S), ops.reduce_norm(None, part, S, M, N, out)) for w in ws])   # per pair of launches
        res.append(f'S{S} {us:.1f}')
    print(f'M={M} N={N} K={K}: default splits {ops.gemm_splits(M, N, K)} |', '  '.join(res))
