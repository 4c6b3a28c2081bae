"""GEMM ablation lab: time a few path shapes with alternative builds of the library (tools/micro/lab_build/*.so,
built with -DGEMM_ABLATE=n: 1 = no global loads in the K loop, 2 = no LDS stores, 3 = no MFMA)."""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import vlaser_amd._lib as L
if len(sys.argv) > 1 and sys.argv[1] != 'base':
    L.LIB_PATH = os.path.join(ROOT, 'tools', 'micro', 'lab_build', sys.argv[1])
from vlaser_amd import ops
sys.path.insert(0, os.path.join(ROOT, 'tools', 'micro'))
from kernel_lab import timeit, rnd
BF = torch.bfloat16
for (M, N, K, name, bm) in [(1025, 3072, 1024, 'vit qkv', 128), (1025, 3072, 1024, 'vit qkv', 64), (560, 17920, 1536, 'sft gate/up', 128),
                            (560, 17920, 1536, 'sft gate/up', 64), (17920, 1536, 576, 'sft wgrad gu', 128), (384, 1536, 8960, 'llm down', 32)]:
    x = rnd(M, K, std=1.0); ws = [rnd(N, K) for _ in range(8)]
    out = torch.zeros(M, N, dtype=BF, device='cuda')
    us = timeit([lambda w=w: ops.gemm(L.EPI_NONE, x, w, out=out, force_bm=bm) for w in ws])
    print(f'{sys.argv[1] if len(sys.argv) > 1 else "base":12s} {name:12s} M={M} N={N} K={K} bm={bm}: {us:7.2f} us {2.0 * M * N * K / us / 1e6:7.1f} TF')
