"""Yardstick lab (r06): this package's LDS-DMA GEMM beside the vendor library GEMM (torch.mm -> hipBLASLt / rocBLAS) on the path's own shapes, same box, same
method: us per launch inside a HIP graph, 8 weight buffers cycled (every launch misses L2 for its weights, as in the layer sequence), plain bf16 output, no epilogue
on either side (the product fuses bias / residual / SwiGLU / norms into these launches; the library side would need extra launches for them -- not counted here).
Not a product path: the library is only measured.   python tools/micro/blaslt_ab.py"""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from vlaser_amd import ops, _lib as L
sys.path.insert(0, os.path.join(ROOT, 'tools', 'micro'))
from kernel_lab import timeit, rnd
BF = torch.bfloat16
SHAPES = [(384, 2048, 1536, 'prefill qkv'), (384, 1536, 1536, 'prefill o_proj'), (384, 17920, 1536, 'prefill gate/up'), (384, 1536, 8960, 'prefill down'),
          (1025, 3072, 1024, 'vit qkv'), (1025, 1024, 1024, 'vit proj'), (1025, 4096, 1024, 'vit fc1'), (1025, 1024, 4096, 'vit fc2'),
          (560, 2048, 1536, 'sft qkv'), (560, 17920, 1536, 'sft gate/up'), (560, 1536, 8960, 'sft down'), (560, 151680, 1536, 'sft lm_head'),
          (3408, 4608, 3584, '8B qkv'), (3408, 37888, 3584, '8B gate/up'), (3408, 3584, 18944, '8B down'), (4096, 4096, 4096, 'square 4096'), (8192, 8192, 8192, 'square 8192')]
print('| shape | M | N | K | this package us (TFLOP/s) | torch.mm us (TFLOP/s) | ratio lib / ours | max abs diff |\n|---|---|---|---|---|---|---|---|')
for (M, N, K, name) in SHAPES:
    nbuf = 8 if N * K < (1 << 28) else 2
    x = rnd(M, K, std=1.0); ws = [rnd(N, K) for _ in range(nbuf)]
    out = torch.zeros(M, N, dtype=BF, device='cuda')
    out2 = torch.zeros(M, N, dtype=BF, device='cuda')
    us = timeit([lambda w=w: ops.gemm(L.EPI_NONE, x, w, out=out) for w in ws])
    wts = [w.t() for w in ws]
    try:
        us2 = timeit([lambda w=w: torch.mm(x, w, out=out2) for w in wts])
    except Exception as e:
        print(f'| {name} | {M} | {N} | {K} | {us:.1f} | torch.mm not capturable: {str(e)[:60]} | | |'); continue
    fl = 2.0 * M * N * K / 1e6
    print(f'| {name} | {M} | {N} | {K} | {us:.1f} ({fl / us:.0f}) | {us2:.1f} ({fl / us2:.0f}) | {us2 / us:.2f} | {(out.float() - out2.float()).abs().max().item():.3g} |', flush=True)
