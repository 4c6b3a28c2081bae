"""A/B of the LDS-DMA refill issued as one burst per K-step against the same pieces SPREAD between the K-step's MFMAs (r05; csrc/gemm.hip template parameter SPREAD; lab codes
1905 / 1305 / 1205 / 1106 / 1445 beside 1900 / 1300 / 1200 / 1100 / 1440).  us per launch inside a HIP graph, 8 weight buffers cycled, bit-level comparison (same arithmetic).
    python tools/micro/spread_lab.py"""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from vlaser_amd import ops, _lib as L
sys.path.insert(0, os.path.join(ROOT, 'tools', 'micro'))
from kernel_lab import timeit, rnd
BF = torch.bfloat16
SHAPES = [(560, 17920, 1536, 'sft gate/up forward', (1900, 1905)), (384, 17920, 1536, 'prefill gate/up', (1200, 1205)), (560, 8960, 1536, 'half gate/up', (1200, 1205, 1900, 1905)),
          (560, 1536, 8960, 'sft down, no split-K', (1100, 1106)), (384, 1536, 1536, 'prefill o', (1100, 1106)), (1025, 4096, 1024, 'vit fc1', (1440, 1445)), (1025, 1024, 4096, 'vit fc2, no split-K', (1440, 1445, 1100, 1106)),
          (3408, 8192, 3584, '8B-sized (M = 3408)', (1300, 1305, 1200, 1205))]
print('| shape | M | N | K | configuration | us per launch | TFLOP/s | == first |')
print('|---|---|---|---|---|---|---|---|')
for (M, N, K, name, cfgs) in SHAPES:
    x = rnd(M, K, std=1.0); ws = [rnd(N, K) for _ in range(8 if N * K < 1e8 else 3)]
    ref = None
    for cfg in cfgs:
        out = torch.zeros(M, N, dtype=BF, device='cuda')
        try:
            us = timeit([lambda w=w: ops.gemm(L.EPI_NONE, x, w, out=out, force_bm=cfg) for w in ws])
        except Exception as e:
            print(f'| {name} | | | | {cfg} | {str(e)[:80]} | | |'); continue
        if ref is None:
            ref = out.clone()
        print(f'| {name} | {M} | {N} | {K} | {cfg} | {us:.2f} | {2.0 * M * N * K / us / 1e6:.0f} | {torch.equal(out, ref)} |', flush=True)
