"""A/B of the LDS-DMA refill issued as one burst per K-step against the same pieces SPREAD between the K-step's MFMAs (r05; csrc/gemm.hip template parameter SPREAD; since adopted: the
product codes 1100 / 1200 / 1300 / 1440 / 1500 / 1900 ARE the spread form, 1101 / 1201 / 1304 / 1441 / 1501 / 1904 the burst form of r02-r04).  us per launch inside a HIP graph, 8 weight buffers cycled, bit-level comparison (same arithmetic).
    python tools/micro/spread_lab.py"""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from vlaser_amd import ops, _lib as L
sys.path.insert(0, os.path.join(ROOT, 'tools', 'micro'))
from kernel_lab import timeit, rnd
BF = torch.bfloat16
SHAPES = [(560, 17920, 1536, 'sft gate/up forward', (1904, 1900)), (384, 17920, 1536, 'prefill gate/up', (1201, 1200)), (560, 8960, 1536, 'half gate/up', (1201, 1200, 1904, 1900)),
          (560, 1536, 8960, 'sft down, no split-K', (1101, 1100)), (384, 1536, 1536, 'prefill o', (1101, 1100, 1501, 1500)), (384, 2560, 1536, 'prefill qkv', (1564, 1501, 1500)),
          (1025, 4096, 1024, 'vit fc1', (1441, 1440)), (1025, 1024, 4096, 'vit fc2, no split-K', (1441, 1440, 1101, 1100)), (3408, 8192, 3584, '8B-sized (M = 3408)', (1304, 1300, 1201, 1200))]
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

# (the TN form and the 4-wave 64x64 tile were measured with the spread refill in r05z and keep the burst: profiles/r05z_spread_lab.md)
