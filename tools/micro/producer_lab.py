"""A/B of the LDS-DMA GEMM ring with and without PRODUCER waves (r05: 4 extra waves, one per SIMD, issue every LDS-DMA piece; the 8 compute waves only pass the barrier,
read fragments and issue MFMAs -- csrc/gemm.hip, template parameter PRD; lab code 2100 beside 1100).  us per launch inside a HIP graph, 8 weight
buffers cycled, outputs compared bit for bit (same arithmetic, same order).   python tools/micro/producer_lab.py"""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from vlaser_amd import ops, _lib as L
sys.path.insert(0, os.path.join(ROOT, 'tools', 'micro'))
from kernel_lab import timeit, rnd
BF = torch.bfloat16
SHAPES = [(560, 1536, 8960, 'K-loop bound: sft down, no split-K', (1100, 2100)), (1025, 1024, 4096, 'vit fc2, no split-K', (1100, 2100)), (560, 2048, 1536, 'sft qkv', (1100, 2100)),
          (1025, 4096, 1024, 'vit fc1', (1100, 2100))]       # (2200 / 2900 = the 128x256 / 192x256 tiles with producers spilled at 12 waves' 168 registers and were removed)
print('| shape | M | N | K | configuration | us per launch | TFLOP/s | == first |')
print('|---|---|---|---|---|---|---|---|')
for (M, N, K, name, cfgs) in SHAPES:
    x = rnd(M, K, std=1.0); ws = [rnd(N, K) for _ in range(8)]
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
