"""Tile / ring-depth lab for the LDS-DMA GEMM on the path's latency-bound single-round shapes: us per launch inside a HIP graph (8 weight
buffers cycled: every launch misses L2 for its weights as in the layer sequence), all configurations checked against the register-staged
64-row kernel.   python tools/micro/gemm_cfg_lab.py [fused]"""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from vlaser_amd import ops, _lib as L
sys.path.insert(0, os.path.join(ROOT, 'tools', 'micro'))
from kernel_lab import timeit, rnd
BF = torch.bfloat16
CFGS = [int(c) for c in os.environ.get('CFGS', '64,1500,1506,1532,1564,1100,1105,1440,1200').split(',')]
SHAPES = [(384, 2048, 1536, 'llm qkv'), (384, 1536, 1536, 'llm o_proj'), (560, 2048, 1536, 'sft qkv'), (560, 1536, 1536, 'sft o_proj'),
          (1025, 1024, 1024, 'vit proj'), (1025, 3072, 1024, 'vit qkv'), (1025, 4096, 1024, 'vit fc1'), (256, 1536, 4096, 'mlp1.1'), (384, 17920, 1536, 'llm gate/up')]
for (M, N, K, name) in SHAPES:
    x = rnd(M, K, std=1.0); ws = [rnd(N, K) for _ in range(8)]
    ref = None
    for cfg in CFGS:
        out = torch.zeros(M, N, dtype=BF, device='cuda')
        try:
            us = timeit([lambda w=w: ops.gemm(L.EPI_NONE, x, w, out=out, force_bm=cfg) for w in ws])
        except Exception as e:
            print(f'{name:12s} cfg {cfg}: {str(e)[:80]}'); continue
        if ref is None:
            ref = out.clone()
        blocks = {64: (64, 128), 1500: (64, 128), 1506: (64, 128), 1532: (32, 128), 1564: (64, 64), 1100: (128, 128), 1105: (128, 128), 1440: (144, 128), 1200: (128, 256), 1300: (256, 256), 1900: (192, 256)}[cfg]
        nb = -(-M // blocks[0]) * -(-N // blocks[1])
        print(f'{name:12s} M={M:5d} N={N:5d} K={K:5d} cfg {cfg:5d} ({nb:4d} wgs): {us:7.2f} us {2.0 * M * N * K / us / 1e6:7.1f} TF  maxdiff {(out.float() - ref.float()).abs().max().item():.3g}')
    print()
