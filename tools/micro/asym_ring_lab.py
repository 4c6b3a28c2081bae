"""A/B of the two-stage LDS-DMA GEMM rings against the ASYMMETRIC ring (r05: a third stage for the W operand alone -- weights two steps ahead, activations one; csrc/gemm.hip,
template parameter ASYM; since adopted: configuration codes 1900 / 1300 ARE the asymmetric rings, 1901 / 1302 / 1301 the r03-r04 two-stage rings, 1903 = asymmetric with the refill requested first).  us per launch
inside a HIP graph, 8 weight buffers cycled (HBM-cold weights as in the layer sequence), outputs compared bit for bit.   python tools/micro/asym_ring_lab.py"""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from vlaser_amd import ops, _lib as L
sys.path.insert(0, os.path.join(ROOT, 'tools', 'micro'))
from kernel_lab import timeit, rnd
BF = torch.bfloat16
SHAPES = [(560, 17920, 1536, 'sft gate/up forward', (1901, 1903, 1900, 1200)), (560, 8960, 1536, 'half gate/up', (1901, 1903, 1900)), (1025, 4096, 1024, 'vit fc1', (1440, 1901, 1900)),
          (3408, 8192, 3584, '8B-sized (M = 3408)', (1302, 1301, 1300, 1901, 1900, 1200)), (3408, 37888, 3584, '8B gate/up', (1302, 1300, 1900)), (3408, 3584, 18944, '8B down', (1302, 1300, 1900))]
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

# (TN form, r05u: a third stage for the N-side operand gave 42.4 against 41.9 us on the gate/up weight gradient, 22.2 against 21.6 on down's: not kept)
