"""GPU-box lab: prefill attention kernel time for the path's shapes (HIP-graph timed); run with VLASER_ATTN_QT1=1 to force one query tile per wave."""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tools', 'micro'))
from vlaser_amd import ops, _lib as L
from kernel_lab import timeit, rnd
BF = torch.bfloat16
for (name, B, S, nq, nkv, hd, mode) in [('vit T=1', 1, 1025, 16, 16, 64, L.ATTN_FULL), ('vit T=13', 13, 1025, 16, 16, 64, L.ATTN_FULL), ('2B S=384', 1, 384, 12, 2, 128, L.ATTN_CAUSAL),
                                        ('2B S=560', 1, 560, 12, 2, 128, L.ATTN_CAUSAL), ('8B S=3408', 1, 3408, 28, 4, 128, L.ATTN_CAUSAL)]:
    Sp = (S + 63) // 64 * 64
    q = rnd(B, S, nq * hd); k = rnd(B, nkv, Sp, hd); vt = rnd(B, nkv, hd, Sp)
    out = torch.zeros(B, S, nq * hd, dtype=BF, device='cuda')
    f = lambda: ops.attn_prefill(q, k, vt, out, B, S, S, nq, nkv, hd, (S * nq * hd, hd, nq * hd), (nkv * Sp * hd, Sp * hd), (nkv * hd * Sp, hd * Sp),
                                 (S * nq * hd, nq * hd), Sp, hd ** -0.5, mode)
    us = timeit([f] * 8)
    fl = 4.0 * B * nq * S * S * hd * (0.5 if mode == L.ATTN_CAUSAL else 1.0)
    print(f'{os.environ.get("VLASER_ATTN_QT1", "0")} {name:10s}: {us:8.2f} us  {fl / us / 1e6:7.1f} TF (causal counted as half)')
