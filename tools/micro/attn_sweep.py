"""GPU-box lab: prefill attention time vs sequence length (fixed cost vs per-key-tile cost); VLASER_ATTN_KS selects the in-workgroup key split."""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tools', 'micro'))
from vlaser_amd import ops, _lib as L
from kernel_lab import timeit, rnd
BF = torch.bfloat16
for (nq, nkv, hd, mode) in [(16, 16, 64, L.ATTN_FULL), (12, 2, 128, L.ATTN_CAUSAL), (12, 2, 128, L.ATTN_FULL)]:
    for S in (64, 128, 256, 512, 1024, 2048):
        B = 1
        Sp = (S + 63) // 64 * 64
        q = rnd(B, S, nq * hd); k = rnd(B, nkv, Sp, hd); vt = rnd(B, nkv, hd, Sp)
        out = torch.zeros(B, S, nq * hd, dtype=BF, device='cuda')
        f = lambda: ops.attn_prefill(q, k, vt, out, B, S, S, nq, nkv, hd, (S * nq * hd, hd, nq * hd), (nkv * Sp * hd, Sp * hd), (nkv * hd * Sp, hd * Sp),
                                     (S * nq * hd, nq * hd), Sp, hd ** -0.5, mode)
        us = timeit([f] * 8)
        print(f'ks={os.environ.get("VLASER_ATTN_KS", "auto")} heads {nq}/{nkv} hd {hd} mode {mode} S={S:5d}: {us:8.2f} us, {us / (Sp // 64):6.2f} us per key tile, blocks {Sp // 64 * nq}')
