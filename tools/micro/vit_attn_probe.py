"""ViT attention vs token count: what the 1025th token (cls) costs -- T = 1024 is 256 workgroups and 8 full 128-key tiles, T = 1025 is 272 workgroups
(16 CUs host two) and a 9th key tile holding one key.   python tools/micro/vit_attn_probe.py"""
import os, sys
import torch
ROOT = os.environ.get('GRAFT_REPO_ROOT', '/root/repo')
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tools', 'micro'))
from vlaser_amd import ops, _lib as L
from kernel_lab import timeit, rnd
BF = torch.bfloat16
Sp, C = 1088, 1024
qv = rnd(1, 16, Sp, 64, std=1.0); kv = rnd(1, 16, Sp, 64, std=1.0); vv = rnd(1, 16, 64, Sp, std=1.0)
for T in (960, 1024, 1025, 1088):
    outv = torch.zeros(1, T, C, dtype=BF, device='cuda')
    us = timeit([lambda: ops.attn_prefill(qv, kv, vv, outv, 1, T, T, 16, 16, 64, (16 * Sp * 64, Sp * 64, 64), (16 * Sp * 64, Sp * 64), (16 * 64 * Sp, 64 * Sp), (T * C, C), Sp, 1.0, L.ATTN_FULL)] * 8)
    print(f'ViT attention T={T}: {us:.2f} us  ({(T + 63) // 64 * 16} workgroups)')
