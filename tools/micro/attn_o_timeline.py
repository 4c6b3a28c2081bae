"""In-kernel timeline of attn_oproj_kernel (lab build -DAO_TIMELINE -> tools/micro/lab_build/libvlaser_aotl.so): wave 0 of workgroup (0, 0), expert geometry.
    VLASER_HIP_LIB=$PWD/tools/micro/lab_build/libvlaser_aotl.so python tools/micro/attn_o_timeline.py"""
import ctypes as C
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tools', 'micro'))
from vlaser_amd import ops, _lib as L  # noqa: E402
from kernel_lab import rnd, timeit  # noqa: E402

nq_tok, kv_len, nq, nkv, smax, H = 4, 389, 12, 2, 448, 768
q = rnd(nq_tok, nq * 128, std=1.0)
ks = [rnd(1, nkv, smax, 128, std=1.0) for _ in range(8)]; vts = [rnd(1, nkv, 128, smax, std=1.0) for _ in range(8)]
wos = [rnd(H, nq * 128) for _ in range(8)]
valid = torch.tensor([277], dtype=torch.int32, device='cuda')
parts = ops.attn_partial_buffers(1, nkv, 'cuda')
out = torch.zeros(nkv, nq_tok, H, dtype=torch.float32, device='cuda')
args = [ops.attn_skinny_args(q, k, vt, parts, 1, nq_tok, kv_len, nq, nkv, 128, (nq_tok * nq * 128, 128, nq * 128), (nkv * smax * 128, smax * 128), (nkv * 128 * smax, 128 * smax),
                             smax, 128 ** -0.5, L.ATTN_PREFIX, 1, valid_len=valid, blk_start=384) for k, vt in zip(ks, vts)]
us = timeit([lambda a=a, w=w: ops.launch_attn_oproj(a, w, out, H) for a, w in zip(args, wos)])
print(f'attn_oproj: {us:.2f} us per launch (8 K / V / W_o sets cycled)')
if hasattr(L.lib(), 'vlaser_attn_oproj_debug_read'):
    ops.launch_attn_oproj(args[0], wos[0], out, H); torch.cuda.synchronize()
    buf = (C.c_longlong * 32)()
    L.lib().vlaser_attn_oproj_debug_read(buf)
    t = list(buf)[:8]
    names = ['start', 'Q + chunk 0 + W_o requested', 'chunk 0 processed', 'all chunks processed', 'partials in LDS (barrier)', 'merged rows in LDS (barrier)', 'GEMV partials in LDS (barrier)', 'end']
    for i, n in enumerate(names):
        print(f'{n:34s} {t[i] - t[0]:8d} cycles  (+{t[i] - t[max(i - 1, 0)]})')
