"""In-kernel timeline of attn_bwd_dq_kernel (lab build with -DAB_TIMELINE: tools/micro/lab_build/libvlaser_abtl.so, see the header of this script's
build line in profiles/): cycle stamps of wave 0 of the last query tile (longest key chain) at S = 560.
    VLASER_HIP_LIB=$PWD/tools/micro/lab_build/libvlaser_abtl.so [VLASER_ATTN_BWD_KS=1|2] python tools/micro/attn_bwd_timeline.py"""
import ctypes as C
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tools', 'micro'))
from vlaser_amd import ops, _lib as L  # noqa: E402
from kernel_lab import rnd  # noqa: E402

BF = torch.bfloat16
S, nq, nkv, hd, sm = 560, 12, 2, 128, 576
q = rnd(S, nq * hd); dao = rnd(S, nq * hd); Kc = rnd(nkv, sm, hd); vt = rnd(nkv, hd, sm)
out = torch.zeros(S, nq * hd, dtype=BF, device='cuda')
lse = torch.zeros(nq * S, dtype=torch.float32, device='cuda'); delta = torch.zeros_like(lse)
dq = torch.zeros(S, nq * hd, dtype=BF, device='cuda'); dk = torch.zeros_like(dq); dv = torch.zeros_like(dq)
ops.attn_prefill(q, Kc, vt, out, 1, S, S, nq, nkv, hd, (S * nq * hd, hd, nq * hd), (nkv * sm * hd, sm * hd), (nkv * hd * sm, hd * sm), (S * nq * hd, nq * hd), sm,
                 hd ** -0.5, L.ATTN_CAUSAL, lse_out=lse)
for _ in range(5):
    ops.attn_bwd(q, Kc, vt, out, dao, lse, delta, dq, dk, dv, S, nq, nkv, sm, hd ** -0.5)
torch.cuda.synchronize()
buf = (C.c_longlong * 128)()
assert L.lib().vlaser_attn_bwd_debug_read(buf) == 0
t = list(buf)[:64]
n = int(t[62])
print(f'{n} stamps; cycles since the first (s_memtime ticks), delta to previous')
names = ['start', 'prologue done']
prev = t[0]
for i in range(n):
    lab = names[i] if i < len(names) else ('tile staged' if (i - 2) % 2 == 0 else 'S/dP/dS done')
    if i == n - 2: lab = 'loop done'
    if i == n - 1: lab = 'end'
    print(f'{i:3d} {lab:16s} {t[i] - t[0]:8d} {t[i] - prev:8d}')
    prev = t[i]
