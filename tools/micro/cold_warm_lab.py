"""Lab (r06): does the gate/up weight stream of the <= 16-row path care where its 27.5 MB come from -- 28 separately allocated buffers (the model), 28 slices of one
contiguous allocation, the same buffer every launch (Infinity-Cache / TLB warm), 4 or 12 buffers cycled (inside / beyond the 256 MB Infinity Cache)?  In-graph us per launch.
python tools/micro/cold_warm_lab.py"""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from vlaser_amd import ops, _lib as L
sys.path.insert(0, os.path.join(ROOT, 'tools', 'micro'))
from kernel_lab import timeit, rnd
BF = torch.bfloat16
M, H, I, NL = 4, 768, 8960, 28
h = rnd(M, H, std=1.0); nw = torch.ones(H, dtype=BF, device='cuda')
parts = torch.randn(8, M, H, device='cuda') * 0.1
out = torch.zeros(M, I, dtype=BF, device='cuda'); hout = torch.zeros(M, H, dtype=BF, device='cuda')
raw = [rnd(2 * I, H) for _ in range(NL)]
def run(ws, name):
    us = timeit([lambda w=w: ops.skinny(L.PRO_NORM, L.SK_SWIGLU, h, w, M, partials=parts, n_partials=3, norm_w=nw, h_out=hout, out=out, ldo=I) for w in ws], reps=30)
    print(f'{name}: {us:.2f} us per launch ({2 * I * H * 2 / us / 1e3:.0f} GB/s)', flush=True)
wgu = [ops.pack_skinny(w, 1, 2) for w in raw]
run(wgu, '28 separately allocated weight buffers (as the model holds them)')
# one contiguous arena: the packed tensors copied into slices of ONE allocation
n = wgu[0].t.numel()
arena = torch.empty(NL * n, dtype=wgu[0].t.dtype, device='cuda')
import copy
wa = []
for i, w in enumerate(wgu):
    c = copy.copy(w); c.t = arena[i * n:(i + 1) * n].view_as(w.t); c.t.copy_(w.t); wa.append(c)
run(wa, '28 slices of ONE contiguous allocation')
run([wgu[0]] * NL, 'the SAME buffer 28 times (27.5 MB: warm in the Infinity Cache, TLB warm)')
run([wgu[i % 4] for i in range(NL)], '4 buffers cycled (110 MB: inside the 256 MB Infinity Cache)')
run([wgu[i % 12] for i in range(NL)], '12 buffers cycled (330 MB: beyond the Infinity Cache)')
