"""us per launch of vlaser_vla_step (tail of an Euler step + action encoder in one launch) against the four launches it replaces, HIP-graph timed."""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tools', 'micro'))
from vlaser_amd import ops, _lib as L
from kernel_lab import timeit, rnd
BF, F32 = torch.bfloat16, torch.float32
dev = 'cuda'
M, W, ad, n = 4, 768, 7, 10
w1, b1, w2, b2, w3, b3 = rnd(W, ad), rnd(W), rnd(W, 2 * W), rnd(W), rnd(W, W), rnd(W)
wd, bd, nw = rnd(ad, W), rnd(ad), torch.ones(W, dtype=BF, device=dev)
w21, cs = ops.fold_action_encoder(w1, b1, w2, b2, W, ad, n, 10000.0)
h = rnd(M, W, std=1.0); parts = torch.randn(7, M, W, device=dev) * 0.1
a0, a1 = torch.zeros(16, ad, device=dev), torch.zeros(16, ad, device=dev)
vel = torch.zeros(16, ad, device=dev)
hout = torch.zeros(16, W, dtype=BF, device=dev)
fin = (h, parts, 7, M, 0, nw, 1e-6, wd, bd)
us = timeit([lambda: ops.vla_step(a0, a1, w21, cs[1], w3, b3, hout, M, W, ad, finish=fin, vel_out=vel, dt=0.1),
             lambda: ops.vla_step(a1, a0, w21, cs[2], w3, b3, hout, M, W, ad, finish=fin, vel_out=vel, dt=0.1)] * 4)
print(f'vla_step (finish + encoder): {us:.2f} us per launch')
us = timeit([lambda: ops.vla_step(a0, a0, w21, cs[0], w3, b3, hout, M, W, ad)] * 8)
print(f'vla_step (encoder only)    : {us:.2f} us per launch')
xcat, e2 = torch.zeros(16, 2 * W, dtype=BF, device=dev), torch.zeros(16, W, dtype=BF, device=dev)
w2p, w3p = ops.pack_skinny(w2), ops.pack_skinny(w3)


def four():
    ops.vla_euler(h, parts, 7, M, nw, 1e-6, wd, bd, a0, W, ad, 0.1, 0.0, False, vel_out=vel)
    ops.vla_prep(a0, w1, b1, xcat, M, W, ad, 0.1, 10000.0)
    ops.skinny(L.PRO_PLAIN, L.SK_BIAS_SILU, xcat, w2p, M, out=e2, ldo=W, bias=b2)
    ops.skinny(L.PRO_PLAIN, L.SK_BIAS, e2, w3p, M, out=hout, ldo=W, bias=b3)


us = timeit([four] * 4)
print(f'vla_euler + vla_prep + linear_2 + linear_3: {us:.2f} us per group of 4 launches')
