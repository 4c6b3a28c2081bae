"""In-kernel timeline of the skinny kernel (wall_clock64 stamps, 100 MHz) for the action-expert shapes."""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from vlaser_amd import ops, _lib as L
BF = torch.bfloat16; dev = 'cuda'
rnd = lambda *s, std=0.03: (torch.randn(*s, device=dev) * std).to(BF)
M, H, I = 4, 768, 8960
h = rnd(M, H, std=1.0); nw = torch.ones(H, dtype=BF, device=dev)
parts = torch.randn(8, M, H, device=dev) * 0.1
out = torch.zeros(M, I, dtype=BF, device=dev); hout = torch.zeros(M, H, dtype=BF, device=dev)
for tpu in (2, 1):
    ws = [ops.pack_skinny(rnd(2 * I, H), 1, tpu) for _ in range(6)]
    dbg = torch.zeros(256 * 8, dtype=torch.int64, device=dev)
    for w in ws:   # last launch is HBM-cold for its own weights, I-cache warm-ish
        dbg.zero_()
        ops.skinny(L.PRO_NORM, L.SK_SWIGLU, h, w, M, partials=parts, n_partials=3, norm_w=nw, h_out=hout, out=out, ldo=I, dbg=dbg)
    torch.cuda.synchronize()
    d = dbg.view(256, 8).cpu()
    d = d[d[:, 0] > 0]
    t0 = d[:, 0].min()
    rel = (d - t0).float() * 10  # ns
    names = ['start', 'phase1 done', 'prologue done', 'first batch MFMA done', 'first unit reduce barrier', 'end']
    print(f'--- gate/up NORM+SWIGLU tpu={tpu}: {d.shape[0]} blocks; ns relative to earliest block start')
    for i, n in enumerate(names):
        c = rel[:, i]
        print(f'  {n:28s} min {c.min():8.0f} median {c.median():8.0f} max {c.max():8.0f}')
