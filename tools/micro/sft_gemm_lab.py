"""Tile lab for the big GEMMs of the SFT step at S = 560 rows (Qwen2-1.5B widths): us per launch inside a HIP graph (8 weight buffers cycled), per
LDS-DMA tile configuration.   python tools/micro/sft_gemm_lab.py"""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tools', 'micro'))
from vlaser_amd import ops, _lib as L
from kernel_lab import timeit, rnd
BF = torch.bfloat16
S, H, I, NQ = 560, 1536, 8960, 2048
CFGS = [int(c) for c in os.environ.get('CFGS', '0,1100,1200,1300,1900').split(',')]


def row(name, fl, fn):
    res = []
    for cfg in CFGS:
        try:
            us = timeit(fn(cfg))
            res.append(f'{cfg}: {us:6.2f} us ({fl / us / 1e6:4.0f} TF)')
        except Exception as e:
            res.append(f'{cfg}: {type(e).__name__}')
    print(f'{name:34s} ' + '   '.join(res))


x = rnd(S, H, std=1.0)
wgu = [rnd(2 * I, H) for _ in range(8)]
act, gu = torch.zeros(S, I, dtype=BF, device='cuda'), torch.zeros(S, 2 * I, dtype=BF, device='cuda')
row('gate/up fwd NT SWIGLU+aux [560x17920x1536]', 2.0 * S * 2 * I * H, lambda c: [lambda w=w: ops.gemm(L.EPI_SWIGLU, x, w, out=act, aux_out=gu, ld_aux=gu.stride(0), force_bm=c) for w in wgu])
dgu = rnd(S, 2 * I, std=1.0)
dx = torch.zeros(S, H, dtype=BF, device='cuda')
part = torch.zeros(8 * S * H, dtype=torch.float32, device='cuda')
for sp in (1, 2, 4):
    row(f'gate/up dgrad NN PARTIAL x{sp} [560x1536x17920]', 2.0 * S * 2 * I * H, lambda c: [lambda w=w: ops.gemm_nn(L.EPI_PARTIAL, dgu, w, out_f32=part, k_splits=sp, force_bm=c) for w in wgu])
wd = [rnd(H, I) for _ in range(8)]
dh = rnd(S, H, std=1.0)
dgu_o = torch.zeros(S, 2 * I, dtype=BF, device='cuda')
row('down dgrad NN SWIGLU_BWD [560x8960x1536]', 2.0 * S * I * H, lambda c: [lambda w=w: ops.gemm_nn(L.EPI_SWIGLU_BWD, dh, w, out=dgu_o, res=gu, force_bm=c) for w in wd])
dact_o = torch.zeros(S, I, dtype=BF, device='cuda')
row('down dgrad NN NONE (no epilogue) [560x8960x1536]', 2.0 * S * I * H, lambda c: [lambda w=w: ops.gemm_nn(L.EPI_NONE, dh, w, out=dact_o, force_bm=c) for w in wd])
a_ = rnd(S, I, std=1.0)
for sp in (1, 2, 4):
    row(f'down fwd NT PARTIAL x{sp} [560x1536x8960]', 2.0 * S * I * H, lambda c: [lambda w=w: ops.gemm(L.EPI_PARTIAL, a_, w, out_f32=part, k_splits=sp, force_bm=c) for w in wd])
print('splits the step picks: down fwd', ops.gemm_splits(S, H, I), ' gate/up dgrad', ops.gemm_splits(S, H, 2 * I, nn=True))
