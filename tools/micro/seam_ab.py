"""Lab (r06): the split-K seam (GEMM -> fp32 slabs -> reduce_norm_exact: reduce + bias / layer scale / residual + norm) against the whole-K composition of kernels that already
exist (GEMM with the residual epilogue -> bf16 residual stream, then a stand-alone norm launch): same rounding points (h rounded to bf16 once, the norm taken of the rounded h),
in-graph time of the PAIR of launches, weights cycled over 8 buffers (HBM-cold as in the layer sequence).   python tools/micro/seam_ab.py"""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from vlaser_amd import ops, _lib as L
sys.path.insert(0, os.path.join(ROOT, 'tools', 'micro'))
from kernel_lab import timeit, rnd
BF = torch.bfloat16
print('| seam | M | N | K | split-K slabs + reduce_norm (splits) us | whole-K residual epilogue + norm launch us | whole-K GEMM alone us | max abs diff of x | of h |\n|---|---|---|---|---|---|---|---|---|')
for (name, M, N, K, kind) in [('ViT proj', 1025, 1024, 1024, 'ln'), ('ViT fc2', 1025, 1024, 4096, 'ln'), ('LLM o_proj (prefill)', 384, 1536, 1536, 'rms'), ('LLM down (prefill)', 384, 1536, 8960, 'rms'),
                              ('LLM o_proj (S = 560)', 560, 1536, 1536, 'rms'), ('LLM down (S = 560)', 560, 1536, 8960, 'rms'), ('8B o_proj (S = 3408)', 3408, 3584, 3584, 'rms'),
                              ('8B down (S = 3408)', 3408, 3584, 18944, 'rms'), ('ViT proj, 13 tiles', 13325, 1024, 1024, 'ln'), ('ViT fc2, 13 tiles', 13325, 1024, 4096, 'ln'),
                              ('2B o_proj (S = 3408)', 3408, 1536, 1536, 'rms'), ('2B down (S = 3408)', 3408, 1536, 8960, 'rms')]:
    a = rnd(M, K, std=1.0)
    ws = [rnd(N, K) for _ in range(8 if N * K < (1 << 26) else 3)]
    h0 = rnd(M, N, std=1.0)
    bias, ls = rnd(N, std=0.1), rnd(N, std=0.5)
    nw, nb = (1 + 0.1 * torch.randn(N, device='cuda')).to(BF), rnd(N, std=0.1)
    part = torch.zeros(ops.split_slab_elems(M, N), dtype=torch.float32, device='cuda')
    sp = ops.gemm_splits(M, N, K, part.numel())
    hA, xA, hB, xB = h0.clone(), torch.zeros(M, N, dtype=BF, device='cuda'), h0.clone(), torch.zeros(M, N, dtype=BF, device='cuda')

    def seam(w, h, x):
        ops.gemm(L.EPI_PARTIAL, a, w, out_f32=part, k_splits=sp)
        if kind == 'ln':
            ops.reduce_norm(h0, part, sp, M, N, h, x, bias=bias, ls=ls, norm=2, norm_w=nw, norm_b=nb, eps=1e-6)
        else:
            ops.reduce_norm(h0, part, sp, M, N, h, x, norm=1, norm_w=nw, eps=1e-6)

    def whole(w, h, x):
        if kind == 'ln':
            ops.gemm(L.EPI_BIAS_LS_RES, a, w, out=h, bias=bias, res=h0, ls=ls)
            ops.layernorm(h, nw, nb, 1e-6, out=x)
        else:
            ops.gemm(L.EPI_RES, a, w, out=h, res=h0)
            ops.rmsnorm(h, nw, 1e-6, out=x)

    tA = timeit([lambda w=w: seam(w, hA, xA) for w in ws])             # (timeit: us per list entry = per PAIR of launches)
    tB = timeit([lambda w=w: whole(w, hB, xB) for w in ws])
    tG = timeit([lambda w=w: ops.gemm(L.EPI_RES if kind == 'rms' else L.EPI_BIAS_LS_RES, a, w, out=hB, res=h0, **({} if kind == 'rms' else dict(bias=bias, ls=ls))) for w in ws])
    seam(ws[0], hA, xA); whole(ws[0], hB, xB)
    print(f'| {name} | {M} | {N} | {K} | {tA:.1f} ({sp}) | {tB:.1f} | {tG:.1f} | {(xA.float() - xB.float()).abs().max().item():.3g} | {(hA.float() - hB.float()).abs().max().item():.3g} |', flush=True)
