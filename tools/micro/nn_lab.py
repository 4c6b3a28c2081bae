"""GPU-box lab: the NN dgrad GEMM (weight read as stored) against the NT kernel on a transposed copy, SFT shapes (HIP-graph timed)."""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tools', 'micro'))
from vlaser_amd import ops, _lib as L
from kernel_lab import timeit, rnd
BF = torch.bfloat16
for (name, M, Kc, N) in [('dgrad qkv', 560, 2048, 1536), ('dgrad o', 560, 1536, 1536), ('dgrad gate/up', 560, 17920, 1536), ('dgrad down', 560, 1536, 8960),
                         ('dgrad head', 128, 151680, 1536), ('vit-size', 1025, 4096, 1024)]:
    x = rnd(M, Kc); w = rnd(Kc, N, std=0.03); wt = w.t().contiguous()
    sp = ops.gemm_splits(M, N, Kc)
    part = torch.zeros(max(sp, 1) * M * N, dtype=torch.float32, device='cuda')
    out = torch.zeros(M, N, dtype=BF, device='cuda')
    if sp > 1:
        f_nt = lambda: ops.gemm(L.EPI_PARTIAL, x, wt, out_f32=part, k_splits=sp)
        f_nn = lambda: ops.gemm_nn(L.EPI_PARTIAL, x, w, out_f32=part, k_splits=sp)
    else:
        f_nt = lambda: ops.gemm(L.EPI_NONE, x, wt, out=out)
        f_nn = lambda: ops.gemm_nn(L.EPI_NONE, x, w, out=out)
    t_nt, t_nn = timeit([f_nt] * 8), timeit([f_nn] * 8)
    t_1 = timeit([lambda: ops.gemm_nn(L.EPI_NONE, x, w, out=out)] * 8) if Kc <= 4096 else float('nan')
    t_32 = timeit([lambda: ops.gemm(L.EPI_NONE, x, wt, out=out, force_bm=32)] * 8) if Kc <= 4096 else float('nan')
    fl = 2.0 * M * N * Kc
    print(f'{name:14s} M={M} K={Kc} N={N} splits={sp}: NT on W^T {t_nt:7.2f} us ({fl / t_nt / 1e6:6.1f} TF)   NN on W {t_nn:7.2f} us ({fl / t_nn / 1e6:6.1f} TF)   NN one pass {t_1:7.2f} us   NT 32-row one pass {t_32:7.2f} us')
