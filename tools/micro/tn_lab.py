"""Lab: TN weight-gradient GEMM (transposing LDS reads) vs the NT GEMM on pre-transposed operands, SFT shapes."""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tools', 'micro'))
from vlaser_amd import ops, _lib as L
from kernel_lab import timeit, rnd
BF = torch.bfloat16
S = 560
for (N, K, name) in [(17920, 1536, 'gate/up'), (1536, 8960, 'down'), (2048, 1536, 'qkv'), (1536, 1536, 'o')]:
    dys = [rnd(S, N, std=1.0) for _ in range(6)]; x = rnd(S, K, std=1.0)
    out = torch.zeros(N, K, dtype=BF, device='cuda')
    us = timeit([lambda d=d: ops.gemm_tn(d, x, out) for d in dys])
    Sp = 576
    dts = [torch.zeros(N, Sp, dtype=BF, device='cuda') for _ in range(6)]; xt = torch.zeros(K, Sp, dtype=BF, device='cuda')
    for d, t in zip(dys, dts): t[:, :S] = d.t()
    xt[:, :S] = x.t()
    out2 = torch.zeros(N, K, dtype=BF, device='cuda')
    us2 = timeit([lambda t=t: ops.gemm(L.EPI_NONE, t, xt, out=out2) for t in dts])
    fl = 2.0 * S * N * K
    dps = [torch.zeros(Sp, N, dtype=BF, device='cuda') for _ in range(6)]; xp = torch.zeros(Sp, K, dtype=BF, device='cuda')
    for d, t in zip(dys, dps): t[:S] = d
    xp[:S] = x
    out3 = torch.zeros(N, K, dtype=BF, device='cuda')
    res = []
    for cfg in (1100, 1200, 1300, 1140, 1240, 1340):
        try:
            u = timeit([lambda t=t: ops.gemm_tn_lds(t, xp, out3, Sp, force_cfg=cfg) for t in dps])
            res.append(f'{cfg}: {u:6.2f} us ({fl / u / 1e6:5.0f} TF)')
        except Exception as e:
            res.append(f'{cfg}: {type(e).__name__}')
    ops.gemm_tn_lds(dps[-1], xp, out3, Sp)
    print(f'   LDS-DMA TN  ' + '   '.join(res) + f'   max diff vs TN {(out3.float() - out.float()).abs().max().item():.3g}')
    print(f'wgrad {name:8s} [{N}x{K}] S={S}: TN {us:7.2f} us ({fl / us / 1e6:6.1f} TF)   NT on transposed {us2:7.2f} us ({fl / us2 / 1e6:6.1f} TF)   max diff {(out.float() - out2.float()).abs().max().item():.3g}')
