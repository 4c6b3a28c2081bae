"""SFT backward's largest launch: dX = d(gate|up) [560, 17920] @ W_gu [17920, 1536] (NN form, split-K slabs + reduce): splits x tile configuration.
    python tools/micro/dgrad_gu_lab.py"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tools', 'micro'))
from vlaser_amd import ops, _lib as L  # noqa: E402
from kernel_lab import rnd, timeit  # noqa: E402
BF = torch.bfloat16
S, N, K = 560, 1536, 17920
dgu = rnd(S, K, std=1.0)
ws = [rnd(K, N) for _ in range(6)]
out = torch.zeros(S, N, dtype=BF, device='cuda')
part = torch.zeros(16 * S * N, dtype=torch.float32, device='cuda')
print('chooser picks', ops.gemm_splits(S, N, K, nn=True), 'splits')
for cfg in (0, 1200, 1900, 1300, 1100):
    row = []
    for sp in (4, 5, 7, 8, 10, 14):
        if K % (64 * sp):
            continue
        try:
            us = timeit([lambda w=w: (ops.gemm_nn(L.EPI_PARTIAL, dgu, w, out_f32=part, k_splits=sp, force_bm=cfg), ops.reduce_norm(None, part, sp, S, N, out)) for w in ws])
            row.append(f'{sp:2d} splits {us:6.1f}')
        except Exception as e:      # noqa: BLE001
            row.append(f'{sp:2d} splits {type(e).__name__}')
    print(f'cfg {cfg:5d}: ' + '   '.join(row) + '   (us per GEMM + reduce)')
