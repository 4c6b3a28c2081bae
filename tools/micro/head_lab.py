import os, sys
import torch
ROOT = os.environ.get('GRAFT_REPO_ROOT', '/root/repo')
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tools', 'micro'))
from vlaser_amd import ops, _lib as L
from kernel_lab import timeit, rnd
BF = torch.bfloat16
R, H, Vp = 128, 1536, 151680
dlog = rnd(R, Vp, std=1.0)
ws = [rnd(Vp, H) for _ in range(3)]
part = torch.zeros(64 * R * H, dtype=torch.float32, device='cuda')
out = torch.zeros(R, H, dtype=BF, device='cuda')
print('chooser picks', ops.gemm_splits(R, H, Vp, nn=True))
for sp in (5, 8, 10, 15, 16, 24, 30):
    if Vp % (sp * 64):
        continue
    for cfg in (0, 1500, 1100):
        try:
            us = timeit([lambda w=w: ops.gemm_nn(L.EPI_PARTIAL, dlog, w, out_f32=part, k_splits=sp, force_bm=cfg) for w in ws])
            us2 = timeit([lambda: ops.reduce_norm(None, part, sp, R, H, out)] * 4) if sp <= 8 else float('nan')
            print(f'lm_head dgrad NN PARTIAL x{sp:2d} cfg {cfg:4d}: {us:7.1f} us ({466e6 / us / 1e6:.2f} TB/s)   reduce {us2:.1f} us')
        except Exception as e:
            print(sp, cfg, str(e)[:90])
