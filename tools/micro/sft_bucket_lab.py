"""SFT step time vs layers per gradient / AdamW bucket (one rank: finer buckets shorten the forward's tail behind the pipelined AdamW).   for b in 4 2 1 4 2 1; do python tools/micro/sft_bucket_lab.py $b; done"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from vlaser_amd import config as C, synth  # noqa: E402
from vlaser_amd.sft import SFTModel  # noqa: E402

torch.set_grad_enabled(False)
cfg = C.vlaser_2b()
sd = synth.vlm_state_dict(cfg, device='cuda:0', dtype=torch.bfloat16)
g = torch.Generator().manual_seed(1000)
S, R = 560, 128
ids = torch.cat([torch.randint(1, 151643, (41,), generator=g), torch.full((256,), cfg.img_context_token_id), torch.randint(1, 151643, (S - 41 - 256,), generator=g)])[None]
labels = torch.full_like(ids, -100)
labels[0, -R:] = ids[0, -R:]
pv = torch.randn(1, 3, 448, 448, generator=g).to('cuda:0').to(torch.bfloat16)
bl = int(sys.argv[1])
m = SFTModel(cfg, device='cuda:0', max_seq_len=576, bucket_layers=bl)
m.load_state_dict(sd)
del sd
torch.cuda.empty_cache()
for _ in range(3):
    m.step(pv, ids, labels)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(20):
    m.step(pv, ids, labels)
torch.cuda.synchronize()
print(f'bucket_layers {bl}: {len(m.buckets):2d} buckets, {(time.perf_counter() - t0) / 20 * 1e3:.2f} ms per step')
