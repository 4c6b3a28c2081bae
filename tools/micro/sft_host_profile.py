"""Where the HOST time of one SFT forward + backward goes (cProfile over 5 calls, device idle-synced between them).   python tools/micro/sft_host_profile.py"""
import cProfile
import os
import pstats
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from vlaser_amd import config as C, synth  # noqa: E402
from vlaser_amd.sft import SFTModel  # noqa: E402

torch.set_grad_enabled(False)
cfg = C.vlaser_2b()
m = SFTModel(cfg, device='cuda:0', max_seq_len=576)
m.load_state_dict(synth.vlm_state_dict(cfg, device='cuda:0', dtype=torch.bfloat16))
g = torch.Generator().manual_seed(1000)
S, R = 560, 128
ids = torch.cat([torch.randint(1, 151643, (41,), generator=g), torch.full((256,), cfg.img_context_token_id), torch.randint(1, 151643, (S - 41 - 256,), generator=g)])[None]
labels = torch.full_like(ids, -100)
labels[0, -R:] = ids[0, -R:]
pv = torch.randn(1, 3, 448, 448, generator=g).to('cuda:0').to(torch.bfloat16)
for _ in range(3):
    m.forward_backward(pv, ids, labels)
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
for _ in range(5):
    m.forward_backward(pv, ids, labels)
pr.disable()
torch.cuda.synchronize()
st = pstats.Stats(pr)
st.sort_stats('tottime').print_stats(22)
